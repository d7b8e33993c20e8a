#!/bin/bash
# GPU box: per-class roofline of the trunk's convolutions (kernel trace + two PMC passes).   tools/conv_classes.sh <tag>
set -u
TAG=${1:-cc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/conv_classes.py 10 | tail -1 > $OUT/plan.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o p -- python tools/conv_classes.py 10 > /dev/null 2> $OUT/kt.err
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python tools/conv_classes.py 10 > /dev/null 2> $OUT/pmc_$c.err; done
python tools/conv_classes_report.py $OUT/plan.json "$(find $OUT/kt -name '*kernel_trace.csv' | head -1)" "$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $OUT/conv_classes.json $OUT/conv_classes.md
head -3 "$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)"
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
