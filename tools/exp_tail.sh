#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-tail}
mkdir -p $OUT
for synth in planted worst; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$synth -o t -- python tools/tail_only.py $synth > $OUT/tail_$synth.txt 2>&1
  s=$(find $OUT/t_$synth -name '*kernel_stats.csv' | head -1)
  echo "== $synth"; grep "pod::" $s | cut -d, -f1-5 | cut -c1-120
  rm -rf $OUT/t_$synth
done
