#!/bin/bash
# EXPERIMENT: where does the time of the small tail kernels go?  PMC passes over the hot path (planted inputs).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc}
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -E "ICACHE|IFETCH|SQ_WAIT_INST|SQ_INSTS_VALU |SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INST_CYCLES" $OUT/counters.txt | head -40
CMD="python bench.py --no-cnn --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-diagnostics"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- $CMD > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "pod::" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  else
    tail -3 $OUT/p$i.err
  fi
  rm -rf $OUT/p$i
done
