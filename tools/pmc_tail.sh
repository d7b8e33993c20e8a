#!/bin/bash
# GPU box: where does the time of the small tail kernels go?  rocprofv3 --pmc passes over the hot path (planted inputs):
# wavefronts / wave cycles / VALU + SALU instruction counts / instruction-cache traffic per kernel.  Counter passes only
# (no trace domains in the same run).     tools/pmc_tail.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
CMD="python bench.py --no-cnn --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --no-diagnostics"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_INSTS_LDS"; do
  i=$((i+1))
  raw="gpurun_out/${TAG}/raw_p$i"
  rocprofv3 --pmc $set --output-format csv -d "$raw" -o p -- $CMD > /dev/null 2> "$OUT/p$i.err"
  f=$(find "$raw" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY' | tee -a "$OUT/pmc_tail.txt"
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
    tail -3 "$OUT/p$i.err"
  fi
  find "$raw" -type f -delete
done
