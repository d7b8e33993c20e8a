#!/bin/bash
# GPU box: rocprofv3 evidence for pod_wino_conv3x3 on the launch bench.py times (19 runs x 5 levels of the 768x1344 frame):
# kernel-trace stats, then FETCH_SIZE / WRITE_SIZE and the SQ counters in separate --pmc passes.
#   tools/profile_wino.sh <tag>      -> gpurun_out/<tag>/wino_{stats.csv,pmc.txt,events.txt}
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python tools/wino_only.py 20 19 bench 2>&1 | grep wino > "$OUT/wino_events.txt"; cat "$OUT/wino_events.txt"
raw="$OUT/raw_wino"
rocprofv3 --kernel-trace --stats --output-format csv -d "$raw" -o w -- python tools/wino_only.py 20 19 bench > /dev/null 2>&1
s=$(find "$raw" -name '*kernel_stats.csv' | head -1); head -6 "$s" | cut -c1-260 > "$OUT/wino_stats.csv"; cat "$OUT/wino_stats.csv"
find "$raw" -type f -delete
: > "$OUT/wino_pmc.txt"
for counters in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  timeout 600 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$raw" -o p -- python tools/wino_only.py 2 19 bench > /dev/null 2>&1
  f=$(find "$raw" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" >> "$OUT/wino_pmc.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wino_conv3x3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%s %.8g %d" % (k, sum(v) / len(v), len(v)))
PY
  find "$raw" -type f -delete 2>/dev/null
done
cat "$OUT/wino_pmc.txt"
