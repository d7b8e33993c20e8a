#!/bin/bash
# GPU box: PMC counters of pod_wino_conv3x3 (separate rocprofv3 --pmc passes, kernel-trace only).
#   tools/wino_pmc.sh <tag> "<counter list 1>" "<counter list 2>" ...
set -u
TAG=${1:-wino}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python tools/wino_only.py 5 > "$OUT/plain.txt" 2>&1; grep wino "$OUT/plain.txt"
i=0
for counters in "$@"; do
  i=$((i+1))
  raw="$OUT/raw_$i"
  timeout 600 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$raw" -o p -- python tools/wino_only.py 2 > "$OUT/pmc_$i.log" 2>&1
  f=$(find "$raw" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wino_conv3x3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-40s mean %.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
  else
    echo "no counter file for: $counters"; tail -5 "$OUT/pmc_$i.log"
  fi
  find "$raw" -type f -delete 2>/dev/null
done
