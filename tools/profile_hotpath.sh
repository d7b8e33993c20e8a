#!/bin/bash
# GPU box: rocprofv3 kernel trace of the hot path alone (one stream), planted and worst-case inputs; one-image timelines.
#   tools/profile_hotpath.sh <tag>      -> gpurun_out/<tag>/hot_{planted,worst}_{stats.csv,timeline.txt}
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for synth in planted busy worst; do
  extra=""; mode=$synth
  if [ "$synth" = "busy" ]; then mode=planted; extra="--boxes 120"; fi
  raw="gpurun_out/${TAG}/raw_hot_${synth}"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$raw" -o hp -- python bench.py --no-cnn --streams 1 --steps 60 --warmup 5 --synth $mode $extra --no-cpu-baseline --no-diagnostics > "$OUT/bench_hot_$synth.json" 2> "$OUT/bench_hot_$synth.err"
  f=$(find "$raw" -name '*kernel_trace.csv' | head -1)
  python tools/trace_timeline.py "$f" > "$OUT/hot_${synth}_timeline.txt" 2>&1
  s=$(find "$raw" -name '*kernel_stats.csv' | head -1)
  head -20 "$s" | cut -c1-200 > "$OUT/hot_${synth}_stats.csv"
  echo "== $synth"; tail -16 "$OUT/hot_${synth}_timeline.txt"; cut -c1-300 "$OUT/bench_hot_$synth.json"
  find "$raw" -type f -delete
done
