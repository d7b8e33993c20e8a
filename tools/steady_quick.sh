#!/bin/bash
# GPU box: steady-state per-kernel table of one config on one stream (rocprofv3 kernel trace).  tools/steady_quick.sh <tag> [bench args...]
set -u
TAG=${1:-q}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
dir="$OUT/raw"
rocprofv3 --kernel-trace --stats --output-format csv -d "$dir" -o p -- python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-diagnostics --no-graphs "$@" > "$OUT/full1.out" 2> "$OUT/full1.err"
cp "$(find "$dir" -name '*kernel_trace.csv' | head -1)" "$OUT/full1_kernel_trace.csv"
python tools/steady_state.py "$OUT/full1_kernel_trace.csv" 8 6 > "$OUT/steady_state.txt" 2>&1
python tools/k12_launches.py "$OUT/full1_kernel_trace.csv" 6 > "$OUT/k12_launches.txt" 2>&1
rm -rf "$dir" "$OUT/full1_kernel_trace.csv"
head -40 "$OUT/steady_state.txt"
