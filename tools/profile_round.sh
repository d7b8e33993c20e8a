#!/bin/bash
# Runs on the GPU box (via gpurun): the round's rocprofv3 evidence.
#   tools/profile_round.sh r02     -> gpurun_out/prof_<tag>/...   (then: python tools/summarize_profile.py <tag> -> profiles/<tag>_*)
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
stats() {   # name, command...: rocprofv3 --kernel-trace --stats, keep the stats csv (+ the trace when KEEP_TRACE=1)
  local name=$1; shift
  local dir="gpurun_out/prof_${TAG}/raw_${name}"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$dir" -o p -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
  cp "$(find "$dir" -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv" 2>/dev/null
  if [ "${KEEP_TRACE:-0}" = "1" ]; then cp "$(find "$dir" -name '*kernel_trace.csv' | head -1)" "$OUT/${name}_kernel_trace.csv" 2>/dev/null; fi
  find "$dir" -type f -delete
}
# (1) K1 alone, ONE variant per run (>= 100 launches each): the table row whose average reproduces `roofline` by itself
stats k1_class python tools/k1_only.py 120
K1_DENSE=1 stats k1_dense python tools/k1_only.py 120
K1_FUSED=1 stats k1_fused python tools/k1_only.py 120          # the product launch since round 4: merge + score in one kernel
python tools/k1_only.py 120 > "$OUT/k1_events.txt" 2>&1
K1_DENSE=1 python tools/k1_only.py 120 >> "$OUT/k1_events.txt" 2>&1
K1_FUSED=1 python tools/k1_only.py 120 >> "$OUT/k1_events.txt" 2>&1
K1_FUSED=1 K1_FUSED_PLANES=1 python tools/k1_only.py 120 >> "$OUT/k1_events.txt" 2>&1
# (2) HBM traffic of K1: separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), no trace domains
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- python tools/k1_only.py 12 > /dev/null 2>&1
  K1_DENSE=1 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_dense_$c" -o p -- python tools/k1_only.py 12 > /dev/null 2>&1
  K1_FUSED=1 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_fused_$c" -o p -- python tools/k1_only.py 12 > /dev/null 2>&1
done
# (3) the hot path alone, one stream: per-kernel stats + one-image timelines, planted and worst-case inputs
for synth in planted worst; do
  KEEP_TRACE=1 stats hot_$synth python bench.py --no-cnn --streams 1 --steps 100 --warmup 10 --synth $synth --no-cpu-baseline --no-diagnostics
  python tools/trace_timeline.py "$OUT/hot_${synth}_kernel_trace.csv" > "$OUT/hot_${synth}_timeline.txt" 2>&1
  rm -f "$OUT/hot_${synth}_kernel_trace.csv"
done
# (4) the headline command on one stream (per-kernel durations are only meaningful when images do not overlap): steady state
KEEP_TRACE=1 stats full1 python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-diagnostics --no-graphs
python tools/steady_state.py "$OUT/full1_kernel_trace.csv" 8 6 > "$OUT/steady_state.txt" 2>&1
rm -f "$OUT/full1_kernel_trace.csv"
# (5) bench lines: the default command (200 timed steps after 20 warm-up), the driver's shape, the other single-GPU configs,
#     the 2-rank functional check and config 5 with one member per rank (all ranks on this one GPU, gloo)
python bench.py > "$OUT/bench_default_200steps.json" 2> "$OUT/bench_default.err"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_20steps.json" 2>> "$OUT/bench_default.err"
python bench.py --fp32-mfma --steps 60 --warmup 10 --no-cpu-baseline --no-diagnostics > "$OUT/bench_fp32_mfma.json" 2>> "$OUT/bench_default.err"
for c in cfg2 cfg4 cfg5; do python bench.py --config $c --steps 120 --warmup 10 --no-cpu-baseline > "$OUT/bench_$c.json" 2>> "$OUT/bench_default.err"; done
for c in cfg2 cfg4; do python bench.py --config $c --steps 120 --warmup 10 --no-cpu-baseline --no-diagnostics --no-graphs > "$OUT/bench_${c}_no_graphs.json" 2>> "$OUT/bench_default.err"; done
POD_BENCH_BACKEND=gloo POD_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-diagnostics > "$OUT/bench_gloo2_shared_gpu.json" 2>> "$OUT/bench_default.err"
POD_BENCH_BACKEND=gloo POD_BENCH_SHARE_GPU=1 python bench.py --gpus 6 --config cfg5 --ensemble-per-gpu --steps 12 --warmup 2 > "$OUT/bench_cfg5_gloo6_shared_gpu.json" 2>> "$OUT/bench_default.err"
# (6) round 5: the sparse bbox tower as the step (planted / worst), one-stream steady state of it, the trunk's convolution classes
python bench.py --sparse-bbox --steps 60 --warmup 10 --no-cpu-baseline --no-diagnostics > "$OUT/bench_sparse_planted.json" 2>> "$OUT/bench_default.err"
python bench.py --sparse-bbox --synth worst --steps 60 --warmup 10 --no-cpu-baseline --no-diagnostics > "$OUT/bench_sparse_worst.json" 2>> "$OUT/bench_default.err"
for st in 1 3; do python bench.py --streams $st --steps 60 --warmup 10 --no-cpu-baseline --no-diagnostics > "$OUT/bench_streams$st.json" 2>> "$OUT/bench_default.err"; done
ls "$OUT"
