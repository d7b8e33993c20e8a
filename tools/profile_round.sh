#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of bench.py + separate PMC passes for K1.
#   tools/profile_round.sh r01     -> gpurun_out/prof_<tag>/...   (copy the summaries into profiles/ afterwards)
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# (1) per-kernel time of the same command the bench line comes from (hot path only, so the trace stays small)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hotpath -o hp -- python bench.py --no-cnn --steps 40 --no-cpu-baseline > $OUT/bench_hotpath.json 2> $OUT/bench_hotpath.err
# (2) the headline command (conv net in the timed region)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -o full -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err
# (2b) the same with one stream: per-kernel durations are only meaningful when kernels of different images do not overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full1 -o full1 -- python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline > $OUT/bench_full_streams1.json 2> $OUT/bench_full_streams1.err
python tools/steady_state.py $OUT/full1/full1_kernel_trace.csv 8 6 > $OUT/steady_state.txt 2>&1
# (3) HBM traffic of K1: separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python tools/k1_only.py 12 > /dev/null 2>&1
  K1_DENSE=1 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_dense_$c -o p -- python tools/k1_only.py 12 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ -o p -- python tools/k1_only.py 12 > /dev/null 2>&1
python tools/k1_only.py 60 > $OUT/k1_events.txt 2>&1
K1_DENSE=1 python tools/k1_only.py 60 >> $OUT/k1_events.txt 2>&1
ls -R $OUT | head -40
