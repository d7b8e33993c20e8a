#!/bin/bash
# GPU-box check of a build (via gpurun): parity tests, the N=1 bench line, the 2-rank functional check, tail timings.
#   tools/gpu_check.sh <tag> [pytest-args...]
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x "$@" > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -15 $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 40 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench exit $?"
tail -c 1500 $OUT/bench_n1.json; tail -5 $OUT/bench_n1.err
POD_BENCH_BACKEND=gloo POD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-diagnostics > $OUT/bench_gloo2_shared_gpu.json 2> $OUT/bench_gloo2.err; echo "bench2 exit $?"
cat $OUT/bench_gloo2_shared_gpu.json; tail -5 $OUT/bench_gloo2.err
timeout 300 python tools/tail_only.py planted > $OUT/tail_planted.txt 2>&1
timeout 300 python tools/tail_only.py worst > $OUT/tail_worst.txt 2>&1
cat $OUT/tail_planted.txt $OUT/tail_worst.txt
