#!/bin/bash
set -u
TAG=${1:-misc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_torch_ops_gpu.py tests/test_ensemble_dist_gpu.py -q 2>&1 | grep -v amdgpu.ids | tail -15
POD_BENCH_BACKEND=gloo POD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 6 --config cfg5 --ensemble-per-gpu --steps 12 --warmup 2 > $OUT/bench_cfg5_gloo6_shared_gpu.json 2> $OUT/bench_cfg5.err; echo "cfg5 exit $?"
cat $OUT/bench_cfg5_gloo6_shared_gpu.json; grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/bench_cfg5.err | tail -5
