#!/bin/bash
# EXPERIMENT: does a MIOpen tuning pass (no gfx950 perf db ships with this ROCm) speed up the conv net?
#   tools/miopen_tune.sh <tag> <seconds>
set -u
TAG=${1:-tune}; LIMIT=${2:-1500}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT/db
export MIOPEN_USER_DB_PATH=$OUT/db
echo "== before (empty user db)"; VARIANTS=default timeout 600 python tools/cnn_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== tuning (MIOPEN_FIND_ENFORCE=3, $LIMIT s)"
MIOPEN_FIND_ENFORCE=3 VARIANTS=cudnn.benchmark timeout $LIMIT python tools/cnn_bench.py > $OUT/tune.log 2>&1; echo "tune exit $?"; tail -3 $OUT/tune.log
ls -la $OUT/db | head
echo "== after (tuned user db), immediate mode and find mode"
VARIANTS=default,cudnn.benchmark timeout 900 python tools/cnn_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
