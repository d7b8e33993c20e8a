#!/bin/bash
# GPU-box run of the parity tests only:  tools/gpu_pytest.sh <tag> [pytest-args...]
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -q "$@" > gpurun_out/$TAG/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/$TAG/pytest_gpu.txt
tail -60 gpurun_out/$TAG/pytest_gpu.txt
