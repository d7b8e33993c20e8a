#!/bin/bash
# GPU box: full -m gpu suite, then hot-path timelines (planted + worst).   tools/exp_round.sh <tag>
set -u
TAG=${1:-exp}
bash tools/gpu_pytest.sh $TAG -x | tail -15
bash tools/profile_hotpath.sh $TAG
