#!/bin/bash
# GPU box, end of a round: the -m gpu suite at the final tree (log kept), then the round's profile set.   tools/gpu_final.sh <tag>
set -u
TAG=${1:-r05}
bash tools/gpu_pytest.sh $TAG -x | tail -6
mkdir -p gpurun_out/prof_$TAG; bash tools/profile_round.sh $TAG > gpurun_out/prof_$TAG/round.log 2>&1
POD_WINO_SPLIT=1 bash tools/profile_wino.sh ${TAG}ws > /dev/null 2>&1
POD_WINO_SPLIT=0 bash tools/profile_wino.sh ${TAG}w > /dev/null 2>&1
bash tools/conv_classes.sh ${TAG}cc > /dev/null 2>&1
bash tools/steady_quick.sh ${TAG}sq_sparse --sparse-bbox > /dev/null 2>&1
bash tools/steady_quick.sh ${TAG}sq_cfg2 --config cfg2 > /dev/null 2>&1
tail -3 gpurun_out/prof_$TAG/round.log
