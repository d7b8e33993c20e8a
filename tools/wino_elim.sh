#!/bin/bash
# HERE (build container): elimination builds of pod_wino_conv3x3 -> pod_compare_amd/lib/elim<bits>/libpod_mi355x.so
#   tools/wino_elim.sh build "1 2 4 8 16 32 64 128 15 31 159"
# GPU box: time each on the bench launch
#   tools/wino_elim.sh run "1 2 ..."  -> gpurun_out/<tag>/wino_elim.txt
set -u
cmd=$1; bits=$2; TAG=${3:-r03}
if [ "$cmd" = build ]; then
  for b in $bits; do
    POD_BUILD_TAG=elim$b POD_TAG_SOURCES=k11_wino_conv.hip POD_EXTRA_DEFINES="-DPOD_WINO_ELIM=$b" python -m pod_compare_amd.build > /dev/null || exit 1
  done
else
  mkdir -p gpurun_out/$TAG
  out=gpurun_out/$TAG/wino_elim.txt; : > $out
  echo "elim 0: $(python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
  for b in $bits; do
    echo "elim $b: $(POD_MI355X_LIB=pod_compare_amd/lib/elim$b/libpod_mi355x.so python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
  done
fi
