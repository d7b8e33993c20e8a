#!/bin/bash
# Elimination / variant builds of pod_wino_conv3x3_split (K12) WITH phase stamps, so that shader cycles (not DVFS-confounded wall time)
# price each ingredient.   HERE:  tools/wino_elim12.sh build "0 1 2 4 8 6 15 v8"      (vN: -DPOD_WINO_VAR=N)
# GPU box:  tools/wino_elim12.sh run "0 1 2 ..." <tag>   -> gpurun_out/<tag>/wino_elim12.txt
set -u
cmd=$1; bits=$2; TAG=${3:-r04}
defs() { case $1 in v*) echo "-DPOD_WINO_VAR=${1#v}";; *) echo "-DPOD_WINO_ELIM=$1";; esac; }
if [ "$cmd" = build ]; then
  for b in $bits; do
    POD_TRACE=1 POD_BUILD_TAG=s12e$b POD_TAG_SOURCES=k12_wino_conv_split.hip POD_EXTRA_DEFINES="$(defs $b)" python -m pod_compare_amd.build > /dev/null || exit 1
  done
else
  mkdir -p gpurun_out/$TAG
  out=gpurun_out/$TAG/wino_elim12.txt; : > $out
  for b in $bits; do
    L=pod_compare_amd/lib/s12e$b/libpod_mi355x.so
    echo "== build $b: $(POD_WINO_SPLIT=1 POD_MI355X_LIB=$L python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
    POD_WINO_SPLIT=1 POD_MI355X_LIB=$L python tools/wino_trace.py 19 bench 2>&1 | grep -E "K loop|workgroup total|shader clock|first loads|prologue|store pass|dump" | tee -a $out
  done
fi
