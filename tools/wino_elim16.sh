#!/bin/bash
# Elimination builds of the eight-wavefront form of pod_wino_conv3x3_split (K16) WITH phase stamps (tools/wino_elim12.sh is K12's).
# HERE:  tools/wino_elim16.sh build "0 1 2 4 8 6 15"      GPU box:  tools/wino_elim16.sh run "0 1 2 ..." <tag>  -> gpurun_out/<tag>/wino_elim16.txt
set -u
cmd=$1; bits=$2; TAG=${3:-r06}
if [ "$cmd" = build ]; then
  for b in $bits; do
    POD_TRACE=1 POD_BUILD_TAG=s16e$b POD_WITH_K16=1 POD_TAG_SOURCES="k16_wino_conv_split8.hip k12_wino_conv_split.hip" POD_EXTRA_DEFINES="-DW8_ELIM=$b ${W8_DEFS:-}" python -m pod_compare_amd.build > /dev/null || exit 1
  done
else
  mkdir -p gpurun_out/$TAG
  out=gpurun_out/$TAG/wino_elim16.txt; : > $out
  for b in $bits; do
    L=pod_compare_amd/lib/s16e$b/libpod_mi355x.so
    for form in ${FORMS:-8}; do
      echo "== build $b form $form: $(POD_WINO_FORM=$form POD_WINO_SPLIT=1 POD_MI355X_LIB=$L python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
      POD_WINO_FORM=$form POD_WINO_SPLIT=1 POD_MI355X_LIB=$L python tools/wino_trace.py 19 bench 2>&1 | grep -E "K loop|workgroup total|shader clock|first loads|prologue|store pass|dump" | tee -a $out
    done
  done
fi
