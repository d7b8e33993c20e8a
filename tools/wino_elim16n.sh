#!/bin/bash
# GPU box: wall time of the bench launch for the NON-traced elimination builds of K16 (tags s16n<bits>; built here with
#   POD_BUILD_TAG=s16n$b POD_WITH_K16=1 POD_TAG_SOURCES="k16_wino_conv_split8.hip k12_wino_conv_split.hip" POD_EXTRA_DEFINES=-DW8_ELIM=$b python -m pod_compare_amd.build)
out=gpurun_out/${2:-r06}/wino_elim16n.txt; mkdir -p $(dirname $out); : > $out
echo "shipped form 4: $(POD_WINO_FORM=4 python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
echo "shipped form 8: $(POD_WINO_FORM=8 python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
for b in $1; do
  echo "elim $b form 8: $(POD_WINO_FORM=8 POD_MI355X_LIB=pod_compare_amd/lib/s16n$b/libpod_mi355x.so python tools/wino_only.py 20 19 bench 2>&1 | grep wino)" | tee -a $out
done
