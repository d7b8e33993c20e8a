#!/bin/bash
# K1f tuning: tagged builds of k1f_merge_score_fused.hip with other launch geometry, timed by rocprofv3 on tools/k1_only.py.
#   HERE:     tools/k1f_variants.sh build
#   GPU box:  tools/k1f_variants.sh run <tag>  -> gpurun_out/<tag>/k1f_variants.txt
set -u
declare -A V=( [base]="" [c2]="-DPOD_K1F_CELLS=2" [c2b4]="-DPOD_K1F_CELLS=2 -DPOD_K1F_BATCH=4" [b3]="-DPOD_K1F_BATCH=3" [b4]="-DPOD_K1F_BATCH=4" [b4w2]="-DPOD_K1F_BATCH=4 -DPOD_K1F_WPE=2" \
               [noscore]="-DPOD_K1F_NOSCORE" [w2]="-DPOD_K1F_WAVES=2" [w1]="-DPOD_K1F_WAVES=1" [w8]="-DPOD_K1F_WAVES=8" [b4noscore]="-DPOD_K1F_BATCH=4 -DPOD_K1F_NOSCORE" [c2noscore]="-DPOD_K1F_CELLS=2 -DPOD_K1F_NOSCORE")
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do
    POD_BUILD_TAG=k1f_$k POD_TAG_SOURCES=k1f_merge_score_fused.hip POD_EXTRA_DEFINES="${V[$k]} -DPOD_K1F_VARIANT" python -m pod_compare_amd.build > /dev/null || echo "build $k failed"
  done
else
  TAG=${2:-k1f}; mkdir -p gpurun_out/$TAG; out=gpurun_out/$TAG/k1f_variants.txt; : > $out
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  for k in base noscore b3 b4 b4w2 b4noscore c2 c2b4 c2noscore w1 w2 w8 ${EXTRA_KEYS:-}; do
    L=pod_compare_amd/lib/k1f_$k/libpod_mi355x.so
    [ -f $L ] || continue
    d=gpurun_out/$TAG/raw_$k
    K1_FUSED=1 POD_MI355X_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python tools/k1_only.py 120 > /dev/null 2>&1
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    echo "$k: $(grep k1f_merge_score $f | head -1 | awk -F, '{print "calls",$2," avg_ns",$4," min",$6," max",$7}')" | tee -a $out
    rm -rf $d
  done
fi
