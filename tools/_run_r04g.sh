cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
for synth in planted worst; do
echo "== $synth two launches"; python tools/k1_only.py 120 4 $synth 2>&1 | grep -v amdgpu
echo "== $synth fused W=2"; K1_FUSED=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep -v amdgpu
echo "== $synth fused W=2 + planes"; K1_FUSED=1 K1_FUSED_PLANES=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep -v amdgpu
echo "== $synth fused W=4"; POD_MI355X_LIB=pod_compare_amd/lib/k1fw4/libpod_mi355x.so K1_FUSED=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep -v amdgpu
echo "== $synth fused W=1"; POD_MI355X_LIB=pod_compare_amd/lib/k1fw1/libpod_mi355x.so K1_FUSED=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep -v amdgpu
done > $O/k1.txt 2>&1
cat $O/k1.txt
bash tools/profile_hotpath.sh r04g > $O/prof.txt 2>&1; tail -60 $O/prof.txt
