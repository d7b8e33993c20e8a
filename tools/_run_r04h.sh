cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04h
O=gpurun_out/r04h
for synth in planted worst; do
echo "== $synth two launches"; python tools/k1_only.py 120 4 $synth 2>&1 | grep -v amdgpu
echo "== $synth shipped fused"; K1_FUSED=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep "K1 events"
echo "== $synth shipped fused + planes"; K1_FUSED=1 K1_FUSED_PLANES=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep "K1 events"
for v in wpe2b2 wpe3b2w2 c2wpe3b2 c2wpe2b2 wpe3b2w8; do
echo "== $synth fused $v"; POD_MI355X_LIB=pod_compare_amd/lib/k1f_$v/libpod_mi355x.so K1_FUSED=1 python tools/k1_only.py 120 4 $synth 2>&1 | grep "K1 events"
done
done > $O/k1c.txt 2>&1
cat $O/k1c.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_run_image_gpu.py tests/test_native_exact_gpu.py -q -x 2>&1 | tail -3
