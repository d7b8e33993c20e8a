cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_wino_conv_gpu.py tests/test_run_image_gpu.py tests/test_native_exact_gpu.py -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python bench.py --no-cnn --streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-diagnostics > $O/hot_planted.json 2>$O/err.txt
timeout 600 python bench.py --no-cnn --streams 1 --steps 100 --warmup 10 --synth worst --no-cpu-baseline --no-diagnostics > $O/hot_worst.json 2>>$O/err.txt
tail -c 400 $O/hot_planted.json; tail -c 400 $O/hot_worst.json
