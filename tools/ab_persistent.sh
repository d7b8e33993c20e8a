cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab
for mode in 1 0; do
  for s in 1 3; do
    POD_WINO_PERSISTENT=$mode python bench.py --steps 100 --warmup 15 --streams $s --no-cpu-baseline --no-diagnostics | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persistent=$mode streams=$s', round(d['value'],2), round(d['ms_per_step'],3))"
  done
done
for mode in 1 0; do
  POD_WINO_PERSISTENT=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/raw$mode -o t -- python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-diagnostics > /dev/null 2>&1
  f=$(find gpurun_out/ab/raw$mode -name '*kernel_trace.csv' | head -1)
  echo "persistent=$mode"; python tools/steady_state.py "$f" 8 6 | head -6
  rm -rf gpurun_out/ab/raw$mode
done
