#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
run() {  # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o hp -- python bench.py --no-cnn --streams 1 --steps 60 --warmup 5 --no-cpu-baseline --no-diagnostics > $OUT/$name.json 2> $OUT/$name.err
  f=$(find $OUT/$name -name '*kernel_trace.csv' | head -1)
  python tools/trace_timeline.py $f > $OUT/${name}_timeline.txt 2>&1
  echo "== $name"; tail -11 $OUT/${name}_timeline.txt
  rm -rf $OUT/$name
}
run base A=1
run grid384 POD_K23_GRID=384
timeout 600 python -m pytest tests/test_run_image_gpu.py tests/test_hip_parity.py tests/test_native_exact_gpu.py -q -x 2>&1 | tail -5
