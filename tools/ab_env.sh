#!/bin/bash
# GPU box: A/B of an environment switch on bench.py, interleaved on one box.  tools/ab_env.sh <VAR> "<configs>" <out> [reps]
set -u
var=$1; out=$3; reps=${4:-2}; mkdir -p $(dirname $out); : > $out
for c in $2; do for r in $(seq $reps); do for v in 0 1; do
  line=$(env $var=$v python bench.py --config $c --steps 150 --warmup 20 --no-cpu-baseline --no-diagnostics 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")
  echo "$c $var=$v  $line" | tee -a $out
done; done; done
