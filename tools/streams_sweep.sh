#!/bin/bash
# GPU box: images/s of bench.py against the number of HIP streams per GPU, per config.  tools/streams_sweep.sh "<configs>" "<streams>" <out>
set -u
out=$3; mkdir -p $(dirname $out); : > $out
for c in $1; do for s in $2; do
  v=$(python bench.py --config $c --streams $s --steps 120 --warmup 20 --no-cpu-baseline --no-diagnostics 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")
  echo "$c streams=$s  $v" | tee -a $out
done; done
