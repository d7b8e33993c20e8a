#!/bin/bash
# GPU box: sample rocm-smi (power, shader clock, temperature) while a kernel-only loop runs.  tools/power_probe.sh "<command>" <out>
# e.g. tools/power_probe.sh "POD_WINO_SPLIT=1 python tools/wino_only.py 3000 19 bench" gpurun_out/r05d/power_k12.txt
set -u
cmd=$1; out=$2; mkdir -p $(dirname $out); : > $out
echo "# idle" >> $out; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -E "Power|sclk|mclk|fclk|Temp|cap" >> $out
rocm-smi --showmaxpower 2>&1 | grep -i -E "max|cap" >> $out
( eval "$cmd" > $out.cmd 2>&1 ) &
pid=$!
sleep 12                                   # (import + warm-up)
for i in 1 2 3 4 5 6; do
  echo "# sample $i" >> $out
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" >> $out
  sleep 1
done
wait $pid
tail -2 $out.cmd >> $out; rm -f $out.cmd
