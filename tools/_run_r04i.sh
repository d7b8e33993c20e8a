cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
for c in cfg3 cfg2; do
  dir=$O/raw_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $dir -o p -- python bench.py --config $c --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-diagnostics --no-graphs > $O/full1_$c.out 2> $O/full1_$c.err
  python tools/steady_state.py "$(find $dir -name '*kernel_trace.csv' | head -1)" 8 6 > $O/steady_$c.txt 2>&1
  find $dir -type f -delete
done
cat $O/steady_cfg3.txt; cat $O/steady_cfg2.txt
