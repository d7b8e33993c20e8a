cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
for c in cfg2 cfg4 cfg3; do for s in 3 4 6 8; do
python bench.py --config $c --streams $s --steps 120 --warmup 10 --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c streams $s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_image'],3))"
done; done 2>&1 | tee $O/streams.txt
python bench.py --config cfg5 --steps 40 --warmup 5 --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'host', round(d['host_enqueue_ms_per_image'],3))" | tee -a $O/streams.txt
