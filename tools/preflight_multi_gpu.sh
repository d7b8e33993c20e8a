#!/bin/bash
# Preflight for a MULTI-GPU node: run this once on any box with N >= 2 MI355X before the driver's scaling run
# (`bench.py --gpus 1 / 2 / 4 / 8`).  No GPU node was reachable from the build environment in rounds 1-6: this script is what checks, on first
# contact, everything the multi-rank path assumes.      tools/preflight_multi_gpu.sh [N]        (N: default = every visible GPU)
#
#   1. tests/test_multi_gpu.py           RCCL's first contact, image-sharded bench / apply_net on two nccl ranks, config 5 over RCCL p2p (>= 6 GPUs)
#   2. one 20-step `bench.py --gpus N` line per BASELINE config, launched the way the driver launches it (torch.distributed.run, 127.0.0.1);
#      cfg5 in its one-member-per-GPU topology when N >= 6.  Checked per line: rccl_ranks == N, collective_backend == nccl, N distinct
#      rank_devices, ranks_share_one_gpu false, flush_ms present, value > 0, and the host binding of rank 0 (config.host_binding)
#   3. the N = 1 line through the launcher against the plain `python bench.py` line: within 3 %
# Output: gpurun_out/preflight/*.json + a verdict per check on stdout; exit code 0 only if every check holds.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH="$PWD"
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
N=${1:-$NGPU}
OUT=gpurun_out/preflight; mkdir -p $OUT
fail=0
say() { echo "[preflight] $*"; }
# PREFLIGHT_DRY=1: the script's own plumbing on a ONE-GPU box (N = 1, every line through the launcher with POD_BENCH_FORCE_DIST=1: RCCL with one rank)
if [ "${PREFLIGHT_DRY:-0}" = 1 ]; then N=1; export POD_BENCH_FORCE_DIST=1; say "DRY RUN on one GPU: checks the script, not the node";
elif [ "$NGPU" -lt 2 ] || [ "$N" -lt 2 ] || [ "$N" -gt "$NGPU" ]; then say "needs >= 2 visible GPUs and N <= their number (visible: $NGPU, asked: $N): nothing to check"; exit 2; fi
say "$NGPU GPUs visible, checking N = $N"
python - <<'PY'
import torch
from pod_compare_amd import hostbind
for i in range(torch.cuda.device_count()):
    b = hostbind.bind_rank_to_gpu_numa(i, enable=False)
    print("[preflight] cuda:%d pci %s numa node %s cpus %s %s" % (i, b["pci"], b["numa_node"], b["cpus"], b.get("why_not", "")))
PY
say "1. tests/test_multi_gpu.py"
python -m pytest tests/test_multi_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $OUT/test_multi_gpu.txt
grep -q " passed" $OUT/test_multi_gpu.txt && ! grep -q "failed\|error" $OUT/test_multi_gpu.txt || { say "FAIL: tests/test_multi_gpu.py"; fail=1; }
line() {   # name, nproc, bench args... -> $OUT/<name>.json
  local name=$1 np=$2; shift 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $np "$@" \
    2> $OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json
}
check() {  # name, expected ranks
  python - "$OUT/$1.json" "$2" <<'PY' || fail=1
import json, sys
path, n = sys.argv[1], int(sys.argv[2])
try:
    d = json.load(open(path))
except Exception as e:
    print("[preflight] FAIL %s: no JSON line (%s)" % (path, e)); sys.exit(1)
c, bad = d["config"], []
if d["n_gpus"] != n or c.get("rccl_ranks") != n: bad.append("rccl_ranks %s != %d" % (c.get("rccl_ranks"), n))
if n > 1 and c.get("collective_backend") != "nccl": bad.append("backend %s" % c.get("collective_backend"))
if c.get("ranks_share_one_gpu"): bad.append("ranks share a GPU")
rd = c.get("rank_devices")
if rd is not None and len(set(rd)) != n: bad.append("rank_devices %s not distinct" % rd)
if n > 1 and "flush_ms" in d and d["flush_ms"] is None: bad.append("no flush_ms")
if not d["value"] > 0: bad.append("value %s" % d["value"])
print("[preflight] %s %s: %.1f %s on %d GPU(s), %.2f ms/step, flush %s ms, host binding of rank 0: %s%s" % (
    "FAIL" if bad else "ok  ", path, d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d.get("flush_ms"), c.get("host_binding"), ("  <- " + "; ".join(bad)) if bad else ""))
sys.exit(1 if bad else 0)
PY
}
say "2. one 20-step line per BASELINE config on $N GPUs"
for cfg in cfg2 cfg3 cfg4; do
  [ "${PREFLIGHT_DRY:-0}" = 1 ] && [ $cfg != cfg3 ] && continue
  line ${cfg}_n$N $N --steps 20 --warmup 5 --config $cfg --no-cpu-baseline --no-diagnostics; check ${cfg}_n$N $N
done
if [ "${PREFLIGHT_DRY:-0}" = 1 ]; then :
elif [ "$N" -ge 6 ]; then
  line cfg5_per_gpu_n$N $N --steps 12 --warmup 2 --config cfg5 --ensemble-per-gpu; check cfg5_per_gpu_n$N $N
else
  line cfg5_n$N $N --steps 20 --warmup 5 --config cfg5 --no-cpu-baseline --no-diagnostics; check cfg5_n$N $N     # five members per GPU, images sharded
fi
say "3. N = 1 through the launcher against the plain line (3 %)"
export POD_BENCH_FORCE_DIST=1; line cfg3_n1_launcher 1 --steps 40 --warmup 10 --no-cpu-baseline --no-diagnostics; check cfg3_n1_launcher 1
unset POD_BENCH_FORCE_DIST
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-diagnostics 2> $OUT/cfg3_n1_plain.err | grep '^{' | tail -1 > $OUT/cfg3_n1_plain.json; check cfg3_n1_plain 1
python - $OUT/cfg3_n1_launcher.json $OUT/cfg3_n1_plain.json $OUT/cfg3_n$N.json $N <<'PY' || fail=1
import json, sys
a, b, c, n = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), json.load(open(sys.argv[3])), int(sys.argv[4])
r = a["value"] / b["value"]
print("[preflight] N = 1: launcher %.1f vs plain %.1f images/s (ratio %.3f); N = %d: %.1f = %.2f x the plain line (weak scaling: ideal %d)" % (
    a["value"], b["value"], r, n, c["value"], c["value"] / b["value"], n))
sys.exit(0 if abs(r - 1.0) <= 0.03 else 1)
PY
[ $fail = 0 ] && say "ALL CHECKS HOLD" || say "SOME CHECKS FAILED (see above; lines and stderr under $OUT/)"
exit $fail
