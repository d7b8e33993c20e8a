"""Runs only K1 (pod_mc_merge_score) on BASELINE-size planted inputs: for rocprofv3 counter passes.
    python tools/k1_only.py [iters] [images] [synth]        K1_DENSE=1: also merge box_delta / box_reg_var densely"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A
from pod_compare_amd.hip import ptr as P, current_stream, check

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 4
synth = sys.argv[3] if len(sys.argv) > 3 else "planted"
N = int(os.environ.get("K1_RUNS", "10"))
dev = torch.device("cuda", 0)
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
kw = {}
if os.environ.get("K1_SINGLE", "0") == "1":      # one level with the same number of anchors (diagnostic: cost of the level structure)
    padded, kw = (1024, 1344), dict(strides=(8,), sizes=(A.ANCHOR_SIZES[0],))
heads = [synthetic.planted_head_outputs(padded, N, seed=1000 + i, num_boxes=24, mode=synth, device=dev, **kw) for i in range(n_img)]
hp = hotpath.HotPath(heads[0].shapes, heads[0].anchors, hotpath.PathParams(), n_runs=N, has_cls_var=True, cov_dims=4, device=dev,
                     dense_box_merge=os.environ.get("K1_DENSE", "0") == "1")
lvs = [hp._levels(h.cls, h.delta, h.cls_var, h.reg_var, None) for h in heads]
st = current_stream()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
prune = os.environ.get("K1_PRUNE", "1") == "1"
evb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
fused = os.environ.get("K1_FUSED", "0") == "1"      # pod_merge_score_fused (K1f) instead of K1 + K1b; K1_FUSED_PLANES=1: also store the merged planes
def launch(j):
    hp.lib.pod_reset_counters(P(hp.counters), 8, st)
    if j >= 0: ev[j][0].record()
    if fused:
        pl = os.environ.get("K1_FUSED_PLANES", "0") == "1"
        check(hp.lib.pod_merge_score_fused(hp.cfg, lvs[j % n_img], P(hp.mean_cls) if pl else None, P(hp.mean_cls_var) if pl else None,
                                           P(hp.cand_keys), P(hp.cand_count), P(hp.probs_dense), st), "k1f")
        if j >= 0: ev[j][1].record()
        return
    check(hp.lib.pod_mc_merge_score(hp.cfg, lvs[j % n_img], P(hp.mean_cls), P(hp.mean_cls_var), P(hp.mean_delta),
                                    P(hp.mean_reg_var), P(hp.cand_keys), P(hp.cand_count), P(hp.maybe_bits) if prune else None, st), "k1")
    if j >= 0: ev[j][1].record()
    if prune:
        if j >= 0: evb[j][0].record()
        check(hp.lib.pod_score_maybe(hp.cfg, lvs[j % n_img], P(hp.mean_cls), P(hp.mean_cls_var), P(hp.maybe_bits),
                                     P(hp.cand_keys), P(hp.cand_count), P(hp.probs_dense), st), "k1b")
        if j >= 0: evb[j][1].record()
for j in range(3): launch(-1)
torch.cuda.synchronize()
torch.cuda._sleep(3_000_000)       # let the host run ahead so event pairs are back-to-back on the device
for j in range(iters): launch(j)
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print("K1 events: avg %.2f us  min %.2f  med %.2f  max %.2f  (R=%d N=%d, %s, %s)" % (1e3 * sum(ms) / len(ms), 1e3 * ms[0], 1e3 * ms[len(ms) // 2], 1e3 * ms[-1], hp.R, N, synth, "all channels" if hp.dense_box_merge else "class channels"))
if prune and not fused:
    mb = sorted(a.elapsed_time(b) for a, b in evb)
    print("K1b events: avg %.2f us  min %.2f" % (1e3 * sum(mb) / len(mb), 1e3 * mb[0]))
print("cand counts", hp.counters[:5].tolist(), "maybe anchors", int(sum(bin(int(x) & (2**64 - 1)).count("1") for x in hp.maybe_bits.tolist())) if prune and not fused else 0)
