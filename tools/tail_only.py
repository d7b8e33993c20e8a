"""Times the tail kernels (K2..K7) separately with HIP events on BASELINE-size planted inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A, hip
N = 10; dev = torch.device("cuda", 0)
synth = sys.argv[1] if len(sys.argv) > 1 else "planted"
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
h = synthetic.planted_head_outputs(padded, N, seed=1000, num_boxes=24, mode=synth, device=dev)
hp = hotpath.HotPath(h.shapes, h.anchors, hotpath.PathParams(), n_runs=N, has_cls_var=True, cov_dims=4, device=dev)
P, st, lib, cfg = hip.ptr, hip.current_stream(), hp.lib, hp.cfg
lv = hp.candidates(h.cls, h.delta, h.cls_var, h.reg_var, None)
hp.decode(lv, None); hp.nms(); torch.cuda.synchronize()
print("n =", int(hp.n_total), "keep =", int(hp.n_keep))
def timeit(name, fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda._sleep(2_000_000)
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("%-22s avg %.2f us  min %.2f" % (name, 1e3 * sum(ms) / len(ms), 1e3 * ms[0]))
# K2 consumes (zeroes) the per-level counts: rebuild the candidate lists first, then put the counts back before each launch
hip.check(lib.pod_mc_merge_score(cfg, lv, P(hp.mean_cls), P(hp.mean_cls_var), None, None, P(hp.cand_keys), P(hp.cand_count), P(hp.maybe_bits), st), "k1")
hip.check(lib.pod_score_maybe(cfg, lv, P(hp.mean_cls), P(hp.mean_cls_var), P(hp.maybe_bits), P(hp.cand_keys), P(hp.cand_count), P(hp.probs_dense), st), "k1b")
saved_counts = hp.counters.clone()
saved_keys = hp.cand_keys.clone()          # a big level's slices are compacted in place
def restore():
    hp.counters.copy_(saved_counts)
    hp.cand_keys.copy_(saved_keys)
timeit("restore (2 copies)", restore)
def k2():
    restore()
    lib.pod_level_topk(cfg, lv, P(hp.cand_keys), P(hp.cand_count), P(hp.sel_keys), P(hp.sel_count), P(hp.cat_keys), P(hp.cat_level), P(hp.n_total), st)
timeit("K2 topk + restore", k2)
timeit("K2b gather", lambda: lib.pod_gather_candidates(cfg, lv, P(hp.anchors), P(hp.cat_keys), P(hp.cat_level), P(hp.n_total), P(hp.cand_count), P(hp.probs_dense), P(hp.cand_anchor_idx), P(hp.cand_level), P(hp.cand_score), P(hp.cand_class), P(hp.cand_probs), P(hp.cand_delta), P(hp.cand_reg_var), P(hp.cand_anchor), P(hp.cand_run_delta), st))
timeit("K3 decode_cov", lambda: hp.decode(lv, None))
timeit("K4 nms", lambda: hp.nms())
timeit("K5+K7 bayes_od", lambda: hp.postprocess("bayes_od", (750, 1333), (720, 1280)))
timeit("K7 std", lambda: hp.finalize(hp.keep, hp.n_keep, hp.boxes, hp.cov, hp.cand_score, hp.cand_class, hp.cand_probs, (750, 1333), (720, 1280)))
