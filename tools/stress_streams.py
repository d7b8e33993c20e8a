"""Stress: the hot path on S streams x many images; every repeat of an input set must reproduce its first result bit for bit
(per-stream workspaces, self-cleaning counters / bitmap, no cross-stream state).  GPU box only.
    python tools/stress_streams.py [streams] [images] [synth]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
synth = sys.argv[3] if len(sys.argv) > 3 else "planted"
dev = torch.device("cuda", 0)
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
heads = [synthetic.planted_head_outputs(padded, 10, seed=1000 + i, num_boxes=24, mode=synth, device=dev) for i in range(3)]
streams = [torch.cuda.Stream() for _ in range(S)]
hps = [hotpath.HotPath(heads[0].shapes, heads[0].anchors, hotpath.PathParams(), n_runs=10, has_cls_var=True, cov_dims=4, device=dev) for _ in range(S)]
torch.cuda.synchronize()
dets = []
for i in range(n):
    h = heads[i % 3]
    mode = ("bayes_od", "standard_nms", "anchor_statistics")[(i // 3) % 3]
    with torch.cuda.stream(streams[i % S]):
        dets.append((i % 3, mode, hps[i % S].run(mode, h.cls, h.delta, h.cls_var, h.reg_var, image_size=(750, 1333), out_size=(720, 1280), draw_id=i % 3)))
torch.cuda.synchronize()
first, bad = {}, 0
for k, mode, d in dets:
    key = (k, mode)
    m = d.count()
    if key not in first:
        first[key] = (m, d.records[:m].clone())
    else:
        m0, r0 = first[key]
        if m != m0 or not torch.equal(d.records[:m], r0):
            bad += 1
print("%d images on %d streams, %d distinct (input, mode) pairs, mismatches: %d" % (n, S, len(first), bad))
sys.exit(1 if bad else 0)
