"""Hot path device time against the number of candidates: planted inputs with more and more objects (BASELINE geometry,
N = 10, bayes_od).  HIP events around pod_run_image, 30 images per point.   python tools/sweep_candidates.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A
dev = torch.device("cuda", 0)
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
hp = None
print("| planted boxes | candidates n | detections | us per image |\n|---|---|---|---|")
for boxes, mode in ((24, "planted"), (45, "planted"), (60, "planted"), (90, "planted"), (120, "planted"), (250, "planted"), (500, "planted"), (1000, "planted"), (24, "worst")):
    h = synthetic.planted_head_outputs(padded, 10, seed=77, num_boxes=boxes, device=dev, mode=mode)
    if hp is None:
        hp = hotpath.HotPath(h.shapes, h.anchors, hotpath.PathParams(), n_runs=10, has_cls_var=True, cov_dims=4, device=dev)
    run = lambda: hp.run("bayes_od", h.cls, h.delta, h.cls_var, h.reg_var, image_size=(750, 1333), out_size=(720, 1280))
    for _ in range(5):
        det = run()
    torch.cuda.synchronize()
    n, m = int(hp.n_total.item()), det.count()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    torch.cuda._sleep(2_000_000)
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    us = sorted(1e3 * a.elapsed_time(b) for a, b in ev)
    print("| %d (%s) | %d | %d | %.1f |" % (boxes, mode, n, m, sum(us) / len(us)))
    del h
