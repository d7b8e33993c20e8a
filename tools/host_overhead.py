"""Host-side cost of one hot-path image (Python + ctypes + HIP launches), piece by piece.  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A, hip
N = 10; dev = torch.device("cuda", 0)
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
h = synthetic.planted_head_outputs(padded, N, seed=1000, num_boxes=24, device=dev)
hp = hotpath.HotPath(h.shapes, h.anchors, hotpath.PathParams(), n_runs=N, has_cls_var=True, cov_dims=4, device=dev)
def t(name, fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print("%-34s host %.1f us/call   (incl. device drain %.1f us/call)" % (name, 1e6 * dt / n, 1e6 * t1 / n))
t("_levels()", lambda: hp._levels(h.cls, h.delta, h.cls_var, h.reg_var, None))
t("new_detections()", lambda: hp.new_detections((720, 1280)))
out = hp.new_detections((720, 1280))
t("PodDetections struct", lambda: hip.PodDetections(out.ptr("boxes"), out.ptr("cov"), out.ptr("scores"), out.ptr("classes"), out.ptr("probs"), out.ptr("records"), out.ptr("n_det")))
t("current_stream()", lambda: hip.current_stream())
lv = hp._levels(h.cls, h.delta, h.cls_var, h.reg_var, None)
d = hip.PodDetections(*[out.ptr(n) for n in ("boxes", "cov", "scores", "classes", "probs", "records", "n_det")])
st = hip.current_stream()
t("pod_run_image (C call only)", lambda: hp.lib.pod_run_image(hp.cfg, lv, hp.ws, 1, 0, 0, 750, 1333, 720, 1280, d, st))
t("run(one_call=True)", lambda: hp.run("bayes_od", h.cls, h.delta, h.cls_var, h.reg_var, image_size=(750, 1333), out_size=(720, 1280)))
t("run(one_call=False)", lambda: hp.run("bayes_od", h.cls, h.delta, h.cls_var, h.reg_var, image_size=(750, 1333), out_size=(720, 1280), one_call=False))
