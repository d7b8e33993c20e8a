"""Diagnostics: wall-clock (100 MHz) phase stamps of k2_level_topk on the BASELINE geometry.
    POD_TRACE=1 POD_BUILD_TAG=trace python -m pod_compare_amd.build     (here)
    POD_MI355X_LIB=pod_compare_amd/lib/trace/libpod_mi355x.so python tools/trace_k2.py [planted|worst]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A, hip
dev = torch.device("cuda", 0)
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
synth = sys.argv[1] if len(sys.argv) > 1 else "worst"
h = synthetic.planted_head_outputs(padded, 10, seed=1000, num_boxes=24, device=dev, mode=synth)
hp = hotpath.HotPath(h.shapes, h.anchors, hotpath.PathParams(), n_runs=10, has_cls_var=True, cov_dims=4, device=dev)
for i in range(6):
    hp.run("bayes_od", h.cls, h.delta, h.cls_var, h.reg_var, image_size=(750, 1333), out_size=(720, 1280))
torch.cuda.synchronize()
host = np.zeros((128, 16), dtype=np.int64)
lib = ctypes.CDLL(hip.library_path())
lib.pod_k2_trace_dump.argtypes = [ctypes.c_void_p]
assert lib.pod_k2_trace_dump(host.ctypes.data) == 0
t0 = host[:80, 0][host[:80, 0] > 0].min()
names = {0: "start", 1: "counts", 2: "slice: keys loaded", 3: "slice: passes", 4: "slice: compacted", 6: "ticket", 7: "fence", 8: "final: keys loaded",
         9: "final: passes", 10: "final: compacted", 11: "final: sorted", 12: "written"}
for wg in range(80):
    row = host[wg]
    if row[6] == 0 and row[12] == 0 and wg % 16:
        continue
    print("wg %2d (level %d slice %2d): " % (wg, wg // 16, wg % 16) + "  ".join("%s %.2f" % (names[i], (row[i] - t0) / 100.0) for i in sorted(names) if row[i] >= t0))
