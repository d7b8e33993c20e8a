"""Diagnostics: phase time stamps of the fused gather + decode kernel (build with POD_TRACE=1 python -m pod_compare_amd.build --force)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A, hip
dev = torch.device("cuda", 0)
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
synth = sys.argv[1] if len(sys.argv) > 1 else "planted"
h = synthetic.planted_head_outputs(padded, 10, seed=1000, num_boxes=24, device=dev, mode=synth)
hp = hotpath.HotPath(h.shapes, h.anchors, hotpath.PathParams(), n_runs=10, has_cls_var=True, cov_dims=4, device=dev)
for i in range(6):
    hp.run("bayes_od", h.cls, h.delta, h.cls_var, h.reg_var, image_size=(750, 1333), out_size=(720, 1280))
torch.cuda.synchronize()
ctypes.CDLL(hip.library_path()).pod_trace_dump()
