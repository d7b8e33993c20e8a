"""Does capturing the hot-path chain (K1..K7) in a HIP graph pay?  Eager ctypes launches vs graph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import hotpath, synthetic, anchors as A
dev = torch.device("cuda", 0); N = 10
padded = A.padded_size(*A.resize_shortest_edge(720, 1280))
heads = [synthetic.planted_head_outputs(padded, N, seed=1000 + i, num_boxes=24, device=dev) for i in range(4)]
hp = hotpath.HotPath(heads[0].shapes, heads[0].anchors, hotpath.PathParams(), n_runs=N, has_cls_var=True, cov_dims=4, device=dev)
def chain(h):
    lv = hp.candidates(h.cls, h.delta, h.cls_var, h.reg_var, None)
    hp.decode(lv, None)
    return hp.postprocess("bayes_od", (750, 1333), (720, 1280))
for h in heads: chain(h)
torch.cuda.synchronize()
def bench(fn, tag, n=200):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    print("%-28s %7.1f us/image" % (tag, (time.perf_counter() - t) / n * 1e6))
bench(lambda i: chain(heads[i % 4]), "eager")
graphs = []
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for h in heads:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            det = chain(h)
        graphs.append((g, det))
torch.cuda.synchronize()
bench(lambda i: graphs[i % 4][0].replay(), "graph replay")
print("detections", [int(d.n_det) for _, d in graphs])
