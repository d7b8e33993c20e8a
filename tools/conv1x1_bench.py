"""GPU box: pod_conv1x1_split against what it replaces (MIOpen's NCHW conv2d without bias + one pod_bias_act pass) on the 1x1 convolutions
of a ResNet-50-FPN at the benchmark frame (768 x 1344): per shape and weighted by how often a forward runs it.   python tools/conv1x1_bench.py"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from pod_compare_amd import hip  # noqa: E402
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402

sys.path.insert(0, "tools")
from conv1x1_shapes import SHAPES  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


tot_new = tot_old = 0.0
lib = hip.load()
for name, cin, cout, h, w, s, res, calls in SHAPES:
    torch.manual_seed(cin + cout)
    wt = torch.randn(cout, cin, 1, 1, device="cuda") * (2.0 / cin) ** 0.5
    b = torch.randn(cout, device="cuda")
    x = torch.randn(1, cin, h, w, device="cuda").relu()
    xcl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    conv = Conv1x1(wt, b, s)
    ho, wo = conv.out_hw(h, w)
    r_cl = torch.randn(ho * wo, cout, device="cuda") if res else None
    r_nchw = torch.randn(1, cout, ho, wo, device="cuda") if res else None
    g = torch.cuda.CUDAGraph()              # ten calls replayed as one graph: no host time in the figure (the model replays its forward the same way)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        conv(xcl, h, w, relu=True, residual=r_cl)
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                conv(xcl, h, w, relu=True, residual=r_cl)
    t_new = timed(g.replay) / 10

    def old():
        y = F.conv2d(x, wt, None, stride=s)
        hip.check(lib.pod_bias_act(y.data_ptr(), b.data_ptr(), hip.ptr(r_nchw), None, y.numel(), cout, ho * wo, 1, 0.0, 0, 0, hip.current_stream()), "pod_bias_act")
        return y
    t_old = timed(old)
    gf = 2.0 * ho * wo * cin * cout / 1e9
    print("%-20s %4d->%4d %3dx%3d s%d splits %d : K13 %6.1f us (%5.1f TFLOP/s)   MIOpen + bias_act %6.1f us (%5.1f)   x%d" % (
        name, cin, cout, ho, wo, s, conv.splits_for(ho * wo), t_new, gf / t_new * 1e3, t_old, gf / t_old * 1e3, calls))
    tot_new += calls * t_new
    tot_old += calls * t_old
print("per image (39 calls): K13 %.3f ms   MIOpen + bias_act %.3f ms" % (tot_new / 1e3, tot_old / 1e3))
