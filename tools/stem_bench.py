"""GPU box: the ResNet stem at the benchmark frame (768 x 1344): pod_stem7x7_split + pod_maxpool3x3s2_cl (channels-last out) against what they
replace (MIOpen's conv2d without bias + pod_bias_act + torch's max_pool2d + the transposing copy to channels-last), replayed as graphs."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pod_compare_amd import hip  # noqa: E402
from pod_compare_amd.conv1x1 import Stem7x7, maxpool3x3s2_cl  # noqa: E402


def graphed(fn, n=10):
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 30 / n


h, w = 768, 1344
wt = torch.randn(64, 3, 7, 7, device="cuda") * 0.1
b = torch.randn(64, device="cuda")
x = torch.randn(1, 3, h, w, device="cuda")
stem, lib = Stem7x7(wt, b), hip.load()
t_conv = graphed(lambda: stem(x))
y, ho, wo = stem(x)
t_pool = graphed(lambda: maxpool3x3s2_cl(y, ho, wo))


def old():
    z = F.conv2d(x, wt, None, stride=2, padding=3)
    hip.check(lib.pod_bias_act(z.data_ptr(), b.data_ptr(), None, None, z.numel(), 64, ho * wo, 1, 0.0, 0, 0, hip.current_stream()), "pod_bias_act")
    z = F.max_pool2d(z, kernel_size=3, stride=2, padding=1)
    return z.permute(0, 2, 3, 1).reshape(-1, 64).contiguous()


t_old = graphed(old)
gf = 2.0 * ho * wo * 64 * 147 / 1e9
print("pod_stem7x7_split %.1f us (%.1f TFLOP/s of the 7x7 convolution's fp32 arithmetic)  pod_maxpool3x3s2_cl %.1f us  | MIOpen conv + pod_bias_act + max_pool2d + copy %.1f us"
      % (t_conv, gf / t_conv * 1e3, t_pool, t_old))
