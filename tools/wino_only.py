"""GPU box: a few launches of pod_wino_conv3x3 on the p3 trunk shape (18 runs of 90x160x256 -> 256), for rocprofv3.
    python tools/wino_only.py [launches] [copies]"""
import sys

import torch

sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 18
levels = [(90, 160)]
dev = torch.device("cuda")
torch.manual_seed(0)
conv = WinoConv(torch.randn(256, 256, 3, 3, device=dev) * 0.03, torch.randn(256, device=dev))
src = torch.randn(copies * 90 * 160, 256, device=dev)
dst = torch.empty_like(src)
tab = block_table(levels, copies, dev)
for _ in range(n):
    conv(src, dst, tab, relu=True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n):
    conv(src, dst, tab, relu=True)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
print("wino p3 x%d: %.3f ms/launch = %.1f TFLOP/s direct-equivalent, %.1f executed on the matrix cores" % (
    copies, ms, 2.0 * copies * 14400 * 256 * 256 * 9 / ms / 1e9, 2.0 * copies * 14400 * 256 * 256 * 9 / ms / 1e9 * 16 / 36))
