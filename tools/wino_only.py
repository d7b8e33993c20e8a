"""GPU box: a few launches of pod_wino_conv3x3 for rocprofv3 / HIP-event timing.
    python tools/wino_only.py [launches] [copies] [p3|bench]
p3: the 90x160 map alone; bench: the five FPN levels of the bench frame (768x1344 padded), the launch bench.py times."""
import sys

import torch

sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 18
mode = sys.argv[3] if len(sys.argv) > 3 else "p3"
levels = [(90, 160)] if mode == "p3" else [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]
dev = torch.device("cuda")
torch.manual_seed(0)
conv = WinoConv(torch.randn(256, 256, 3, 3, device=dev) * 0.03, torch.randn(256, device=dev))
tab = block_table(levels, copies, dev)
src = torch.randn(tab.pod_pixels, 256, device=dev)
dst = torch.empty_like(src)
for _ in range(n):
    conv(src, dst, tab, relu=True, dropout_p=float(__import__("os").environ.get("WP", "0.1")), seed=1)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n):
    conv(src, dst, tab, relu=True, dropout_p=float(__import__("os").environ.get("WP", "0.1")), seed=1)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
direct = 2.0 * 9 * tab.pod_pixels * 256 * 256
tiles = copies * sum(((h + 1) // 2) * ((w + 3) // 4) for h, w in levels)      # 2 x 4 output tiles
print("wino %s x%d: %.3f ms/launch = %.1f TFLOP/s direct-equivalent, %.1f executed on the matrix cores (%d tiles, %d with block padding)" % (
    mode, copies, ms, direct / ms / 1e9, 2.0 * 24 * tiles * 256 * 256 / ms / 1e9, tiles, tab.shape[0] * 32))
