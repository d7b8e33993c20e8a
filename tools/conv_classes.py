"""GPU box: every convolution CLASS of the channels-last trunk (the 19 1x1 shapes of tools/conv1x1_shapes.py on pod_conv1x1_split, the 7
3x3 / stride-1 shapes of the bottlenecks and the FPN on pod_wino_conv3x3_split), N launches each, in one process -- for rocprofv3
--kernel-trace and --pmc passes; tools/conv_classes_report.py attributes the dispatches to the classes by their order.
    python tools/conv_classes.py [launches per class]      prints the plan (class, calls, FLOPs, algorithmic bytes) as JSON on the last line"""
import json
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from conv1x1_shapes import SHAPES  # noqa: E402
from pod_compare_amd import amax  # noqa: E402
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# 3x3 / stride-1 convolutions of the 768 x 1344 frame: (name, C, K, H, W, calls per image)
W3 = [("res2 conv2", 64, 64, 192, 336, 3), ("res3 conv2", 128, 128, 96, 168, 4), ("res4 conv2", 256, 256, 48, 84, 6), ("res5 conv2", 512, 512, 24, 42, 3),
      ("fpn output3", 256, 256, 96, 168, 1), ("fpn output4", 256, 256, 48, 84, 1), ("fpn output5", 256, 256, 24, 42, 1)]
plan = []
torch.manual_seed(0)
for name, cin, cout, h, w, s, res, calls in SHAPES:
    conv = Conv1x1(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, torch.randn(cout, device="cuda"), s)
    x = torch.randn(h * w, cin, device="cuda").relu()
    ho, wo = conv.out_hw(h, w)
    r = torch.randn(ho * wo, cout, device="cuda") if res else None
    amax.of(x)                                    # (pod_absmax once, outside the counted launches)
    splits = conv.splits_for(ho * wo)
    torch.cuda.synchronize()
    for _ in range(n):
        conv(x, h, w, relu=True, residual=r)
    torch.cuda.synchronize()
    # algorithmic bytes: the input pixels the stride reads, the filter terms (2 f16 per value), bias, residual, output
    bytes_ = 4 * (ho * wo * cin + ho * wo * cout * (2 if res else 1) + cout) + 4 * cin * cout
    plan.append({"class": name, "kind": "1x1", "calls_per_image": calls, "launches": n, "kernels_per_launch": 2 if splits > 1 else 1, "splits": splits,
                 "flop": 2.0 * ho * wo * cin * cout, "executed_f16_flop": 3 * 2.0 * (-(-ho * wo // 64) * 64) * cin * cout, "bytes": bytes_})
for name, c, k, h, w, calls in W3:
    conv = WinoConv(torch.randn(k, c, 3, 3, device="cuda") * 0.03, torch.randn(k, device="cuda"))
    x = torch.randn(h * w, c, device="cuda").relu()
    tab = block_table([(h, w)], 1, "cuda", channels=max(c, k))
    amax.of(x)
    splits = conv.splits_for(int(tab.shape[0]))
    torch.cuda.synchronize()
    for _ in range(n):
        conv.channels_last_of_one_image(x, tab, relu=True)
    torch.cuda.synchronize()
    tiles = int(tab.shape[0]) * 32
    plan.append({"class": name, "kind": "3x3", "calls_per_image": calls, "launches": n, "kernels_per_launch": 2 if splits > 1 else 1, "splits": splits,
                 "flop": 2.0 * 9 * h * w * c * k, "executed_f16_flop": 3 * 2.0 * 24 * tiles * c * k, "bytes": 4 * (h * w * (c + k) + k) + 4 * 24 * c * k})
print(json.dumps(plan))
