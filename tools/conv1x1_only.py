"""GPU box: ONE 1x1 shape of the backbone on pod_conv1x1_split, n launches (for rocprofv3 --pmc / --kernel-trace passes).
python tools/conv1x1_only.py <shape index into tools/conv1x1_shapes.py> [launches]"""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from conv1x1_shapes import SHAPES  # noqa: E402
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
name, cin, cout, h, w, s, res, calls = SHAPES[int(sys.argv[1])]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wt = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
b = torch.randn(cout, device="cuda")
x = torch.randn(h * w, cin, device="cuda").relu()
conv = Conv1x1(wt, b, s)
ho, wo = conv.out_hw(h, w)
r = torch.randn(ho * wo, cout, device="cuda") if res else None
for _ in range(n):
    conv(x, h, w, relu=True, residual=r)
torch.cuda.synchronize()
print(name, "splits", conv.splits_for(ho * wo))
