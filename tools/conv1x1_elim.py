"""GPU box: a few 1x1 shapes on the library named by POD_MI355X_LIB (elimination builds of k13: tools/README.md)."""
import sys
import torch
sys.path.insert(0, ".")
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
for name, cin, cout, h, w, s, res in (("res4 conv3", 256, 1024, 48, 84, 1, True), ("res2 conv3", 64, 256, 192, 336, 1, True), ("res3 conv3", 128, 512, 96, 168, 1, True),
                                      ("res5 conv3", 512, 2048, 24, 42, 1, True), ("res4 conv1", 1024, 256, 48, 84, 1, False)):
    torch.manual_seed(1)
    wt = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    x = torch.randn(h * w, cin, device="cuda").relu()
    conv = Conv1x1(wt, b, s)
    ho, wo = conv.out_hw(h, w)
    r = torch.randn(ho * wo, cout, device="cuda") if res else None
    for splits in (1, conv.splits_for(ho * wo)):
        for _ in range(5):
            conv(x, h, w, relu=True, residual=r, n_splits=splits)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            conv(x, h, w, relu=True, residual=r, n_splits=splits)
        e1.record()
        torch.cuda.synchronize()
        print("%-12s splits %d: %6.1f us" % (name, splits, 1e3 * e0.elapsed_time(e1) / 30))
