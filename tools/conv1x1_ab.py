"""GPU box: sha-256 of pod_conv1x1_split's outputs on a few shapes (seeded inputs): run it with two builds of the library
(POD_MI355X_LIB=...) to compare them bit for bit across processes (round 4: the LDS form against the direct-fragment form, -DPOD_C1_DIRECT)."""
import hashlib
import sys
import torch
sys.path.insert(0, ".")
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
for cin, cout, h, w, s, res, splits in ((256, 1024, 48, 84, 1, True, 1), (64, 256, 192, 336, 1, True, 1), (1024, 256, 48, 84, 1, False, 4), (512, 1024, 96, 168, 2, False, 1),
                                        (2048, 256, 24, 42, 1, True, 16), (96, 128, 37, 53, 2, True, 1), (64, 64, 5, 7, 1, False, 2)):
    g = torch.Generator(device="cuda").manual_seed(cin * 31 + cout)
    wt = torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * 0.05
    b = torch.randn(cout, device="cuda", generator=g)
    x = torch.randn(h * w, cin, device="cuda", generator=g)
    conv = Conv1x1(wt, b, s)
    ho, wo = conv.out_hw(h, w)
    r = torch.randn(ho * wo, cout, device="cuda", generator=g) if res else None
    y = conv(x, h, w, relu=False, residual=r, n_splits=splits)
    print("%4d -> %4d %3dx%3d s%d splits %2d: %s" % (cin, cout, ho, wo, s, splits, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:24]))
