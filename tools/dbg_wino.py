"""GPU box debug: pod_wino_conv3x3 with one-tap filters (output = shifted input) -- shows which pixels / channels go wrong."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W = int(sys.argv[3]) if len(sys.argv) > 3 else 16
copies = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for dy in range(3):
    for dx in range(3):
        w = torch.zeros(64, C, 3, 3, device="cuda")
        for k in range(min(64, C)):
            w[k, k, dy, dx] = 1.0
        conv = WinoConv(w, None)
        x = (torch.arange(copies * H * W * C, device="cuda", dtype=torch.float32) % 997).view(copies * H * W, C)
        out = torch.full((copies * H * W, 64), float("nan"), device="cuda")
        conv(x, out, block_table([(H, W)], copies, "cuda"))
        torch.cuda.synchronize()
        want = F.conv2d(x.view(copies, H, W, C).permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).reshape(copies * H * W, 64)
        err = (out - want).abs().view(copies, H, W, 64)
        bad = ~(err <= 1e-2)
        print("tap", dy, dx, "max err", float(err.max()), "bad", int(bad.sum()))
        if bad.any():
            print("  bad per row", bad.sum((0, 2, 3)).tolist())
            print("  bad per col", bad.sum((0, 1, 3)).tolist())
            for n, y, xx, c in bad.nonzero()[:4].tolist():
                print("  ", n, y, xx, c, float(out.view(copies, H, W, 64)[n, y, xx, c]), float(want.view(copies, H, W, 64)[n, y, xx, c]))
