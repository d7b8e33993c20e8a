"""GPU box: pod_conv1x1_split per shape of the ResNet-50-FPN, per number of input-channel splits over workgroup sets (grid.y) and per number of
wavefronts sharing a tile's K range inside the workgroup (round 5) -- the policy of Conv1x1.splits_for / auto_waves is read off this table.
python tools/conv1x1_splits.py"""
import sys
import torch
sys.path.insert(0, ".")
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
sys.path.insert(0, "tools")
from conv1x1_shapes import SHAPES  # noqa: E402


def timed(fn, n=40):
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


for name, cin, cout, h, w, s, res, calls in SHAPES:
    wt = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    x = torch.randn(h * w, cin, device="cuda").relu()
    conv = Conv1x1(wt, b, s)
    ho, wo = conv.out_hw(h, w)
    r = torch.randn(ho * wo, cout, device="cuda") if res else None
    tiles = ((ho * wo + 63) // 64) * (cout // 64)
    row, best = [], None
    for splits in (1, 2, 4, 8, 16):
        if (cin // 16) % splits or (cin // 16) // splits < 2 or tiles * splits > 5000:
            continue
        for waves in (1, 2, 4):
            per = (cin // 16) // splits
            if waves > 1 and (per % (2 * waves) or tiles * splits * waves > 4096):
                continue
            g = torch.cuda.CUDAGraph()          # replayed as a graph: no host time in the figure
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                conv(x, h, w, relu=True, residual=r, n_splits=splits, waves=waves)
                with torch.cuda.graph(g, stream=st):
                    for _ in range(10):
                        conv(x, h, w, relu=True, residual=r, n_splits=splits, waves=waves)
            t = timed(g.replay) / 10
            row.append("%dx%d: %5.1f" % (splits, waves, t))
            if best is None or t < best[0]:
                best = (t, splits, waves)
    pol_s = conv.splits_for(ho * wo)
    pol_w = Conv1x1.auto_waves(tiles * pol_s, (cin // 16) // pol_s)
    print("%-20s %4d->%4d %5d px %5d tiles (policy %dx%d, best %dx%d %.1f us) | %s" % (name, cin, cout, ho * wo, tiles, pol_s, pol_w, best[1], best[2], best[0], "  ".join(row)), flush=True)
