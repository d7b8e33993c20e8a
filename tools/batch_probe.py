"""GPU box: what ONE launch per layer over B images would buy the trunk, measured before building it.  A batched launch of
B images is emulated by a B x taller map (the same tiles, the same K-range policy as batch 1 forced: n_splits / waves of the one-image shape),
1x1 on pod_conv1x1_split, 3x3 on pod_wino_conv3x3_split with B copies on the canvas; ten calls replayed as one graph, against B one-image calls.
    python tools/batch_probe.py [B]"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from conv1x1_shapes import SHAPES  # noqa: E402
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def timed_graph(fn, n=30):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n / 10


tot1 = totb = 0.0
print("1x1 (pod_conv1x1_split), us per IMAGE: one-image launches | one launch over %d images" % B)
for name, cin, cout, h, w, s, res, calls in SHAPES:
    if s != 1:
        hb = h * B            # (stride 2 on a taller map: rows pair up inside an image as long as h is even)
    else:
        hb = h * B
    torch.manual_seed(cin + cout)
    conv = Conv1x1(torch.randn(cout, cin, 1, 1, device="cuda") * (2.0 / cin) ** 0.5, torch.randn(cout, device="cuda"), s)
    ho, wo = conv.out_hw(h, w)
    x1 = torch.randn(h * w, cin, device="cuda").relu()
    xb = torch.randn(hb * w, cin, device="cuda").relu()
    r1 = torch.randn(ho * wo, cout, device="cuda") if res else None
    rb = torch.randn(ho * B * wo, cout, device="cuda") if res else None
    ns = conv.splits_for(ho * wo)
    tiles = ((ho * wo + 63) // 64) * (cout // 64)
    wv = conv.auto_waves(tiles * ns, cin // 16 // ns)
    t1 = timed_graph(lambda: conv(x1, h, w, relu=True, residual=r1))
    tb = timed_graph(lambda: conv(xb, hb, w, relu=True, residual=rb, n_splits=ns, waves=wv)) / B
    print("  %-20s %4d->%4d %3dx%3d s%d splits %2d waves %d: %6.1f | %6.1f   x%d" % (name, cin, cout, ho, wo, s, ns, wv, t1, tb, calls))
    tot1 += calls * t1
    totb += calls * tb
print("  per image (39 calls): %.3f ms | %.3f ms" % (tot1 / 1e3, totb / 1e3))

tot1 = totb = 0.0
print("3x3 (pod_wino_conv3x3_split), us per IMAGE: one-image launches (with their input-channel splits) | one launch over %d images" % B)
for name, (h, w), C, calls in (("res2 conv2", (192, 336), 64, 3), ("res3 conv2", (96, 168), 128, 4), ("res4 conv2", (48, 84), 256, 6), ("res5 conv2", (24, 42), 512, 3),
                               ("fpn output3", (96, 168), 256, 1), ("fpn output4", (48, 84), 256, 1), ("fpn output5", (24, 42), 256, 1)):
    torch.manual_seed(C)
    conv = WinoConv(torch.randn(C, C, 3, 3, device="cuda") * 0.03, torch.randn(C, device="cuda"), split=True)
    t1t, tbt = block_table([(h, w)], 1, "cuda"), block_table([(h, w)], B, "cuda")
    x1, xb = torch.randn(h * w, C, device="cuda").relu(), torch.randn(B * h * w, C, device="cuda").relu()
    yb = torch.empty(B * h * w, C, device="cuda")
    ns = conv.splits_for(int(t1t.shape[0]))
    t1 = timed_graph(lambda: conv.channels_last_of_one_image(x1, t1t, relu=True))
    tb = timed_graph(lambda: conv(xb, yb, tbt, relu=True)) / B
    print("  %-14s C %3d %3dx%3d one-image splits %d (%3d blocks): %6.1f | %6.1f   x%d   (batched: no input-channel split, %d workgroups)" % (
        name, C, h, w, ns, t1t.shape[0], t1, tb, calls, tbt.shape[0] * (C // 64)))
    tot1 += calls * t1
    totb += calls * tb
print("  per image (19 calls): %.3f ms | %.3f ms" % (tot1 / 1e3, totb / 1e3))
