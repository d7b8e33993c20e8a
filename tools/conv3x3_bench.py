"""GPU box: pod_wino_conv3x3_split on the 3x3 / stride-1 convolutions of a ResNet-50-FPN backbone at the benchmark frame (768 x 1344), one
channels-last image per call as the channels-last backbone runs them (small maps: split over input channels + pod_reduce_partials), replayed
as HIP graphs; per shape and per number of splits.   python tools/conv3x3_bench.py"""
import sys
import torch
import torch.nn as nn
sys.path.insert(0, ".")
from pod_compare_amd import modeling, wino  # noqa: E402

# (name, channels, H, W, calls per image)
SHAPES = [("res2 conv2", 64, 192, 336, 3), ("res3 conv2", 128, 96, 168, 4), ("res4 conv2", 256, 48, 84, 6), ("res5 conv2", 512, 24, 42, 3),
          ("fpn output3", 256, 96, 168, 1), ("fpn output4", 256, 48, 84, 1), ("fpn output5", 256, 24, 42, 1)]


def timed(fn, n=30):
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


total = 0.0
for name, c, h, w, calls in SHAPES:
    conv = nn.Conv2d(c, c, 3, padding=1).cuda()
    wc = modeling.wino_of(conv)
    x = torch.randn(h * w, c, device="cuda").relu()
    table = wino.block_table([(h, w)], 1, x.device, channels=c)
    policy = wc.splits_for(int(table.shape[0]))
    row = []
    for s in (1, 2, 4):
        if (c // 16) % (2 * s) or (c // 16) // s < 4:
            continue
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            wc.channels_last_of_one_image(x, table, relu=True, n_splits=s)
            with torch.cuda.graph(g, stream=st):
                for _ in range(10):
                    wc.channels_last_of_one_image(x, table, relu=True, n_splits=s)
        t = timed(g.replay) / 10
        row.append("%d: %6.1f us" % (s, t))
        if s == policy:
            total += calls * t
            gf = 2.0 * 9 * h * w * c * c / 1e9
            row[-1] += " (%5.1f TFLOP/s direct-equivalent)" % (gf / t * 1e3)
    print("%-12s %3d ch %3dx%3d  %3d blocks x %d filter slices (policy %d) | %s   x%d" % (name, c, h, w, table.shape[0], c // 64, policy, "   ".join(row), calls), flush=True)
print("per image (19 calls): %.3f ms" % (total / 1e3))
