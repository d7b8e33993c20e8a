"""GPU box: do a small pod_wino_conv3x3 launch (a res5 bottleneck: 48 workgroups) and a big one (the bench launch: 6 496 workgroups) from
two streams overlap?  Times N small launches on stream B alone, the big ones on stream A alone, and both together -- B on a stream of
normal and of high priority."""
import sys
import time
import torch
sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table

dev = torch.device("cuda")
torch.manual_seed(0)
levels = [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]
big = WinoConv(torch.randn(256, 256, 3, 3, device=dev) * 0.03, torch.randn(256, device=dev))
tb = block_table(levels, 19, dev)
xb = torch.randn(tb.pod_pixels, 256, device=dev); yb = torch.empty_like(xb)
small = WinoConv(torch.randn(512, 512, 3, 3, device=dev) * 0.02, torch.randn(512, device=dev))
ts = block_table([(24, 42)], 1, dev)
xs = torch.randn(ts.pod_pixels, 512, device=dev); ys = torch.empty_like(xs)


def run(n_big, n_small, prio):
    sa = torch.cuda.Stream()
    sb = torch.cuda.Stream(priority=prio)
    for s, f in ((sa, lambda: big(xb, yb, tb, relu=True)), (sb, lambda: small(xs, ys, ts, relu=True))):
        with torch.cuda.stream(s):
            f(); f()
    torch.cuda.synchronize()
    ea, eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)], [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        ea[0].record()
        for _ in range(n_big):
            big(xb, yb, tb, relu=True)
        ea[1].record()
    with torch.cuda.stream(sb):
        eb[0].record()
        for _ in range(n_small):
            small(xs, ys, ts, relu=True)
        eb[1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return (ea[0].elapsed_time(ea[1]) / max(n_big, 1), eb[0].elapsed_time(eb[1]) / max(n_small, 1), 1e3 * wall)


print("big alone: %.3f ms per launch" % run(20, 0, 0)[0])
print("small alone: %.3f ms per launch" % run(0, 200, 0)[1])
for prio, name in ((0, "normal"), (-1, "high")):
    a, b, w = run(20, 200, prio)
    print("together, small on a %s-priority stream: big %.3f ms per launch, small %.3f ms per launch, wall %.1f ms" % (name, a, b, w))

# ---- the same question for an HBM-bound element-wise pass (pod_bias_act with a residual on a res2-sized map: 66 MB in place + 66 MB residual)
from pod_compare_amd import hip
lib, P = hip.load(), hip.ptr
x = torch.randn(1, 256, 192, 336, device=dev); r = torch.randn_like(x); bb = torch.randn(256, device=dev)


def ew():
    hip.check(lib.pod_bias_act(x.data_ptr(), P(bb), P(r), None, x.numel(), 256, 192 * 336, 1, 0.0, 0, 0, hip.current_stream()), "pod_bias_act")


def run2(n_big, n_ew):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        big(xb, yb, tb, relu=True)
    with torch.cuda.stream(sb):
        ew()
    torch.cuda.synchronize()
    ea, eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)], [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        ea[0].record()
        for _ in range(n_big):
            big(xb, yb, tb, relu=True)
        ea[1].record()
    with torch.cuda.stream(sb):
        eb[0].record()
        for _ in range(n_ew):
            ew()
        eb[1].record()
    torch.cuda.synchronize()
    return ea[0].elapsed_time(ea[1]) / max(n_big, 1), eb[0].elapsed_time(eb[1]) / max(n_ew, 1), 1e3 * (time.perf_counter() - t0)


print("bias_act alone: %.3f ms per call" % run2(0, 200)[1])
a, b, w = run2(20, 200)
print("together: big %.3f ms per launch, bias_act %.3f ms per call, wall %.1f ms (big alone would be %.1f, bias_act alone %.1f)" % (a, b, w, 20 * 1.535, 200 * run2(0, 200)[1]))
