"""GPU box, EXPERIMENT build (POD_WITH_K16=1 POD_BUILD_TAG=k16 POD_TAG_SOURCES="k16_wino_conv_split8.hip k12_wino_conv_split.hip" python -m pod_compare_amd.build;
POD_MI355X_LIB=pod_compare_amd/lib/k16/libpod_mi355x.so): the two workgroup forms of pod_wino_conv3x3_split (POD_WINO_FORM_4 = k12, shipped; POD_WINO_FORM_8 =
tools/experiments/k16_wino_conv_split8.hip, round 6: measured and not shipped -- profiles/r06_k16_two_wavefronts_per_simd.md) on the same
launches -- every form of the launch: channels-last with dropout, NCHW planes, replicas, grouped sets, input-channel splits, ragged maps --
compared BIT FOR BIT, then timed on the bench launch (19 runs x 5 levels, C = K = 256) with HIP events.
    python tools/k16_ab.py [timed launches]"""
import sys

import torch

sys.path.insert(0, ".")
from pod_compare_amd import amax, wino  # noqa: E402
from pod_compare_amd.wino import WinoConv, block_table, grouped_launch  # noqa: E402

dev = torch.device("cuda")
BENCH = [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]


def both(fn):
    """fn() -> list of tensors; runs it under form 4 and form 8 and compares."""
    outs = []
    for form in (4, 8):
        wino.FORM = form
        outs.append([t.clone() for t in fn()])
        torch.cuda.synchronize()
    wino.FORM = 0
    return all(torch.equal(a, b) for a, b in zip(*outs)), max(float((a - b).abs().max()) for a, b in zip(*outs)), outs


def case_plain(levels, copies, C, K, p, relu=True):
    torch.manual_seed(C + K)
    conv = WinoConv(torch.randn(K, C, 3, 3, device=dev) * 0.03, torch.randn(K, device=dev), split=True)
    tab = block_table(levels, copies, dev)
    src = torch.randn(tab.pod_pixels, C, device=dev).relu()

    def run():
        dst = torch.full((tab.pod_pixels, conv.Kpad), float("nan"), device=dev)
        conv(src, dst, tab, relu=relu, dropout_p=p, seed=3, offset=1 << 34)
        return [dst, amax.of(dst).clone()]
    return run


def case_planes(levels, copies, C, K):
    torch.manual_seed(C + K + 1)
    conv = WinoConv(torch.randn(K, C, 3, 3, device=dev) * 0.03, torch.randn(K, device=dev), split=True)
    tab = block_table(levels, copies, dev)
    src = torch.randn(tab.pod_pixels, C, device=dev).relu()

    def run():
        dst = torch.full((tab.pod_pixels * K,), float("nan"), device=dev)
        conv(src, dst, tab, planes=True)
        return [dst]
    return run


def case_replicas(levels, copies, C):
    torch.manual_seed(C + 2)
    conv = WinoConv(torch.randn(C, C, 3, 3, device=dev) * 0.03, torch.randn(C, device=dev), split=True)
    t1 = block_table(levels, 1, dev)
    tr = block_table(levels, 1, dev, out_copies=copies)
    src = torch.randn(t1.pod_pixels, C, device=dev)

    def run():
        dst = torch.full((t1.pod_pixels * copies, C), float("nan"), device=dev)
        conv.replicas(src, dst, tr, copies, relu=True, dropout_p=0.2, seed=5, offset=7 << 34)
        return [dst, amax.of(dst).clone()]
    return run


def case_grouped(levels, copies, C):
    torch.manual_seed(C + 3)
    convs = [WinoConv(torch.randn(C, C, 3, 3, device=dev) * 0.03, torch.randn(C, device=dev), split=True) for _ in range(3)]
    tabs = [block_table(levels, c, dev) for c in (copies, copies + 1, 1)]
    srcs = [torch.randn(t.pod_pixels, C, device=dev).relu() for t in tabs]

    def run():
        dsts = [torch.full((t.pod_pixels, C), float("nan"), device=dev) for t in tabs]
        grouped_launch([{"conv": c, "src": s, "dst": d, "table": t, "offset": (i + 1) << 34} for i, (c, s, d, t) in enumerate(zip(convs, srcs, dsts, tabs))],
                       relu=True, dropout_p=0.1, seed=9)
        return dsts
    return run


def case_splits(h, w, C, K, n_splits):
    torch.manual_seed(C + 4)
    conv = WinoConv(torch.randn(K, C, 3, 3, device=dev) * 0.03, torch.randn(K, device=dev), split=True)
    tab = block_table([(h, w)], 1, dev)
    src = torch.randn(h * w, C, device=dev).relu()
    return lambda: [conv.channels_last_of_one_image(src, tab, relu=True, n_splits=n_splits)]


CASES = [("bench launch x3, dropout", case_plain(BENCH, 3, 256, 256, 0.1)),
         ("ragged maps, C 48 -> K 64", case_plain([(23, 40), (7, 9), (1, 1)], 2, 48, 64, 0.0)),
         ("one map, C 16 -> K 128, no relu", case_plain([(17, 33)], 1, 16, 128, 0.0, relu=False)),
         ("C 512 -> K 512 (res5)", case_plain([(24, 42)], 1, 512, 512, 0.0)),
         ("planes K = 63 (cls_score)", case_planes(BENCH, 2, 256, 63)),
         ("planes K = 36, ragged", case_planes([(23, 40), (7, 9)], 3, 256, 36)),
         ("replicas x5", case_replicas(BENCH, 5, 256)),
         ("three grouped sets", case_grouped([(48, 84), (12, 21)], 2, 256)),
         ("input-channel splits x4 (res5)", case_splits(24, 42, 512, 512, 4)),
         ("input-channel splits x2", case_splits(12, 21, 256, 256, 2))]

ok_all = True
for name, fn in CASES:
    ok, diff, _ = both(fn)
    ok_all &= ok
    print("%-36s %s  max |form 8 - form 4| = %.3e" % (name, "BIT-IDENTICAL" if ok else "DIFFERENT", diff))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
conv = WinoConv(torch.randn(256, 256, 3, 3, device=dev) * 0.03, torch.randn(256, device=dev), split=True)
tab = block_table(BENCH, 19, dev)
src = torch.randn(tab.pod_pixels, 256, device=dev).relu()
dst = torch.empty_like(src)
for form in (4, 8, 4, 8):
    wino.FORM = form
    for _ in range(3):
        conv(src, dst, tab, relu=True, dropout_p=0.1, seed=1)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        conv(src, dst, tab, relu=True, dropout_p=0.1, seed=1)
    ev[1].record()
    torch.cuda.synchronize()
    print("form %d: bench launch (19 runs x 5 levels) %.4f ms" % (form, ev[0].elapsed_time(ev[1]) / n))
print("ALL BIT-IDENTICAL" if ok_all else "SOME CASES DIFFER")

# the backbone's / FPN's 3x3 launches (one image, batch 1): few workgroups per launch -- latency, not power, prices them
print("one-image launches (backbone / FPN shapes), us per launch: form 4 | form 8")
for name, hw, C, K in (("res2 conv2", (192, 336), 64, 64), ("res3 conv2", (96, 168), 128, 128), ("res4 conv2", (48, 84), 256, 256), ("res5 conv2", (24, 42), 512, 512),
                           ("fpn output3", (96, 168), 256, 256), ("fpn output4", (48, 84), 256, 256), ("fpn output5", (24, 42), 256, 256),
                           ("head level p3 x1", (96, 168), 256, 256), ("head 5 levels x1", None, 256, 256), ("head 5 levels x4", None, 256, 256)):
    torch.manual_seed(1)
    conv = WinoConv(torch.randn(K, C, 3, 3, device=dev) * 0.03, torch.randn(K, device=dev), split=True)
    copies = 4 if name.endswith("x4") else 1
    tab = block_table(BENCH if hw is None else [hw], copies, dev)
    src = torch.randn(tab.pod_pixels, C, device=dev).relu()
    res = []
    for form in (4, 8):
        wino.FORM = form
        if name.startswith("head"):
            dst = torch.empty(tab.pod_pixels, K, device=dev)
            run = lambda: conv(src, dst, tab, relu=True)
        else:
            run = lambda: conv.channels_last_of_one_image(src, tab, relu=True)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(50):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        res.append(ev[0].elapsed_time(ev[1]) / 50 * 1e3)
    wino.FORM = 0
    print("  %-18s %4d blocks x %d slices, splits %d: %8.1f | %8.1f" % (name, tab.shape[0], conv.Kpad // 64, conv.splits_for(int(tab.shape[0])), res[0], res[1]))
