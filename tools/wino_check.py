"""GPU box: pod_wino_conv3x3 against torch's conv2d (MIOpen) on the head's shapes: max error and time per launch.
    python tools/wino_check.py [copies]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from pod_compare_amd import hip  # noqa: E402
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda")
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 18
LEVELS = [(90, 160), (45, 80), (23, 40), (12, 20), (6, 10)]


def run(levels, copies, C=256, K=256, relu=True, reps=5):
    w = (torch.randn(K, C, 3, 3, device=dev) * 0.05)
    b = torch.randn(K, device=dev)
    conv = WinoConv(w, b)
    xs = [torch.randn(copies, C, h, wd, device=dev).contiguous(memory_format=torch.channels_last) for h, wd in levels]
    flat_in = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).contiguous()
    flat_out = torch.full((flat_in.shape[0], K), float("nan"), device=dev)
    table = block_table(levels, copies, dev)
    conv(flat_in, flat_out, table, relu=relu)
    torch.cuda.synchronize()
    off, worst = 0, 0.0
    for x, (h, wd) in zip(xs, levels):
        ref = F.conv2d(x, w, b, padding=1)
        if relu:
            ref = ref.relu()
        got = flat_out[off:off + copies * h * wd].reshape(copies, h, wd, K).permute(0, 3, 1, 2)
        off += copies * h * wd
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, err)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        conv(flat_in, flat_out, table, relu=relu)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    flops = 2.0 * flat_in.shape[0] * C * K * 9
    ev[0].record()
    for _ in range(reps):
        for x in xs:
            F.conv2d(x, w, None, padding=1)
    ev[1].record()
    torch.cuda.synchronize()
    ms_ref = ev[0].elapsed_time(ev[1]) / reps
    print("levels %s copies %d C %d K %d: rel err %.2e | wino %.3f ms = %.1f TFLOP/s direct-equivalent | MIOpen %.3f ms = %.1f" % (
        levels, copies, C, K, worst, ms, flops / ms / 1e9, ms_ref, flops / ms_ref / 1e9), flush=True)
    return worst


if __name__ == "__main__":
    t0 = time.time()
    run([(16, 16)], 1, C=8, K=64, relu=False, reps=1)
    run([(6, 10), (23, 40)], 2, C=64, K=128, reps=1)
    run(LEVELS[:1], copies)
    run(LEVELS, copies)
    run(LEVELS, 1)
    # predictor shape: K = 63 planes from 9 of 18 runs
    import torch
    from pod_compare_amd.wino import level_pixel_offsets
    w = torch.randn(63, 256, 3, 3, device=dev) * 0.05
    conv = WinoConv(w, None)
    src = torch.randn(level_pixel_offsets(LEVELS, 18)[-1], 256, device=dev)
    out = torch.empty(level_pixel_offsets(LEVELS, 10)[-1] * 63, device=dev)
    tab = block_table(LEVELS, 9, dev, in_copies=18, in_first=9, out_copies=10)
    conv(src, out, tab, planes=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        conv(src, out, tab, planes=True)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    print("predictor K=63, 9 runs, all levels: %.3f ms = %.1f TFLOP/s direct-equivalent (real channels)" % (ms, 2.0 * 9 * 19220 * 256 * 63 * 9 / ms / 1e9))
    print("done in %.1f s" % (time.time() - t0))
