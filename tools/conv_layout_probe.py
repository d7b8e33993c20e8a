"""One head-trunk conv (256->256 3x3, batch 19) on p3 / p4 maps: NCHW vs channels_last input, bias-free, timed with events.
PYTORCH_MIOPEN_SUGGEST_NHWC=1 in the environment lets MIOpen see NHWC descriptors."""
import os, sys, time
import torch
import torch.nn.functional as F
torch.manual_seed(0)
dev = "cuda"
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
for (H, W) in [(96, 168), (48, 84)]:
    for B in (19,):
        x = torch.randn(B, 256, H, W, device=dev)
        for tag, xx, ww in (("NCHW", x, w), ("channels_last", x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last))):
            with torch.no_grad():
                for _ in range(3):
                    y = F.conv2d(xx, ww, None, 1, 1)
                torch.cuda.synchronize(); t = time.perf_counter()
                n = 10
                for _ in range(n):
                    y = F.conv2d(xx, ww, None, 1, 1)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            fl = 2 * B * H * W * 256 * 256 * 9
            print("%dx%d B=%d %-14s %8.1f us  %6.1f TFLOP/s  out channels_last=%s" % (H, W, B, tag, dt * 1e6, fl / dt / 1e12,
                  y.is_contiguous(memory_format=torch.channels_last)), flush=True)
