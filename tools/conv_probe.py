import torch, time, sys
torch.manual_seed(0)
dev = "cuda"
conv = torch.nn.Conv2d(256, 256, 3, padding=1).to(dev)
for (H, W) in [(96, 168), (48, 84)]:
    for B in [1, 2, 3, 4, 5, 6, 8, 10, 19]:
        x = torch.randn(B, 256, H, W, device=dev)
        with torch.no_grad():
            for _ in range(3): y = conv(x)
            torch.cuda.synchronize(); t = time.perf_counter()
            n = 10
            for _ in range(n): y = conv(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        fl = 2 * B * H * W * 256 * 256 * 9
        print("HxW %dx%d  B=%2d  %8.1f us  %6.1f TFLOP/s effective  (%.1f us per image)" % (H, W, B, dt * 1e6, fl / dt / 1e12, dt * 1e6 / B), flush=True)
