"""GPU box debug (POD_WINO_DEBUG_X build): the raw patch values every lane read in the prologue vs what it should have read."""
import sys, torch
sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table
C, H, W = 8, 16, 16
w = torch.zeros(64, C, 3, 3, device="cuda")
conv = WinoConv(w, None)
x = (torch.arange(H * W * C, device="cuda", dtype=torch.float32)).view(H * W, C) + 1.0
out = torch.zeros(H * W, 64, device="cuda")
conv(x, out, block_table([(H, W)], 1, "cuda"))
torch.cuda.synchronize()
got = out.view(-1)[:256 * 48].view(256, 12, 4).cpu()
xin = x.view(H, W, C).cpu()
nbad = 0
for tid in range(256):
    lane, a = tid & 63, tid >> 6
    i32, h = lane & 31, lane >> 5
    ty, tx = i32 >> 2, i32 & 3
    row0 = 0 if a == 0 else 2 if a == 2 else 1
    row1 = 1 if a == 2 else 3 if a == 3 else 2
    for i in range(12):
        r = row1 if i >= 6 else row0
        col = i % 6
        gy, gx = 2 * ty + r - 1, 4 * tx + col - 1
        want = xin[gy, gx, 4 * h:4 * h + 4] if 0 <= gy < H and 0 <= gx < W else torch.zeros(4)
        if not torch.equal(got[tid, i], want):
            nbad += 1
            if nbad < 12:
                print("tid", tid, "piece", i, "got", got[tid, i].tolist(), "want", want.tolist())
print("bad pieces", nbad)
