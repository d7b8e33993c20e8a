"""GPU box: checksum of pod_wino_conv3x3_split's output on the bench launch + a ragged launch, to compare two builds of the library
bit for bit across processes:   POD_MI355X_LIB=<lib A> python tools/k12_ab.py ; POD_MI355X_LIB=<lib B> python tools/k12_ab.py"""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

dev = torch.device("cuda")
for levels, copies, C, K in (([(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)], 3, 256, 256), ([(23, 40), (7, 9), (1, 1)], 2, 48, 64), ([(17, 33)], 1, 16, 128)):
    torch.manual_seed(C)
    conv = WinoConv(torch.randn(K, C, 3, 3, device=dev) * 0.03, torch.randn(K, device=dev), split=True)
    tab = block_table(levels, copies, dev)
    src = torch.randn(tab.pod_pixels, C, device=dev).relu()
    dst = torch.empty(tab.pod_pixels, K, device=dev)
    conv(src, dst, tab, relu=True, dropout_p=0.1, seed=1)
    torch.cuda.synchronize()
    print(C, K, copies, hashlib.sha256(dst.cpu().numpy().tobytes()).hexdigest()[:16], float(dst.abs().max()))
