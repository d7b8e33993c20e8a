"""GPU box: error of pod_wino_conv3x3 against a CPU fp64 direct convolution, per element, in units of 2^-24 * (|w| * |x| + |b|)
(the quantity a forward error bound of any fp32 evaluation is written in); the same for torch's fp32 conv2d on the GPU (MIOpen)
and on the CPU (mkldnn).   python tools/wino_fp64_check.py"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pod_compare_amd.wino import WinoConv, block_table, level_pixel_offsets

U = 2.0 ** -24
torch.manual_seed(0)
for name, levels, copies, C, K, planes in (("bench trunk", [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)], 1, 256, 256, False),
                                           ("cls_score", [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)], 1, 256, 63, True),
                                           ("bbox_pred", [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)], 1, 256, 36, True)):
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K)
    xs = [torch.randn(copies, C, h, wd).relu() for h, wd in levels]                 # post-ReLU activations, as the trunk sees them
    conv = WinoConv(w.cuda(), b.cuda())
    src = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).contiguous().cuda()
    offs = level_pixel_offsets(levels, copies)
    if planes:
        dst = torch.empty(offs[-1] * K, device="cuda")
        conv(src, dst, block_table(levels, copies, "cuda"), planes=True)
    else:
        dst = torch.empty(src.shape[0], K, device="cuda")
        conv(src, dst, block_table(levels, copies, "cuda"))
    worst = {"wino": 0.0, "miopen": 0.0, "cpu32": 0.0}
    rel = {"wino": 0.0, "miopen": 0.0, "cpu32": 0.0}
    for i, (x, (h, wd)) in enumerate(zip(xs, levels)):
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        bound = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
        if planes:
            got = dst[offs[i] * K:offs[i + 1] * K].view(copies, K, h, wd).cpu().double()
        else:
            got = dst[offs[i]:offs[i + 1]].view(copies, h, wd, K).permute(0, 3, 1, 2).cpu().double()
        cands = {"wino": got, "miopen": F.conv2d(x.cuda(), w.cuda(), b.cuda(), padding=1).cpu().double(), "cpu32": F.conv2d(x, w, b, padding=1).double()}
        for k, g in cands.items():
            worst[k] = max(worst[k], float(((g - want).abs() / (U * bound)).max()))
            rel[k] = max(rel[k], float((g - want).abs().max() / want.abs().max()))
    print("%-12s c = max |err| / (2^-24 (|w|*|x| + |b|)):  wino %.2f  miopen fp32 %.2f  cpu fp32 %.2f   | max err / max |y|: wino %.2e miopen %.2e cpu %.2e" % (
        name, worst["wino"], worst["miopen"], worst["cpu32"], rel["wino"], rel["miopen"], rel["cpu32"]))
