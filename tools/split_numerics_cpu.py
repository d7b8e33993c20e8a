"""CPU study (no GPU needed): error constant c of the Winograd F(2,3) x F(4,3) 3x3 convolution against an fp64 direct convolution,
with the products formed (a) in plain fp32, (b) from 3-way bf16 splits / 6 partial products (pod_wino_conv3x3_split today),
(c) from 2-way fp16 splits / 3 partial products with a power-of-two operand scale (the round-5 candidate).
c = max over outputs of |err| / (2^-24 (|w| * |x| + |b|)), the unit tests/test_wino_conv_gpu.py uses.
Transforms run in fp32 (torch CPU); the term matrices are multiplied in fp32 with fp32 accumulation (order differs from the
matrix cores', the rounding model is the same: exact products, fp32 sums).   python tools/split_numerics_cpu.py [K ...]"""
import sys

import torch
import torch.nn.functional as F

torch.set_num_threads(16)
G4 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
G6 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float32)
B4 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
B6 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
A2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
A4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)


def bf16_terms(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        t = r.to(torch.bfloat16).to(torch.float32)
        out.append(t)
        r = r - t
    return out


def f16_terms(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        t = r.to(torch.float16).to(torch.float32)
        out.append(t)
        r = r - t
    return out


def terms64(x64, n, dt):
    out, r = [], x64.clone()
    for _ in range(n):
        t = r.to(torch.float32).to(dt).to(torch.float64)      # (double rounding fp64 -> fp32 -> 16 bit: what a device conversion does)
        out.append(t.to(torch.float32))
        r = r - t
    return out


def wino(x, w, b, mode):
    """x (1, C, H, W) fp32, H % 2 == 0, W % 4 == 0.  mode: fp32 | bf16x6 | f16x3 | f16x3_noscale | f16x4"""
    _, C, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum("ai,kcij,bj->abkc", G4, w, G6)                     # (4, 6, K, C)
    U64 = torch.einsum("ai,kcij,bj->abkc", G4.double(), w.double(), G6.double())
    xp = F.pad(x, (1, 1, 1, 1))[0]
    d = xp.unfold(1, 4, 2).unfold(2, 6, 4)                              # (C, th, tw, 4, 6)
    th, tw = d.shape[1], d.shape[2]
    V = torch.einsum("ai,ctuij,bj->abctu", B4, d, B6).reshape(4, 6, C, th * tw)
    M = torch.empty(4, 6, K, th * tw)
    for a in range(4):
        for p in range(6):
            u, v = U[a, p], V[a, p]
            if mode == "fp32":
                M[a, p] = u @ v
            elif mode == "bf16x6_u64":
                u0, u1, u2 = terms64(U64[a, p], 3, torch.bfloat16)
                v0, v1, v2 = bf16_terms(v, 3)
                acc = u1 @ v1
                for s, t in ((u2, v0), (u0, v2), (u1, v0), (u0, v1), (u0, v0)):
                    acc = acc + s @ t
                M[a, p] = acc
            elif mode == "f16x3_u64":
                su = 2.0 ** (14 - torch.floor(torch.log2(U.abs().amax(dim=(0, 1, 3)).clamp(min=1e-30)))).reshape(K, 1)
                sv = 2.0 ** (15 - torch.ceil(torch.log2(20.0 * x.abs().max())))
                us, vs = terms64(U64[a, p] * su.double(), 2, torch.float16), f16_terms(v * sv, 2)
                acc = us[1] @ vs[0]
                acc = acc + us[0] @ vs[1]
                acc = acc + us[0] @ vs[0]
                M[a, p] = acc / su / sv
            elif mode == "bf16x6":
                u0, u1, u2 = bf16_terms(u, 3)
                v0, v1, v2 = bf16_terms(v, 3)
                acc = u1 @ v1
                for s, t in ((u2, v0), (u0, v2), (u1, v0), (u0, v1), (u0, v0)):
                    acc = acc + s @ t
                M[a, p] = acc
            else:
                if mode.endswith("noscale"):
                    su = sv = 1.0
                else:
                    # power-of-two scales: filter per output channel (static), patch per launch (the abs-max of the INPUT tensor, x 20 =
                    # the largest gain of Bt4 (x) Bt6, rounded up to a power of two, put at 2^15)
                    su = 2.0 ** (14 - torch.floor(torch.log2(U.abs().amax(dim=(0, 1, 3)).clamp(min=1e-30)))).reshape(K, 1)
                    sv = 2.0 ** (15 - torch.ceil(torch.log2(20.0 * x.abs().max())))
                us, vs = f16_terms(u * su, 2), f16_terms(v * sv, 2)
                assert torch.isfinite(us[0]).all() and torch.isfinite(vs[0]).all()
                if mode.startswith("f16x4"):
                    acc = us[1] @ vs[1] + us[1] @ vs[0]
                else:
                    acc = us[1] @ vs[0]
                acc = acc + us[0] @ vs[1]
                acc = acc + us[0] @ vs[0]
                M[a, p] = acc / su / sv
    Y = torch.einsum("ia,abkt,jb->ktij", A2, M, A4).reshape(K, th, tw, 2, 4).permute(0, 1, 3, 2, 4).reshape(1, K, H, W)
    return Y + b.reshape(1, K, 1, 1)


def study(K, levels, act_scale=1.0):
    C, u = 256, 2.0 ** -24
    g = torch.Generator().manual_seed(K)
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, generator=g)
    res = {}
    for h, wd in levels:
        x = torch.randn(1, C, h, wd, generator=g).relu() * act_scale
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        bound = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
        for mode in ("fp32", "bf16x6", "bf16x6_u64", "f16x3", "f16x3_u64"):
            try:
                got = wino(x, w, b, mode).double()
                c = float(((got - want).abs() / (u * bound)).max())
                r = float(((got - want).abs() / (u * bound)).pow(2).mean().sqrt())
            except AssertionError:
                c = r = float("inf")
            res.setdefault(mode, [0.0, 0.0])
            res[mode][0] = max(res[mode][0], c)
            res[mode][1] = max(res[mode][1], r)
    return res


if __name__ == "__main__":
    Ks = [int(a) for a in sys.argv[1:]] or [256, 63, 36]
    levels = [(48, 84), (24, 44), (12, 24), (6, 12)]
    for scale in (1.0, 1e-3, 300.0):
        for K in Ks:
            r = study(K, levels, scale)
            print("act x %-6g K=%-3d " % (scale, K) + "  ".join("%s c=%.2f rms=%.3f" % (m, v[0], v[1]) for m, v in r.items()), flush=True)
