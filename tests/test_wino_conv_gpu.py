"""pod_wino_conv3x3 (csrc/k11_wino_conv.hip): the head's 3x3 convolutions (probabilistic_retinanet.py:403-484) as fp32 Winograd
on the matrix cores.  Referees: a CPU fp64 direct convolution with a per-element bound in units of 2^-24 (|w| * |x| + |b|)
(`test_error_against_an_fp64_direct_convolution`), and torch's conv2d on the same tensors with 2e-5 of the output's scale (fp32
Winograd, F(2,3) down the rows x F(4,3) along the columns, differs from a direct fp32 convolution by ~2e-6 of the scale at C = 256)."""
import math

import pytest
import torch
import torch.nn.functional as F

from pod_compare_amd import hip, modeling
from pod_compare_amd.wino import WinoConv, block_table, level_pixel_offsets

pytestmark = pytest.mark.gpu
TOL = 2e-5


def flat(xs):
    return torch.cat([x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]) for x in xs]).contiguous()


def make(levels, copies, C, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn(K, C, 3, 3, device="cuda", generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device="cuda", generator=g)
    xs = [torch.randn(copies, C, h, wd, device="cuda", generator=g) for h, wd in levels]
    return w, b, xs


@pytest.mark.parametrize("levels,copies,C,K", [
    ([(16, 16)], 1, 8, 64),                                   # one block, one chunk
    ([(1, 1), (2, 3), (17, 33)], 2, 16, 64),                  # degenerate maps, partial blocks
    ([(23, 40), (12, 20), (6, 10)], 3, 64, 128),              # the small FPN levels of a 720p frame (odd sizes, W % 4 != 0)
    ([(45, 80), (6, 10)], 2, 256, 256),                       # the head's channel counts
    ([(20, 24)], 1, 32, 512),
])
@pytest.mark.parametrize("split", [False, True], ids=["fp32-mfma", "f16x3"])
def test_channels_last_output_equals_conv2d(levels, copies, C, K, split):
    """split: pod_wino_conv3x3_split (2-way f16 splits on the f16 matrix cores; needs C % 16 == 0, else the fp32 kernel serves)."""
    w, b, xs = make(levels, copies, C, K)
    conv = WinoConv(w, b, split=split)
    assert conv.split == (split and C % 16 == 0)
    src = flat(xs)
    dst = torch.full((src.shape[0], K), float("nan"), device="cuda")
    conv(src, dst, block_table(levels, copies, "cuda"), relu=True)
    offs = level_pixel_offsets(levels, copies)
    for i, (x, (h, wd)) in enumerate(zip(xs, levels)):
        want = F.conv2d(x, w, b, padding=1).relu()
        got = dst[offs[i]:offs[i + 1]].view(copies, h, wd, K).permute(0, 3, 1, 2)
        assert torch.isfinite(got).all()
        assert float((got - want).abs().max()) <= TOL * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("split", [False, True], ids=["fp32-mfma", "f16x3"])
@pytest.mark.parametrize("K", [63, 36, 90])
def test_predictor_planes_of_a_subset_of_the_runs(K, split):
    """cls_score / bbox_pred / bbox_cov shapes: K real channels (padded to 64 / 128 inside), NCHW planes out, reading runs
    first .. first+count-1 of a buffer of 5 runs per level, writing a buffer of 4 runs per level whose last run stays as it was."""
    levels, in_copies, first, count, out_copies = [(23, 40), (12, 20), (6, 10), (3, 5)], 5, 2, 3, 4
    w, b, xs = make(levels, in_copies, 64, K, seed=K)
    conv = WinoConv(w, b, split=split)
    src = flat(xs)
    offs = level_pixel_offsets(levels, out_copies)
    out = torch.full((offs[-1] * K,), 7.0, device="cuda")
    conv(src, out, block_table(levels, count, "cuda", in_copies=in_copies, in_first=first, out_copies=out_copies), planes=True)
    for i, (x, (h, wd)) in enumerate(zip(xs, levels)):
        got = out[offs[i] * K:offs[i + 1] * K].view(out_copies, K, h, wd)
        want = F.conv2d(x[first:first + count], w, b, padding=1)
        assert float((got[:count] - want).abs().max()) <= TOL * max(1.0, float(want.abs().max()))
        assert bool((got[count:] == 7.0).all())


BENCH_LEVELS = [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]      # the five FPN levels of the 768 x 1344 benchmark frame
C_BOUND = 32.0


def _c_against_fp64(K, planes, split):
    """max over output elements of |err| / (2^-24 (|w| * |x| + |b|)) on the benchmark launch's shapes, referee = fp64 direct convolution"""
    C, copies, u = 256, 1, 2.0 ** -24
    g = torch.Generator().manual_seed(K)
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, generator=g)
    xs = [torch.randn(copies, C, h, wd, generator=g).relu() for h, wd in BENCH_LEVELS]
    conv = WinoConv(w.cuda(), b.cuda(), split=split)
    assert conv.split == split
    src = flat([x.cuda() for x in xs])
    offs = level_pixel_offsets(BENCH_LEVELS, copies)
    dst = torch.full((offs[-1] * K,) if planes else (src.shape[0], K), float("nan"), device="cuda")
    conv(src, dst, block_table(BENCH_LEVELS, copies, "cuda"), planes=planes)
    c_max = 0.0
    for i, (x, (h, wd)) in enumerate(zip(xs, BENCH_LEVELS)):
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        bound = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
        got = (dst[offs[i] * K:offs[i + 1] * K].view(copies, K, h, wd) if planes
               else dst[offs[i]:offs[i + 1]].view(copies, h, wd, K).permute(0, 3, 1, 2)).cpu().double()
        assert torch.isfinite(got).all()
        c_max = max(c_max, float(((got - want).abs() / (u * bound)).max()))
    return c_max


@pytest.mark.parametrize("K,planes", [(256, False), (63, True), (36, True)], ids=["trunk-256", "cls_score-63", "bbox_pred-36"])
def test_error_against_an_fp64_direct_convolution(K, planes):
    """The referee that is not MIOpen: F.conv2d in fp64 on the CPU, on the benchmark launch's shapes (C = 256, five levels) with
    post-ReLU activations.  Per ELEMENT  |err| <= c 2^-24 (|w| * |x| + |b|)  -- the unit every forward error bound of an fp32
    evaluation is written in (a length-n fp32 dot product guarantees c <= n = 2304).  Measured c (tools/wino_fp64_check.py, MI355X):
    pod_wino_conv3x3 11.6 / 10.8 / 13.6 (trunk / cls_score / bbox_pred), MIOpen's fp32 conv2d 2.5 - 2.8, mkldnn's fp32 conv2d on the
    CPU 2.9 - 5.2: fp32 Winograd costs a factor ~4 over a direct fp32 sum and stays two orders of magnitude inside the fp32 class.
    pod_wino_conv3x3_split, rounds 3-4 (exact 3-way bf16 splits, 6 partial products, fp32 accumulate): 10.6 / 9.6 / 8.5; round 5 (2-way f16 splits of the
    scaled operands, 3 partial products): printed by this test.
    THE CONTRACT OF THE SPLIT KERNEL, per shape: c(K12) <= c(K11) -- it is at least as close to the fp64 result as the fp32-MFMA
    kernel on every shape it replaces it on (same Winograd, same operation order: the two differ only in how a product is formed)."""
    c11 = _c_against_fp64(K, planes, False)
    c12 = _c_against_fp64(K, planes, True)
    print("c(K11) = %.2f  c(K12) = %.2f" % (c11, c12))
    assert c11 <= C_BOUND and c12 <= C_BOUND
    assert c12 <= c11, "the split kernel must not be further from fp64 than the fp32-MFMA kernel (%.2f vs %.2f)" % (c12, c11)


def _split2(x, scale):
    n = x.numel()
    terms = torch.empty((2, n), dtype=torch.int16, device="cuda")
    hip.check(hip.load().pod_debug_f16_split2(x.data_ptr(), float(scale), terms.data_ptr(), n, hip.current_stream()), "pod_debug_f16_split2")
    return terms.view(torch.float16)


def test_two_f16_terms_carry_the_scaled_fp32_value_to_2_pow_minus_23():
    """The arithmetic contract of the round-5 split kernels, first half: for the values the kernels' own split code produces
    (pod_debug_f16_split2 runs pod_wino.h: wino_f16_split2, the functions the K loops call)
        x s = x0 + x1 + e,   |e| <= 2^-23 |x s|   (and e = 0 whenever the residual fits 11 bits),
    x0 the nearest-even f16 of x s, x1 the nearest-even f16 of the EXACT residual -- over every binade the scale can put an operand in
    (the launch scales its operand tensor so that |x s| < 2^15), values with long carry chains and ties, both signs.  Below 2^-14 (f16's
    denormals: values more than 2^29 under the tensor's abs-max) the error is absolute: |e| <= 2^-25.
    An operand beyond the scale's promise (|x s| >= 65520) is +-inf in the first term: inf / nan out, never a silently wrong number."""
    g = torch.Generator(device="cuda").manual_seed(12)
    n = 1 << 20
    mant = torch.randint(0, 1 << 23, (n,), device="cuda", generator=g, dtype=torch.int32)
    expo = torch.randint(127 - 30, 127 + 15, (n,), device="cuda", generator=g, dtype=torch.int32)      # 2^-30 .. 2^15
    sign = torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.int32)
    x = ((sign << 31) | (expo << 23) | mant).view(torch.float32)
    special = torch.tensor([0.0, -0.0, 1.0, -1.0, 32767.998046875, -32767.998046875, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 1.0 + 2 ** -11, 1.0 + 2 ** -11 + 2 ** -23,
                            1.0 + 2 ** -10 - 2 ** -23, 1.0 + 2 ** -12, 3.0 * 2 ** -12 + 1.0, 2 ** -14, 2 ** -14 * (1 + 2 ** -10), 2 ** -24, 2 ** -25, 0.1, 1 / 3.0, 1000.0 / 3.0],
                           device="cuda")
    patterns = ((torch.arange(1 << 16, device="cuda", dtype=torch.int32) << 7) | 0x3F800000).view(torch.float32)     # every 16-bit tail in [1, 2)
    x = torch.cat([special, patterns, x[: n - special.numel() - patterns.numel()]])
    for scale in (1.0, 2.0 ** -7, 2.0 ** 9):
        xs = x.double() * scale
        ok = xs.abs() < 2.0 ** 15
        t = _split2(x, scale)
        total = t[0].double() + t[1].double()
        err = (total - xs).abs()
        normal = ok & (xs.abs() >= 2.0 ** -2)                      # 2^-23 |x s| >= 2^-25, the absolute error of a residual in f16's denormals
        assert bool((err[normal] <= 2.0 ** -23 * xs.abs()[normal]).all()), float((err[normal] / xs.abs()[normal]).max())
        assert bool((err[ok] <= torch.clamp(2.0 ** -23 * xs.abs()[ok], min=2.0 ** -25)).all())
        assert torch.equal(t[0][ok].double(), (x[ok] * scale).to(torch.float16).double())          # first term: the nearest-even f16 (torch's conversion)
        assert bool((t[1].double().abs()[ok] <= 2.0 ** -11 * xs.abs()[ok] + 2.0 ** -25).all())     # second: at most half a last place of the first
        few = ((x[ok].view(torch.int32) & 0x3) == 0) & normal[ok]        # 22 significant bits: always exact
        assert bool((err[ok][few] == 0).all())
        # on average: rms error 0.74 in units of 2^-24 |x s| (uniform significands; a correctly rounded fp32 operation: 0.29)
        rms = float(((err[normal] / (2.0 ** -24 * xs.abs()[normal])) ** 2).mean().sqrt())
        assert rms < 0.8, rms
    big = torch.tensor([70000.0, -70000.0, float("inf"), float("nan")], device="cuda")
    t = _split2(big, 1.0)
    assert bool(torch.isinf(t[0][:3]).all()) and bool(torch.isnan(t[0][3]))


def test_split_filter_terms_carry_the_fp32_winograd_filter():
    """Second operand: pod_wino_filter_transform_split's two terms of every Winograd-domain filter value sum to (the power of two s_u) x
    (the fp32 value pod_wino_filter_transform computes -- same transform arithmetic) to 2^-23, s_u putting the filter's abs-max into
    [2^14, 2^15); the abs-max itself sits in the trailer word."""
    K, C = 64, 32
    g = torch.Generator(device="cuda").manual_seed(4)
    w = torch.randn(K, C, 3, 3, device="cuda", generator=g) * 0.05
    U = WinoConv(w, None, split=False).U.view(-1).double().cpu()                 # fp32 kernel's slab order: [chunk8][24][h][j][4]
    raw = WinoConv(w, None, split=True).U
    n_terms = 2 * 24 * K * C
    assert raw.numel() == n_terms + 8
    amax = float(raw[n_terms:n_terms + 2].view(torch.float32).item())
    assert amax == float(U.abs().max())
    su = 2.0 ** (14 - math.floor(math.log2(amax)))
    Us = raw[:n_terms].view(torch.float16).double().cpu()
    assert 2.0 ** 14 <= float(Us.abs().max()) < 2.0 ** 15
    # split slab: [chunk16][q][kb][term][h][j][e]; fp32 slab: [chunk8][q][h2][j64][4]
    S = Us.view(C // 16, 24, 2, 2, 2, 32, 8).sum(dim=3)                           # [chunk16][q][kb][h][j][e]: channel 16 chunk + 8 h + e, k = 32 kb + j
    S = S.permute(1, 2, 4, 0, 3, 5).reshape(24, 64, C)                             # [q][k][c]
    F32 = U.view(C // 8, 24, 2, 64, 4).permute(1, 3, 0, 2, 4).reshape(24, 64, C) * su    # [q][k][c]: channel 8 chunk + 4 h2 + e
    assert bool(((S - F32).abs() <= torch.clamp(2.0 ** -23 * F32.abs(), min=2.0 ** -25)).all())


def _conv_desc(conv, src, dst, table, n_sets=1, first_block=0, replicas=0, n_splits=0, split_stride=0, bias=True, in_amax=True):
    """A PodWinoConv filled by hand (argument-validation tests)."""
    from pod_compare_amd import amax
    d = hip.PodWinoConv()
    d.blocks, d.n_blocks, d.n_sets, d.C, d.K, d.relu, d.p = table.data_ptr(), int(table.shape[0]), n_sets, conv.C, conv.Kpad, 0, 0.0
    d.n_splits, d.split_stride = n_splits, split_stride
    q = d.sets[0]
    q.in_, q.out, q.Us, q.bias = src.data_ptr(), dst.data_ptr(), conv.U.data_ptr(), (hip.ptr(conv.bias) if bias else None)
    q.in_amax = amax.of(src).data_ptr() if in_amax else None
    q.first_block, q.replicas = first_block, replicas
    return d


def _launch_rc(d):
    import ctypes
    return hip.load().pod_wino_conv3x3_split(ctypes.byref(d), hip.current_stream())


def test_split_kernel_is_deterministic_and_independent_of_the_pipeline_position():
    """The software-pipelined K loop (next chunk's operands made behind the running chunk's MFMAs) must give the same bits whatever C
    is (1, 2, 3, 4, 5, 16 chunks: the first two chunks take the serial path, the rest the pipelined one) -- checked against the same
    convolution with the channels zero-padded to the next multiple: extra zero channels add exact zeros."""
    levels, copies, K = [(23, 40), (6, 10)], 2, 64
    for C in (16, 32, 48, 64, 80, 256):
        w, b, xs = make(levels, copies, C, K, seed=C)
        src, table = flat(xs), block_table(levels, copies, "cuda")
        a = WinoConv(w, b, split=True)(src, torch.empty(src.shape[0], K, device="cuda"), table, relu=True)
        again = WinoConv(w, b, split=True)(src, torch.empty(src.shape[0], K, device="cuda"), table, relu=True)
        assert torch.equal(a, again)
        wp = torch.cat([w, torch.zeros(K, 32, 3, 3, device="cuda")], dim=1)
        sp = torch.cat([src, torch.zeros(src.shape[0], 32, device="cuda")], dim=1).contiguous()
        padded = WinoConv(wp, b, split=True)(sp, torch.empty(src.shape[0], K, device="cuda"), table, relu=True)
        assert torch.equal(a, padded), C


@pytest.mark.parametrize("split", [False, True], ids=["fp32-mfma", "f16x3"])
def test_dropout_mask_is_the_one_pod_bias_act_draws(split):
    """bias + ReLU + dropout in the conv's store == the conv without them followed by pod_bias_act on the same channels-last
    tensor (same Philox counters), bit for bit."""
    levels, copies, C, K = [(20, 28), (5, 7)], 2, 32, 64
    w, b, xs = make(levels, copies, C, K, seed=3)
    src, table = flat(xs), block_table(levels, copies, "cuda")
    fused = WinoConv(w, b, split=split)(src, torch.empty(src.shape[0], K, device="cuda"), table, relu=True, dropout_p=0.3, seed=1234, offset=5 << 34)
    plain = WinoConv(w, None, split=split)(src, torch.empty(src.shape[0], K, device="cuda"), table)
    hip.check(hip.load().pod_bias_act(plain.data_ptr(), b.data_ptr(), None, None, plain.numel(), K, 1, 1, 0.3, 1234, 5 << 34,
                                      hip.current_stream()), "pod_bias_act")
    assert torch.equal(fused, plain)
    dropped = float((fused == 0).float().mean())
    assert 0.5 < dropped < 0.8                                  # ReLU zeroes half, dropout 30 % of the rest


@pytest.mark.parametrize("p", [0.25, 0.0])
@pytest.mark.parametrize("levels,replicas,C,K", [([(20, 28), (5, 7)], 5, 32, 64), ([(96, 168), (12, 21), (6, 11)], 19, 64, 128), ([(17, 33)], 127, 16, 64)])
def test_replicas_from_the_store_pass_equal_conv_then_expand_dropout(levels, replicas, C, K, p):
    """pod_wino_conv3x3_split_replicas (the first conv of an MC-dropout subnet writing the runs' masked copies itself) == the conv
    without dropout followed, level by level, by pod_expand_dropout with offset + (first float of the level) / 8: bit for bit; with
    p = 0 (parity mode: the recorded masks are applied afterwards) plain copies."""
    w, b, xs = make(levels, 1, C, K, seed=11)
    src = flat(xs)
    conv = WinoConv(w, b, split=True)
    offn, off1 = level_pixel_offsets(levels, replicas), level_pixel_offsets(levels, 1)
    fused = torch.full((offn[-1], K), float("nan"), device="cuda")
    conv.replicas(src, fused, block_table(levels, 1, "cuda", out_copies=replicas), replicas, relu=True, dropout_p=p, seed=77, offset=9 << 34)
    y = conv(src, torch.empty(src.shape[0], K, device="cuda"), block_table(levels, 1, "cuda"), relu=True)
    want = torch.full_like(fused, float("nan"))
    lib = hip.load()
    for i, (h, wd) in enumerate(levels):
        hip.check(lib.pod_expand_dropout(y[off1[i]:].data_ptr(), want[offn[i]:].data_ptr(), h * wd * K, replicas, p, 77, (9 << 34) + offn[i] * K // 8, None,
                                         hip.current_stream()), "pod_expand_dropout")
    assert torch.equal(fused, want)
    if p == 0.0:
        for i, (h, wd) in enumerate(levels):
            assert torch.equal(fused[offn[i]:offn[i + 1]].view(replicas, h * wd, K), y[off1[i]:off1[i + 1]].expand(replicas, -1, -1))
    else:
        first = fused[offn[0]:offn[1]].view(replicas, -1)
        assert not torch.equal(first[0], first[1])               # every replica its own mask
    assert _launch_rc(_conv_desc(conv, src, fused, block_table(levels, 1, "cuda"), replicas=128)) == -1        # replicas <= 127
    assert _launch_rc(_conv_desc(conv, src, fused, block_table(levels, 1, "cuda"), in_amax=False)) == -1        # the operand scale is not optional


def test_grouped_launch_equals_the_separate_launches():
    """pod_wino_conv3x3_split_grouped: several convolutions of one shape in one grid -- ordinary layers with dropout, first layers with
    replicas, predictors writing planes -- bit for bit the separate launches, set by set."""
    from pod_compare_amd.wino import grouped_launch
    levels, C = [(20, 28), (9, 13), (5, 7)], 64
    lib = hip.load()
    # (a) two trunk layers with dropout, different image counts, own Philox offsets
    convs, srcs, tabs = [], [], []
    for i, copies in enumerate((3, 5)):
        w, b, xs = make(levels, copies, C, 64, seed=20 + i)
        convs.append(WinoConv(w, b, split=True)); srcs.append(flat(xs)); tabs.append(block_table(levels, copies, "cuda"))
    want = [convs[i](srcs[i], torch.empty(srcs[i].shape[0], 64, device="cuda"), tabs[i], relu=True, dropout_p=0.2, seed=5, offset=(7 + i) << 34) for i in range(2)]
    got = [torch.full_like(t, float("nan")) for t in want]
    grouped_launch([{"conv": convs[i], "src": srcs[i], "dst": got[i], "table": tabs[i], "offset": (7 + i) << 34} for i in range(2)], relu=True, dropout_p=0.2, seed=5)
    assert all(torch.equal(g, w_) for g, w_ in zip(got, want))
    # (b) two first layers on ONE input, each storing its own number of masked replicas
    w0, b0, xs = make(levels, 1, C, 64, seed=30)
    w1, b1, _ = make(levels, 1, C, 64, seed=31)
    firsts, src = [WinoConv(w0, b0, split=True), WinoConv(w1, b1, split=True)], flat(xs)
    reps = (4, 7)
    want = [firsts[i].replicas(src, torch.empty(level_pixel_offsets(levels, reps[i])[-1], 64, device="cuda"), block_table(levels, 1, "cuda", out_copies=reps[i]), reps[i],
                               relu=True, dropout_p=0.3, seed=9, offset=(2 + i) << 34) for i in range(2)]
    got = [torch.full_like(t, float("nan")) for t in want]
    grouped_launch([{"conv": firsts[i], "src": src, "dst": got[i], "table": block_table(levels, 1, "cuda", out_copies=reps[i]), "offset": (2 + i) << 34, "replicas": reps[i]}
                    for i in range(2)], relu=True, dropout_p=0.3, seed=9)
    assert all(torch.equal(g, w_) for g, w_ in zip(got, want))
    # (c) four predictors: K = 63 / 36 / 63 / 40 real channels (one 64-channel slice each), planes out, subsets of the images
    jobs = [(63, 0, 3, 3), (36, 0, 3, 3), (63, 1, 2, 3), (40, 2, 1, 3)]          # (K, first image, image count, output images)
    sets, want = [], []
    for j, (K, first, count, out_copies) in enumerate(jobs):
        w, b, _ = make(levels, 1, C, K, seed=40 + j)
        conv = WinoConv(w, b, split=True)
        table = block_table(levels, count, "cuda", in_copies=3, in_first=first, out_copies=out_copies)
        n_out = level_pixel_offsets(levels, out_copies)[-1] * K
        want.append(conv(srcs[0], torch.zeros(n_out, device="cuda"), table, planes=True))
        sets.append({"conv": conv, "src": srcs[0], "dst": torch.zeros(n_out, device="cuda"), "table": table, "planes": True})
    grouped_launch(sets)
    assert all(torch.equal(s_["dst"], w_) for s_, w_ in zip(sets, want))
    # invalid: five sets, a set list that does not start at block 0
    dst = torch.empty(srcs[0].shape[0], 64, device="cuda")
    assert _launch_rc(_conv_desc(convs[0], srcs[0], dst, tabs[0], n_sets=5)) == -1
    assert _launch_rc(_conv_desc(convs[0], srcs[0], dst, tabs[0], first_block=1)) == -1
    assert _launch_rc(_conv_desc(convs[0], srcs[0], dst, tabs[0])) == 0


def test_head_with_grouped_launches_equals_the_head_with_separate_launches(monkeypatch):
    """The whole head, MC-dropout mode (replicas, masks, skipped last run) and eval mode: POD_GROUPED_HEAD on against off, bit for bit."""
    torch.manual_seed(21)
    model = modeling.ProbabilisticRetinaNet(dropout_rate=0.1, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                            bbox_cov_loss="negative_log_likelihood").cuda().eval()
    for q in model.parameters():
        q.requires_grad_(False)
    feats = [torch.randn(1, 256, h, w, device="cuda") for h, w in ((24, 40), (12, 20), (6, 10))]
    for mc, n, skip in ((True, 5, True), (True, 1, False), (False, 1, False), (False, 3, False)):
        outs = []
        for g in (True, False):
            monkeypatch.setattr(modeling, "GROUPED_HEAD", g)
            model.head._drop_calls = 0
            outs.append(model.head(feats, n, mc_dropout=mc, skip_unused_last_run=skip))
        for a, b in zip(outs[0], outs[1]):
            for ta, tb in zip(a, b):
                assert torch.equal(ta, tb), (mc, n, skip)


@pytest.mark.parametrize("H,W,C,K,splits", [(24, 42, 512, 512, 4), (48, 84, 256, 256, 2), (24, 42, 256, 256, 4), (6, 11, 128, 64, 2), (13, 17, 64, 36, 2)])
def test_small_maps_split_over_the_input_channels(H, W, C, K, splits):
    """pod_wino_conv3x3_split_partial + pod_wino_reduce (backbone convolutions on small maps: res4 / res5 / p4 / p5): the input channels
    cut into ranges, one workgroup set each, partial sums added in a fixed order.  Equal to conv2d within the kernel's usual bound,
    equal to the unsplit launch to rounding, bit-reproducible, and the policy picks a split for the shapes it was made for."""
    g = torch.Generator(device="cuda").manual_seed(H * W + C)
    w = torch.randn(K, C, 3, 3, device="cuda", generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device="cuda", generator=g)
    x = torch.randn(1, C, H, W, device="cuda", generator=g).relu()
    conv = WinoConv(w, b, split=True)
    src = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
    table = block_table([(H, W)], 1, "cuda")
    want = F.conv2d(x, w, b, padding=1).relu()
    outs = []
    for s in (1, splits, splits):
        dst = torch.full((K * H * W,), float("nan"), device="cuda")
        conv.planes_of_one_image(src, dst, table, relu=True, n_splits=s)
        outs.append(dst.view(1, K, H, W))
        assert float((outs[-1] - want).abs().max()) <= TOL * max(1.0, float(want.abs().max()))
    assert torch.equal(outs[1], outs[2])
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-6 * max(1.0, float(want.abs().max()))
    if (H, W, C, K) == (24, 42, 512, 512):
        assert conv.splits_for(int(table.shape[0])) == 4          # res5 of the benchmark frame: 48 workgroups -> 192
    if (H, W, C, K) == (48, 84, 256, 256):
        assert conv.splits_for(int(table.shape[0])) == 2          # res4 / p4: 72 -> 144
    assert WinoConv(w, b, split=False).splits_for(int(table.shape[0])) == 1
    parts = torch.empty(3 * H * W * conv.Kpad, device="cuda")
    assert _launch_rc(_conv_desc(conv, src, parts, table, n_splits=3, split_stride=H * W * conv.Kpad, bias=False)) == -1          # whole super-chunks per split
    assert _launch_rc(_conv_desc(conv, src, parts, table, n_splits=2, split_stride=H * W * conv.Kpad, bias=True)) == -1           # partial sums carry no bias


def test_invalid_arguments_are_rejected():
    lib = hip.load()
    x = torch.zeros(256, 8, device="cuda")
    y = torch.zeros(256, 64, device="cuda")
    u = torch.zeros(24 * 64 * 8, device="cuda")
    t = block_table([(16, 16)], 1, "cuda")
    s = hip.current_stream()
    ok = lib.pod_wino_conv3x3(x.data_ptr(), y.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 8, 64, 0, 0, 0.0, 0, 0, None, s)
    assert ok == 0
    for args in ((x.data_ptr(), y.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 12, 64, 0, 0, 0.0, 0, 0, None, s),      # C % 8
                 (x.data_ptr(), y.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 8, 96, 0, 0, 0.0, 0, 0, None, s),       # K % 64
                 (x.data_ptr(), y.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 8, 192, 0, 0, 0.0, 0, 0, None, s),      # K / 64 not in 1,2,4,8
                 (x.data_ptr(), x.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 8, 64, 0, 0, 0.0, 0, 0, None, s),       # in place
                 (x.data_ptr(), y.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 8, 64, 63, 0, 0.5, 0, 0, None, s),      # planes + dropout
                 (x.data_ptr(), y.data_ptr(), u.data_ptr(), None, t.data_ptr(), 1, 8, 64, 0, 0, 1.0, 0, 0, None, s),       # p = 1
                 (x.data_ptr(), y.data_ptr(), None, None, t.data_ptr(), 1, 8, 64, 0, 0, 0.0, 0, 0, None, s)):
        assert lib.pod_wino_conv3x3(*args) == -1
    assert lib.pod_wino_filter_transform(u.data_ptr(), u.data_ptr(), 64, 12, s) == -1


@pytest.mark.parametrize("cov_type", ["diagonal", "full"])
def test_head_on_the_winograd_kernel_equals_the_miopen_head(cov_type):
    """The whole head (eval mode, so deterministic): every conv on pod_wino_conv3x3 (one launch per layer over all levels)
    against the per-level MIOpen path."""
    torch.manual_seed(11)
    model = modeling.ProbabilisticRetinaNet(dropout_rate=0.1, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                            bbox_cov_loss="negative_log_likelihood", bbox_cov_type=cov_type).cuda().eval()
    for q in model.parameters():
        q.requires_grad_(False)
    for conv in list(model.head.cls_subnet) + list(model.head.bbox_subnet):
        conv.weight.mul_(8.0)                                    # std 0.01 filters would shrink the activations to nothing
        conv.bias.normal_(0.0, 0.1)
    feats = [torch.randn(1, 256, h, w, device="cuda") for h, w in ((23, 40), (12, 20), (6, 10))]
    try:
        modeling.WINO_HEAD = True
        got = model.head(feats, 3, mc_dropout=False)
        modeling.WINO_HEAD = False
        want = model.head(feats, 3, mc_dropout=False)
    finally:
        modeling.WINO_HEAD = True
    for g_l, w_l in zip(got, want):
        for g, w in zip(g_l, w_l):
            assert g.shape == w.shape and g.is_contiguous()
            assert float((g - w).abs().max()) <= TOL * max(1.0, float(w.abs().max()))


def test_mc_dropout_head_shapes_skipped_run_and_statistics():
    """MC mode through the Winograd head: (N, A*K, H, W) planes per level; the skipped last run of cls / cls_var / reg_var is
    zero, box_delta's is not; runs differ (independent masks); the mean over runs stays near the eval-mode output."""
    torch.manual_seed(12)
    model = modeling.ProbabilisticRetinaNet(dropout_rate=0.1, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                            bbox_cov_loss="negative_log_likelihood").cuda().eval()
    for q in model.parameters():
        q.requires_grad_(False)
    for conv in list(model.head.cls_subnet) + list(model.head.bbox_subnet):
        conv.weight.mul_(8.0)
    feats = [torch.randn(1, 256, h, w, device="cuda") for h, w in ((24, 32), (12, 16))]
    n = 6
    cls, delta, cls_var, reg_var = model.head(feats, n, mc_dropout=True, skip_unused_last_run=True)
    for l, f in enumerate(feats):
        assert cls[l].shape == (n, 63, f.shape[2], f.shape[3]) and delta[l].shape == (n, 36, f.shape[2], f.shape[3])
        for t in (cls[l], cls_var[l], reg_var[l]):
            assert float(t[n - 1].abs().max()) == 0.0 and float(t[n - 2].abs().max()) > 0.0
        assert float(delta[l][n - 1].abs().max()) > 0.0
        assert float((delta[l][0] - delta[l][1]).abs().max()) > 0.0
    ev = model.head(feats, 1, mc_dropout=False)
    many = model.head(feats, 48, mc_dropout=True)
    mean, ref = many[1][0].mean(0), ev[1][0][0]
    assert float((mean - ref).abs().mean()) < 0.35 * float(ref.abs().mean()) + 1e-6


def test_bench_frame_launch_equals_conv2d_with_dropout_mask_applied():
    """The launch shape of the benchmark (five FPN levels of the 768x1344 frame, C = K = 256; 3 runs here), dropout on: the kept
    elements equal conv2d x 1/(1-p), the dropped fraction is p of the positive ones, and a second launch with the same
    (seed, offset) is bit-identical while another offset draws another mask."""
    levels, copies, C, K, p = [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)], 3, 256, 256, 0.1
    w, b, xs = make(levels, copies, C, K, seed=21)
    conv, src, table = WinoConv(w, b), flat(xs), block_table(levels, copies, "cuda")
    out = conv(src, torch.empty(src.shape[0], K, device="cuda"), table, relu=True, dropout_p=p, seed=77, offset=3 << 34)
    again = conv(src, torch.empty(src.shape[0], K, device="cuda"), table, relu=True, dropout_p=p, seed=77, offset=3 << 34)
    other = conv(src, torch.empty(src.shape[0], K, device="cuda"), table, relu=True, dropout_p=p, seed=77, offset=4 << 34)
    assert torch.equal(out, again) and not torch.equal(out, other)
    offs = level_pixel_offsets(levels, copies)
    for i, (x, (h, wd)) in enumerate(zip(xs, levels)):
        want = F.conv2d(x, w, b, padding=1).relu() / (1.0 - p)
        got = out[offs[i]:offs[i + 1]].view(copies, h, wd, K).permute(0, 3, 1, 2)
        kept = got != 0
        assert float(((got - want) * kept).abs().max()) <= TOL * max(1.0, float(want.abs().max()))
        pos = want > 1e-3 * float(want.abs().max())
        dropped = float((pos & ~kept).sum()) / float(pos.sum())
        assert abs(dropped - p) < 0.01, dropped


def test_concurrent_streams_give_the_serial_result():
    """Three images in flight on three HIP streams (as bench.py runs them) share the transformed filters and the block tables,
    nothing else: each stream's output equals the one-stream output."""
    levels, copies, C, K = [(45, 80), (23, 40)], 4, 256, 256
    w, b, xs = make(levels, copies, C, K, seed=5)
    conv, table = WinoConv(w, b), block_table(levels, copies, "cuda")
    srcs = [flat([x * (1.0 + 0.25 * j) for x in xs]) for j in range(3)]
    serial = [conv(s, torch.empty(s.shape[0], K, device="cuda"), table, relu=True, dropout_p=0.1, seed=9, offset=j << 34) for j, s in enumerate(srcs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [None] * 3
    for rep in range(4):
        for j, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[j] = conv(srcs[j], torch.empty(srcs[j].shape[0], K, device="cuda"), table, relu=True, dropout_p=0.1, seed=9, offset=j << 34)
        torch.cuda.synchronize()
        for j in range(3):
            assert torch.equal(outs[j], serial[j])


def test_transformed_filters_follow_the_parameters():
    """The head caches pod_wino_filter_transform per conv; an in-place parameter update (checkpoint load, BN fold) must refresh it."""
    torch.manual_seed(13)
    head = modeling.ProbabilisticRetinaNetHead(256, 9, 7, 4, 0.01, 0.0, False, False, 4).cuda().eval()
    for q in head.parameters():
        q.requires_grad_(False)
    feats = [torch.randn(1, 256, 12, 20, device="cuda")]
    before = head(feats, 1)[0][0].clone()
    first = head._wino(head.cls_subnet[0])
    assert head._wino(head.cls_subnet[0]) is first                      # cached while the parameters are untouched
    head.cls_subnet[0].weight.mul_(3.0)                                  # in place: _version changes
    assert head._wino(head.cls_subnet[0]) is not first
    after = head(feats, 1)[0][0]
    modeling.WINO_HEAD = False
    try:
        want = head(feats, 1)[0][0]
    finally:
        modeling.WINO_HEAD = True
    assert float((after - before).abs().max()) > 0.0
    assert float((after - want).abs().max()) <= TOL * max(1.0, float(want.abs().max()))


def test_full_frame_model_forward_winograd_head_equals_miopen_head():
    """BASELINE geometry (1280x720 frame -> 750x1333 -> padded 768x1344, R = 193 374 anchors), eval mode (deterministic): the whole
    model forward with the head on pod_wino_conv3x3 against the same model with the head on MIOpen."""
    from pod_compare_amd import synthetic
    torch.manual_seed(17)
    model = modeling.ProbabilisticRetinaNet(dropout_rate=0.1, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                            bbox_cov_loss="negative_log_likelihood").cuda().eval()
    modeling.fold_frozen_bn(model)
    for q in model.parameters():
        q.requires_grad_(False)
    for conv in list(model.head.cls_subnet) + list(model.head.bbox_subnet):
        conv.weight.mul_(8.0)
    img = modeling.resize_test_image(synthetic.synthetic_frame(3, 720, 1280, device="cuda"))
    try:
        modeling.WINO_HEAD = True
        got = model(img, num_mc_dropout_runs=1)
        modeling.WINO_HEAD = False
        want = model(img, num_mc_dropout_runs=1)
    finally:
        modeling.WINO_HEAD = True
    assert sum(t.shape[0] * 0 + t.shape[2] * t.shape[3] * 9 for t in got.cls) == 193374
    for name in ("cls", "delta", "cls_var", "reg_var"):
        for g, w in zip(getattr(got, name), getattr(want, name)):
            assert g.shape == w.shape
            assert float((g - w).abs().max()) <= 1e-4 * max(1.0, float(w.abs().max())), name      # (the backbone is MIOpen in both)


@pytest.mark.parametrize("split", [False, True], ids=["fp32-mfma", "f16x3"])
@pytest.mark.parametrize("mid,H,W,stride", [(64, 47, 83, 1), (128, 25, 42, 2), (256, 13, 21, 1), (512, 6, 11, 2)])
def test_bottleneck_with_conv2_on_the_winograd_kernel_equals_the_miopen_bottleneck(mid, H, W, stride, split, monkeypatch):
    """detectron2's BottleneckBlock (conv1 1x1 [stride], conv2 3x3, conv3 1x1, shortcut; FrozenBN folded) as modeling.Bottleneck runs it
    since round 3 -- conv1 on MIOpen without bias, its bias + ReLU on the pass that lays the map out channels-last
    (pod_bias_act_to_nhwc), conv2 + bias + ReLU on pod_wino_conv3x3 with NCHW planes out -- against the same block with every
    convolution on MIOpen (POD_WINO_BACKBONE=0's path), the res2 .. res5 channel counts, odd map sizes."""
    from pod_compare_amd import modeling, wino
    monkeypatch.setattr(wino, "SPLIT_BF16", split)
    torch.manual_seed(mid)
    cin = 2 * mid if stride == 2 else 4 * mid
    blk = modeling.Bottleneck(cin, 4 * mid, mid, stride).cuda().eval()
    for m in blk.modules():                      # non-trivial frozen statistics, then fold them into the convs
        if isinstance(m, modeling.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0.0, 0.2); m.running_mean.normal_(0.0, 0.2); m.running_var.uniform_(0.5, 1.5)
    assert modeling.fold_frozen_bn(blk) == (4 if cin != 4 * mid else 3)
    x = torch.randn(1, cin, H * stride, W * stride, device="cuda").relu()
    with torch.no_grad():
        monkeypatch.setattr(modeling, "WINO_BACKBONE", True)
        got = blk(x)
        monkeypatch.setattr(modeling, "WINO_BACKBONE", False)
        want = blk(x)
    assert got.shape == want.shape == (1, 4 * mid, H, W)
    assert float((got - want).abs().max()) <= TOL * max(1.0, float(want.abs().max()))
    assert not torch.equal(got, want)            # (two different convolution kernels did run)
