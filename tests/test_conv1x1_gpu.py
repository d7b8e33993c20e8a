"""pod_conv1x1_split (csrc/k13_conv1x1_split.hip): the backbone's / FPN's 1x1 convolutions as a channels-last GEMM with split-operand (round 5: 2-way f16; rounds 3-4: 3 x bf16)
split products.  Referees: torch's conv2d on the same tensors, and an fp64 convolution with the per-element bound of the 3x3 kernels."""
import pytest
import torch
import torch.nn.functional as F

from pod_compare_amd import hip
from pod_compare_amd.conv1x1 import Conv1x1

pytestmark = pytest.mark.gpu

# (Cin, Cout, H_in, W_in, stride): the shapes of a ResNet-50-FPN on a small frame, ragged pixel counts, both channel tilings
SHAPES = [(64, 64, 48, 84, 1), (64, 256, 48, 84, 1), (256, 64, 48, 84, 1), (256, 128, 48, 84, 2), (256, 512, 47, 83, 2), (512, 128, 24, 42, 1),
          (128, 512, 24, 42, 1), (1024, 256, 12, 21, 1), (256, 1024, 12, 21, 1), (2048, 512, 6, 11, 1), (512, 2048, 6, 11, 1), (1024, 2048, 12, 21, 2),
          (2048, 256, 6, 11, 1), (16, 64, 5, 7, 1), (32, 192, 9, 9, 2)]


def make(cin, cout, h, w, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    wt = torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g)
    x = torch.randn(1, cin, h, w, device="cuda", generator=g).relu()
    return wt, b, x


@pytest.mark.parametrize("cin,cout,h,w,stride", SHAPES)
@pytest.mark.parametrize("residual", [False, True])
def test_equals_conv2d_and_stays_inside_the_fp32_class(cin, cout, h, w, stride, residual):
    wt, b, x = make(cin, cout, h, w, cin + cout)
    conv = Conv1x1(wt, b, stride)
    ho, wo = conv.out_hw(h, w)
    xcl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    res = torch.randn(ho * wo, cout, device="cuda") if residual else None
    y = conv(xcl, h, w, relu=True, residual=res)
    pre = F.conv2d(x.double(), wt.double(), b.double(), stride=stride)
    if residual:
        pre = pre + res.double().view(1, ho, wo, cout).permute(0, 3, 1, 2)
    want = pre.relu()
    got = y.view(1, ho, wo, cout).permute(0, 3, 1, 2).double()
    assert got.shape == want.shape and bool(torch.isfinite(got).all())
    bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), stride=stride)
    if residual:
        bound = bound + res.double().abs().view(1, ho, wo, cout).permute(0, 3, 1, 2)
    c = float(((got - want).abs() / (2.0 ** -24 * bound)).max())
    ref32 = F.conv2d(x, wt, b, stride=stride)
    if residual:
        ref32 = ref32 + res.view(1, ho, wo, cout).permute(0, 3, 1, 2)
    c32 = float(((ref32.relu().double() - want).abs() / (2.0 ** -24 * bound)).max())
    print("c(pod_conv1x1_split) = %.2f   c(torch conv2d fp32) = %.2f" % (c, c32))
    assert c <= 8.0                      # a length-Cin fp32 dot product guarantees c <= Cin; MIOpen's fp32 GEMM measures 1.5 - 4


@pytest.mark.parametrize("cin,cout,h,w,stride,splits", [(2048, 512, 24, 42, 1, 4), (1024, 256, 12, 21, 1, 8), (512, 2048, 24, 42, 1, 2), (256, 64, 9, 9, 1, 2)])
def test_split_over_the_input_channels_is_reproducible_and_equal_to_rounding(cin, cout, h, w, stride, splits):
    wt, b, x = make(cin, cout, h, w, 7)
    conv = Conv1x1(wt, b, stride)
    xcl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    res = torch.randn(h * w, cout, device="cuda")
    one = conv(xcl, h, w, relu=True, residual=res, n_splits=1)
    a = conv(xcl, h, w, relu=True, residual=res, n_splits=splits)
    b2 = conv(xcl, h, w, relu=True, residual=res, n_splits=splits)
    assert torch.equal(a, b2)
    assert float((a - one).abs().max()) <= 4e-6 * max(1.0, float(one.abs().max()))
    assert conv.splits_for(h * w) >= 1


def test_invalid_arguments_are_rejected():
    lib = hip.load()
    x = torch.zeros(64, 32, device="cuda")
    y = torch.zeros(64, 64, device="cuda")
    ws = torch.zeros(2 * 64 * 32 + 8, dtype=torch.int16, device="cuda")
    s = hip.current_stream()
    am = torch.zeros(2 * 512, device="cuda")                       # two abs-max records
    ok = lambda *a: lib.pod_conv1x1_split(x.data_ptr(), y.data_ptr(), ws.data_ptr(), None, None, *a, None, 0, am.data_ptr(), am[512:].data_ptr(), s)
    assert lib.pod_conv1x1_split(x.data_ptr(), y.data_ptr(), ws.data_ptr(), None, None, 8, 8, 8, 8, 1, 32, 64, 0, 1, None, 0, None, None, s) == -1     # no in_amax word
    assert lib.pod_conv1x1_filter_split_bytes(64, 32) == ws.numel() * 2
    assert ok(8, 8, 8, 8, 1, 32, 64, 0, 1) == 0
    assert ok(8, 8, 8, 8, 1, 24, 64, 0, 1) == -1        # Cin % 16
    assert ok(8, 8, 8, 8, 1, 32, 96, 0, 1) == -1        # Cout % 64
    assert ok(8, 8, 8, 8, 3, 32, 64, 0, 1) == -1        # stride
    assert ok(8, 8, 4, 8, 1, 32, 64, 0, 1) == -1        # input smaller than the output needs
    assert ok(8, 8, 8, 8, 1, 32, 64, 0, 2) == -1        # split without a partials buffer
    assert lib.pod_conv1x1_filter_split(x.data_ptr(), ws.data_ptr(), 48, 32, s) == -1


def test_channels_last_backbone_equals_the_nchw_backbone():
    """ResNet-50-FPN with every 1x1 convolution on pod_conv1x1_split and the trunk channels-last from the max-pool on (modeling.forward_cl)
    against the round-3 form (MIOpen 1x1 convolutions + element-wise passes, NCHW): the same five feature maps to fp32 rounding through
    ~50 layers, for frames whose maps are ragged (odd sizes, stride-2 convolutions on odd inputs)."""
    from pod_compare_amd import modeling
    torch.manual_seed(0)
    m = modeling.ProbabilisticRetinaNet(cls_var_loss="loss_attenuation", bbox_cov_loss="negative_log_likelihood").cuda().eval()
    g = torch.Generator().manual_seed(1)
    for mod in m.modules():                                   # non-trivial FrozenBN statistics: the fold must carry them into the biases
        if isinstance(mod, modeling.FrozenBatchNorm2d):
            mod.weight.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
            mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
            mod.running_mean.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            mod.running_var.copy_(0.5 + torch.rand(mod.bias.shape, generator=g))
    modeling.fold_frozen_bn(m)
    assert m.bottom_up.cl_eligible() and m.fpn.cl_eligible()
    for hw in ((224, 320), (200, 333), (97, 131)):
        frame = torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, device="cuda")
        x = m.preprocess_image(frame)
        assert m._cl_backbone(x)
        with torch.no_grad():
            cl = m.fpn.forward_cl(m.bottom_up.forward_cl(x))
            ref = m.fpn(m.bottom_up(x))
        for a, b in zip(cl, ref):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), (hw, tuple(a.shape))



@pytest.mark.parametrize("cin,cout,h,w,stride", [(1024, 256, 48, 84, 1), (2048, 512, 24, 42, 1), (512, 128, 96, 168, 1), (512, 1024, 96, 168, 2), (256, 64, 20, 30, 1)])
def test_split_k_inside_the_workgroup_equals_the_cut_over_workgroup_sets_to_the_bit(cin, cout, h, w, stride):
    """waves = 2 / 4 (round 5): the wavefronts of one workgroup share a tile's K range and add their accumulators in LDS in a fixed order --
    the k-steps of a wavefront stay in order, so the result IS that of the same cut over workgroup sets (grid.y partial sums + reduce
    launch), bit for bit, without the partial sums' trip through HBM; reproducible; the library's own choice (waves = 0) is one of them."""
    wt, b, x = make(cin, cout, h, w, 11)
    conv = Conv1x1(wt, b, stride)
    ho, wo = conv.out_hw(h, w)
    xcl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    res = torch.randn(ho * wo, cout, device="cuda")
    outs = {}
    for wv in (1, 2, 4):
        if (cin // 16) % (2 * wv):
            continue
        a = conv(xcl, h, w, relu=True, residual=res, n_splits=1, waves=wv)
        assert torch.equal(a, conv(xcl, h, w, relu=True, residual=res, n_splits=1, waves=wv))
        assert torch.equal(a, conv(xcl, h, w, relu=True, residual=res, n_splits=wv, waves=1)), wv
        outs[wv] = a
    auto = conv(xcl, h, w, relu=True, residual=res, n_splits=1)
    assert any(torch.equal(auto, o) for o in outs.values())
    assert float((outs[1] - outs[max(outs)]).abs().max()) <= 4e-6 * max(1.0, float(outs[1].abs().max()))
    lib = hip.load()
    am = torch.full((512,), 8.0, device="cuda")
    y = torch.empty(ho * wo, cout, device="cuda")
    bad = lambda wv: lib.pod_conv1x1_split(xcl.data_ptr(), y.data_ptr(), conv.Ws.data_ptr(), None, None, ho, wo, h, w, stride, cin, cout, 0, 1, None, wv, am.data_ptr(), None,
                                          hip.current_stream())
    assert bad(3) == -1 and bad(5) == -1 and bad(-1) == -1


@pytest.mark.parametrize("h,w,cin,cout", [(48, 84, 1024, 256), (47, 83, 512, 256), (5, 3, 48, 64), (1, 1, 16, 64), (96, 168, 512, 256), (7, 9, 32, 128)])
def test_half_resolution_residual_equals_the_materialised_upsampling(h, w, cin, cout):
    """POD_C1_RESIDUAL_UP2 (FPN's top-down sum): the residual read at (y >> 1, x >> 1) of the coarser map == the same launch with
    F.interpolate(..., mode="nearest")'s output as a full-resolution residual, bit for bit -- both kernels (whole pairs of k-steps: LDS
    form; an odd count: direct fragments), ragged tiles, odd sizes."""
    import torch.nn.functional as F
    from pod_compare_amd.conv1x1 import Conv1x1
    g = torch.Generator(device="cuda").manual_seed(h * 17 + w + cin)
    wt = torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * 0.05
    b = torch.randn(cout, device="cuda", generator=g)
    x = torch.randn(h * w, cin, device="cuda", generator=g)
    hr, wr = (h + 1) // 2, (w + 1) // 2
    top = torch.randn(hr * wr, cout, device="cuda", generator=g)
    up = F.interpolate(top.view(1, hr, wr, cout).permute(0, 3, 1, 2), size=(h, w), mode="nearest").permute(0, 2, 3, 1).reshape(h * w, cout).contiguous()
    conv = Conv1x1(wt, b, 1)
    for relu in (False, True):
        want = conv(x, h, w, relu=relu, residual=up, n_splits=1)
        got = conv(x, h, w, relu=relu, residual=top, residual_up2=True, n_splits=1)
        assert torch.equal(got, want)
    if (cin // 16) % 2 == 0:
        assert torch.equal(conv(x, h, w, residual=top, residual_up2=True, n_splits=2), conv(x, h, w, residual=up, n_splits=2))     # (split: materialised on the host side)
    lib, s = hip.load(), hip.current_stream()
    y = torch.empty(h * w, cout, device="cuda")
    am = torch.full((512,), 8.0, device="cuda")
    args = lambda res, flags, splits, part: (x.data_ptr(), y.data_ptr(), conv.Ws.data_ptr(), conv.bias.data_ptr(), res, h, w, h, w, 1, cin, cout, flags, splits, part, 0,
                                             am.data_ptr(), None, s)
    assert lib.pod_conv1x1_split(*args(None, 2, 1, None)) == -1                        # the flag without a residual
    if (cin // 16) % 2 == 0:
        part = torch.empty(2, h * w, cout, device="cuda")
        assert lib.pod_conv1x1_split(*args(top.data_ptr(), 2, 2, part.data_ptr())) == -1   # ... or with a split
