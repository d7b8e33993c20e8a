"""pod_im2col3x3s2_cl + pod_conv1x1_split as conv3x3 / stride 2 / padding 1 on a channels-last map (conv1x1.Conv3x3S2): FPN's LastLevelP6P7
(detectron2; the last two convolutions of `self.backbone(images.tensor)`, probabilistic_retinanet.py:96-100).  Referees: the patch matrix
against torch's unfold exactly; an fp64 convolution with the per-element bound of the other split kernels; the FPN's MIOpen path end to end."""
import pytest
import torch
import torch.nn.functional as F

from pod_compare_amd import hip, modeling
from pod_compare_amd.conv1x1 import Conv3x3S2

pytestmark = pytest.mark.gpu


def cl(x):            # (1, C, h, w) -> (h * w, C) channels-last buffer
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("h,w,c", [(24, 42, 32), (23, 41, 16), (1, 1, 4), (2, 3, 8), (12, 21, 256), (5, 1, 4)])
@pytest.mark.parametrize("relu", [False, True])
def test_patch_matrix_equals_unfold_exactly(h, w, c, relu):
    x = torch.randn(1, c, h, w, device="cuda")
    ho, wo = Conv3x3S2.out_hw(h, w)
    y = torch.empty(ho * wo, 9 * c, device="cuda")
    hip.check(hip.load().pod_im2col3x3s2_cl(cl(x).data_ptr(), y.data_ptr(), h, w, c, 1 if relu else 0, hip.current_stream()), "pod_im2col3x3s2_cl")
    want = F.unfold(x.relu() if relu else x, kernel_size=3, stride=2, padding=1)              # (1, C * 9, L), row index c * 9 + tap
    assert want.shape[2] == ho * wo
    want = want[0].view(c, 9, ho * wo).permute(2, 1, 0).reshape(ho * wo, 9 * c)               # -> [pixel][tap][c]
    assert torch.equal(y, want)


def test_patch_matrix_rejects_bad_arguments():
    lib, s = hip.load(), hip.current_stream()
    x = torch.randn(6 * 7, 8, device="cuda")
    y = torch.empty(3 * 4, 72, device="cuda")
    assert lib.pod_im2col3x3s2_cl(x.data_ptr(), y.data_ptr(), 6, 7, 8, 0, s) == 0
    assert lib.pod_im2col3x3s2_cl(None, y.data_ptr(), 6, 7, 8, 0, s) == -1
    assert lib.pod_im2col3x3s2_cl(x.data_ptr(), x.data_ptr(), 6, 7, 8, 0, s) == -1
    assert lib.pod_im2col3x3s2_cl(x.data_ptr(), y.data_ptr(), 0, 7, 8, 0, s) == -1
    assert lib.pod_im2col3x3s2_cl(x.data_ptr(), y.data_ptr(), 6, 7, 6, 0, s) == -1             # C % 4
    assert lib.pod_im2col3x3s2_cl(x.data_ptr() + 4, y.data_ptr(), 6, 7, 8, 0, s) == -1         # 16-byte alignment


@pytest.mark.parametrize("h,w,cin,cout", [(24, 42, 2048, 256), (12, 21, 256, 256), (23, 41, 64, 64), (7, 5, 16, 128), (1, 1, 32, 64)])
@pytest.mark.parametrize("relu_input", [False, True])
def test_stride2_conv_equals_an_fp64_convolution_inside_the_fp32_class(h, w, cin, cout, relu_input):
    g = torch.Generator(device="cuda").manual_seed(h * 131 + w + cin)
    wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g)
    x = torch.randn(1, cin, h, w, device="cuda", generator=g) * 1.5
    conv = Conv3x3S2(wt, b)
    y, ho, wo = conv(cl(x), h, w, relu_input=relu_input)
    xin = x.relu() if relu_input else x
    want = F.conv2d(xin.double(), wt.double(), b.double(), stride=2, padding=1)
    assert (ho, wo) == tuple(want.shape[2:]) and tuple(y.shape) == (ho * wo, cout)
    got = y.view(1, ho, wo, cout).permute(0, 3, 1, 2).double()
    assert bool(torch.isfinite(got).all())
    bound = F.conv2d(xin.double().abs(), wt.double().abs(), b.double().abs(), stride=2, padding=1)
    c = float(((got - want).abs() / (2.0 ** -24 * bound)).max())
    ref32 = F.conv2d(xin, wt, b, stride=2, padding=1).double()
    c32 = float(((ref32 - want).abs() / (2.0 ** -24 * bound)).max())
    print("c(im2col + pod_conv1x1_split) = %.2f   c(torch conv2d fp32) = %.2f" % (c, c32))
    assert c <= 16.0                      # (a length-9 Cin fp32 dot product guarantees c <= 9 Cin; the other split kernels measure 2 - 8)
    y2, _, _ = conv(cl(x), h, w, relu_input=relu_input)
    assert torch.equal(y, y2)             # fixed-order partial sums: the same bits every run


def test_fpn_top_levels_equal_the_miopen_path(monkeypatch):
    torch.manual_seed(3)
    fpn = modeling.FPN().cuda().eval()
    with torch.no_grad():
        for m in (fpn.p6, fpn.p7):
            m.bias.copy_(0.1 * torch.randn_like(m.bias))
    for h5, w5 in ((24, 42), (7, 11)):
        feats = []
        for c, k in ((512, 4), (1024, 2), (2048, 1)):
            feats.append((torch.randn(h5 * k * w5 * k, c, device="cuda").relu(), h5 * k, w5 * k))
        with torch.no_grad():
            monkeypatch.setattr(modeling, "HIP_P6P7", True)
            a = fpn.forward_cl(feats)
            monkeypatch.setattr(modeling, "HIP_P6P7", False)
            b = fpn.forward_cl(feats)
        for ta, tb in zip(a[3:], b[3:]):
            assert ta.shape == tb.shape
            assert float((ta - tb).abs().max()) <= 2e-5 * max(1.0, float(tb.abs().max())), (h5, w5)


def test_fpn_levels_land_in_one_buffer_the_head_reads_without_a_copy():
    """FPN.forward_cl writes p3 .. p7 into consecutive slices of one channels-last buffer and maxes their abs-max into one record;
    modeling.levels_channels_last hands the head that buffer (no concatenation) -- and falls back to the concatenation for any other list."""
    from pod_compare_amd import amax
    torch.manual_seed(4)
    fpn = modeling.FPN().cuda().eval()
    h5, w5 = 7, 11
    feats = [(torch.randn(h5 * k * w5 * k, c, device="cuda").relu(), h5 * k, w5 * k) for c, k in ((512, 4), (1024, 2), (2048, 1))]
    with torch.no_grad():
        outs = fpn.forward_cl(feats)
    buf = modeling.levels_channels_last(outs)
    assert buf.data_ptr() == outs[0].data_ptr() and buf is outs[0]._pod_cl_levels
    want = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]) for f in outs])
    assert torch.equal(buf, want)
    rec = amax.of(buf)                                             # the shared record bounds every level (and equals the maximum: 16 slots)
    assert float(rec.max()) == float(want.abs().max())
    other = [f.clone() for f in outs]                              # not the buffer's slices any more: concatenated
    cat = modeling.levels_channels_last(other)
    assert cat.data_ptr() != buf.data_ptr() and torch.equal(cat, want)
    shuffled = [outs[0], outs[2], outs[1], outs[3], outs[4]]       # same objects, another order: not consecutive slices
    assert modeling.levels_channels_last(shuffled).data_ptr() != buf.data_ptr()


def test_channels_last_fpn_equals_the_nchw_fpn_on_odd_sizes():
    """FPN.forward_cl (laterals with the top-down map read at half resolution, p6 / p7 as patch matrices, one output buffer) against
    FPN.forward (torch's interpolate / conv2d on NCHW tensors) on maps whose sizes are odd at every level."""
    torch.manual_seed(6)
    fpn = modeling.FPN().cuda().eval()
    for (h3, w3) in ((27, 45), (24, 42)):
        sizes = [(h3, w3), ((h3 + 1) // 2, (w3 + 1) // 2)]
        sizes.append(((sizes[1][0] + 1) // 2, (sizes[1][1] + 1) // 2))
        nchw = [torch.randn(1, c, h, w, device="cuda").relu() for c, (h, w) in zip((512, 1024, 2048), sizes)]
        with torch.no_grad():
            want = fpn(nchw)
            got = fpn.forward_cl([(t[0].permute(1, 2, 0).reshape(-1, t.shape[1]).contiguous(), t.shape[2], t.shape[3]) for t in nchw])
        for a, b in zip(got, want):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (h3, w3, tuple(a.shape))
