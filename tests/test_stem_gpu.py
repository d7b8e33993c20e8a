"""pod_stem7x7_split + pod_maxpool3x3s2_cl (csrc/k14_stem_conv.hip): the ResNet stem (7x7 / stride 2 convolution, FrozenBN folded, ReLU) and
its max-pool, channels-last out.  Referees: an fp64 convolution with the per-element bound of the other convolution kernels; torch's
max_pool2d exactly; the model's NCHW stem path end to end."""
import pytest
import torch
import torch.nn.functional as F

from pod_compare_amd import hip, modeling
from pod_compare_amd.conv1x1 import Stem7x7, maxpool3x3s2_cl

pytestmark = pytest.mark.gpu


def make(h, w, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    wt = torch.randn(64, 3, 7, 7, device="cuda", generator=g) * (2.0 / 147) ** 0.5
    b = torch.randn(64, device="cuda", generator=g)
    x = torch.randn(1, 3, h, w, device="cuda", generator=g) * 1.5
    return wt, b, x


@pytest.mark.parametrize("h,w", [(64, 96), (96, 160), (17, 33), (1, 1), (7, 250), (224, 224), (768, 1344)])
@pytest.mark.parametrize("relu", [True, False])
def test_stem_equals_an_fp64_convolution_inside_the_fp32_class(h, w, relu):
    wt, b, x = make(h, w, h * 7 + w)
    y, ho, wo = Stem7x7(wt, b)(x, relu=relu)
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=3)
    assert (ho, wo) == tuple(want.shape[2:]) and tuple(y.shape) == (ho * wo, 64)
    want = want.relu() if relu else want
    got = y.view(1, ho, wo, 64).permute(0, 3, 1, 2).double()
    assert bool(torch.isfinite(got).all())
    bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), stride=2, padding=3)
    c = float(((got - want).abs() / (2.0 ** -24 * bound)).max())
    ref32 = F.conv2d(x, wt, b, stride=2, padding=3)
    c32 = float((((ref32.relu() if relu else ref32).double() - want).abs() / (2.0 ** -24 * bound)).max())
    print("c(pod_stem7x7_split) = %.2f   c(torch conv2d fp32) = %.2f" % (c, c32))
    assert c <= 8.0                      # (a length-147 fp32 dot product guarantees c <= 147)


def test_stem_is_deterministic_and_rejects_bad_arguments():
    wt, b, x = make(50, 70, 3)
    stem = Stem7x7(wt, b)
    a, ho, wo = stem(x)
    assert torch.equal(a, stem(x)[0])
    lib = hip.load()
    y = torch.empty(ho * wo, 64, device="cuda")
    s = hip.current_stream()
    am = torch.full((512,), 8.0, device="cuda")
    assert lib.pod_stem7x7_split(x.data_ptr(), 0, 50, 70, None, None, y.data_ptr(), stem.Ws.data_ptr(), stem.bias.data_ptr(), 50, 70, 1, None, None, s) == -1   # no in_amax word
    args = lambda x_ptr, y_ptr, h, w, hi=50, wi=70, mean=None, std=None: (x_ptr, 0, hi, wi, mean, std, y_ptr, stem.Ws.data_ptr(), stem.bias.data_ptr(), h, w, 1, am.data_ptr(), None, s)
    assert lib.pod_stem7x7_split(*args(x.data_ptr(), y.data_ptr(), 0, 70)) == -1
    assert lib.pod_stem7x7_split(*args(x.data_ptr(), x.data_ptr(), 50, 70)) == -1
    assert lib.pod_stem7x7_split(*args(None, y.data_ptr(), 50, 70)) == -1
    assert lib.pod_stem7x7_split(*args(x.data_ptr(), y.data_ptr(), 50, 70, hi=51)) == -1                     # the frame is larger than its padded extent
    assert lib.pod_stem7x7_split(*args(x.data_ptr(), y.data_ptr(), 50, 70, mean=stem.bias.data_ptr())) == -1   # mean without std
    assert lib.pod_maxpool3x3s2_cl(y.data_ptr(), y.data_ptr(), ho, wo, 64, s) == -1
    assert lib.pod_maxpool3x3s2_cl(y.data_ptr(), a.data_ptr(), ho, wo, 6, s) == -1


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
@pytest.mark.parametrize("h,w", [(200, 328), (97, 131), (720, 1280)])
def test_normalise_and_pad_on_load_equal_preprocess_then_stem(h, w, dtype):
    """The frame normalised ((x - mean) / std) and zero-padded to a multiple of 32 inside the kernel's patch load == the model's
    preprocess_image followed by the stem on the padded tensor: the same fp32 subtraction and division, so bit for bit whenever the two
    calls scale their operand by the same power of two -- forced here by handing the second call the first one's abs-max record (the
    fused call bounds the normalised frame by (max |x| + max |mean|) / min |std|, the other measures it: another power of two moves the
    f16 split's roundings, i.e. the result in its last bits)."""
    from pod_compare_amd import amax
    from pod_compare_amd import anchors as A
    wt, b, _ = make(8, 8, 1)
    stem = Stem7x7(wt, b)
    frame = torch.randint(0, 256, (3, h, w), dtype=torch.uint8, device="cuda").to(dtype)
    mean, std = torch.tensor([103.53, 116.28, 123.675], device="cuda"), torch.tensor([1.0, 57.375, 58.395], device="cuda")
    ph, pw = A.padded_size(h, w)
    x = F.pad((frame.float() - mean.view(3, 1, 1)) / std.view(3, 1, 1), (0, pw - w, 0, ph - h)).unsqueeze(0).contiguous()
    got, ho2, wo2 = stem(frame, mean=mean, std=std, padded_hw=(ph, pw))
    loose, ho, wo = stem(x)
    assert (ho, wo) == (ho2, wo2) and float((got - loose).abs().max()) <= 4e-6 * max(1.0, float(loose.abs().max()))
    amax.attach(x, stem._input_bound(frame, mean, std))
    want, _, _ = stem(x)
    assert torch.equal(got, want)


@pytest.mark.parametrize("h,w,c", [(48, 84, 64), (47, 83, 64), (1, 1, 4), (2, 5, 8), (384, 672, 64)])
def test_channels_last_max_pool_equals_torch_exactly(h, w, c):
    x = torch.randn(h * w, c, device="cuda")
    y, hp, wp = maxpool3x3s2_cl(x, h, w)
    want = F.max_pool2d(x.view(1, h, w, c).permute(0, 3, 1, 2), kernel_size=3, stride=2, padding=1)
    assert (hp, wp) == tuple(want.shape[2:])
    assert torch.equal(y.view(1, hp, wp, c).permute(0, 3, 1, 2), want)


def test_backbone_with_the_hip_stem_equals_the_backbone_with_the_miopen_stem(monkeypatch):
    """The channels-last trunk (res3 .. res5) behind pod_stem7x7_split + pod_maxpool3x3s2_cl against the same trunk behind MIOpen's stem,
    pod_bias_act, torch's max-pool and the transposing copy, on ragged frames."""
    torch.manual_seed(5)
    m = modeling.ProbabilisticRetinaNet().cuda().eval()
    g = torch.Generator().manual_seed(2)
    for mod in m.modules():                                   # non-trivial FrozenBN statistics: the fold must carry them into the biases
        if isinstance(mod, modeling.FrozenBatchNorm2d):
            mod.weight.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
            mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
            mod.running_mean.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            mod.running_var.copy_(0.5 + torch.rand(mod.bias.shape, generator=g))
    modeling.fold_frozen_bn(m)
    for hw in ((200, 328), (97, 131)):
        x = m.preprocess_image(torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, device="cuda"))
        with torch.no_grad():
            monkeypatch.setattr(modeling, "HIP_STEM", True)
            a = m.bottom_up.forward_cl(x)
            monkeypatch.setattr(modeling, "HIP_STEM", False)
            b = m.bottom_up.forward_cl(x)
        for (ta, ha, wa), (tb, hb, wb) in zip(a, b):
            assert (ha, wa) == (hb, wb) and ta.shape == tb.shape
            assert float((ta - tb).abs().max()) <= 2e-4 * max(1.0, float(tb.abs().max())), hw
