"""Row a1 on the GPU: the build's head with every convolution on pod_wino_conv3x3 (K11) or pod_wino_conv3x3_split (K12), held to
the fixtures the REFERENCE's ProbabilisticRetinaNet / ProbabilisticRetinaNetHead produced (oracle/make_golden_model.py;
PR:95-112, 335-361, 365-537) -- eval mode and the MC branch on the reference's recorded dropout masks (`dropout_replay`: the
masks replace the kernels' Philox draws, everything else is the production launch sequence: one launch per layer over all
levels and copies, predictors on sub-ranges of the copies).  Bound: fp32 Winograd through five layers against an fp32 direct
convolution through five layers, 1e-4 of the tensor's scale (measured: see profiles/r04_parity_errors.md)."""
import pytest
import torch

from oracle import model_fixture as mf
from pod_compare_amd import modeling, wino
from tests.head_fixture import EVAL_NAMES, FIELDS, VARIANT_NAMES, HeadFixture, planes_to_reference

pytestmark = pytest.mark.gpu
TOL = 1e-4
KERNELS = [False, True]
IDS = ["K11-fp32-mfma", "K12-f16x3"]


@pytest.fixture(params=KERNELS, ids=IDS)
def split(request):
    old = wino.SPLIT_BF16
    wino.SPLIT_BF16 = request.param
    yield request.param
    wino.SPLIT_BF16 = old


def close(got, want, what):
    got, want = got.cpu().double(), want.double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = float((got - want).abs().max())
    assert err <= TOL * max(1.0, float(want.abs().max())), (what, err)


def uses_wino(head, split):
    conv = modeling.wino_of(head.cls_subnet[1])
    return conv.split == split


@pytest.mark.parametrize("variant", VARIANT_NAMES)
def test_eval_mode_wino_head_equals_the_reference_head(variant, split):
    fx = HeadFixture(variant)
    head = fx.build_head("cuda")
    feats = [f.cuda() for f in fx.features()]
    with torch.no_grad():
        outs = head(feats, 1, mc_dropout=False)
    assert uses_wino(head, split)
    for field, out in zip(FIELDS, outs):
        if out is None:
            assert not fx.present(field)
            continue
        for l in range(len(fx.levels)):
            close(out[l], fx.t("eval_%s_l%d" % (EVAL_NAMES[field], l)), "%s level %d" % (field, l))
            close(planes_to_reference(out[l], fx.per_anchor(field)), fx.t("eval_%s_l%d" % (field, l)), "raw %s level %d" % (field, l))


@pytest.mark.parametrize("variant", [v for v in VARIANT_NAMES if mf.VARIANTS[v]["dropout_rate"] > 0])
@pytest.mark.parametrize("skip_last", [False, True])
def test_mc_dropout_wino_head_equals_the_reference_on_its_recorded_masks(variant, skip_last, split):
    fx = HeadFixture(variant)
    head = fx.build_head("cuda")
    n = fx.runs
    m = n - 1 if skip_last else n
    head.dropout_replay = fx.replay(m)
    feats = [f.cuda() for f in fx.features()]
    with torch.no_grad():
        outs = head(feats, n, mc_dropout=True, skip_unused_last_run=skip_last)
    assert uses_wino(head, split)
    for field, out in zip(FIELDS, outs):
        if out is None:
            assert not fx.present(field)
            continue
        valid = n if (field == "box_delta" or not skip_last) else m
        for l in range(len(fx.levels)):
            got = planes_to_reference(out[l], fx.per_anchor(field))
            want = fx.t("mc_%s_l%d" % (field, l))
            close(got[:valid], want[:valid], "%s level %d" % (field, l))
            if valid < n:
                assert float(out[l][valid:].abs().max()) == 0.0


def test_miopen_head_path_equals_the_reference_too():
    """The per-level MIOpen path of the head (WINO_HEAD off: what runs when a shape is not tileable) on the same fixture."""
    fx = HeadFixture("reg_cls_var_dropout")
    head = fx.build_head("cuda")
    head.dropout_replay = fx.replay()
    feats = [f.cuda() for f in fx.features()]
    old = modeling.WINO_HEAD
    modeling.WINO_HEAD = False
    try:
        with torch.no_grad():
            outs = head(feats, fx.runs, mc_dropout=True)
    finally:
        modeling.WINO_HEAD = old
    for field, out in zip(FIELDS, outs):
        for l in range(len(fx.levels)):
            close(planes_to_reference(out[l], fx.per_anchor(field)), fx.t("mc_%s_l%d" % (field, l)), "%s level %d" % (field, l))
