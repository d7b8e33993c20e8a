"""GPU edge cases and size-independent properties of the HIP path (eps-replay mode vs the CPU oracle unless noted)."""
import pytest
import torch

from oracle import pod_oracle as po
from pod_compare_amd import hotpath, synthetic
from tests.helpers import assert_close
from tests.test_hip_parity import make_path

pytestmark = pytest.mark.gpu


def hip_vs_oracle(ho, mode, image, out, seed, runs, quirk=True, topk=1000, **kw):
    hp = make_path(ho, topk=topk, quirk=quirk)
    hd = ho.to("cuda")
    det = hp.run(mode, hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=image, out_size=out, eps_fn=synthetic.SeededNormals(seed), **kw)
    p = po.PathParams(num_classes=ho.num_classes, merge_quirk=quirk, topk_candidates=topk)
    rl = [synthetic.to_reference_layout(ho, r) for r in range(runs)]
    ref = po.predict(mode, p, image, out, outputs=rl[0] if runs == 1 else None, run_outputs=rl if runs > 1 else None,
                     eps_fn=synthetic.SeededNormals(seed), **kw)
    m = det.count()
    assert m == len(ref), (m, len(ref))
    assert torch.equal(det.classes[:m].cpu().long(), ref.pred_classes)
    assert_close(det.scores[:m].cpu(), ref.scores, "scores", 2e-6, 1e-7)
    assert_close(det.boxes[:m].cpu(), ref.pred_boxes, "boxes")
    assert_close(det.cov[:m].cpu(), ref.pred_boxes_covariance, "cov")
    return hp, det, ref


@pytest.mark.parametrize("padded,image", [((160, 224), (150, 210)), ((96, 352), (90, 345)), ((224, 96), (224, 96))])
def test_ragged_geometries(padded, image):
    """Feature maps whose H*W is not a multiple of 4 / 64 / 256 on several levels (scalar tail paths of K1, unaligned
    bitmap words), non-square and portrait frames."""
    ho = synthetic.planted_head_outputs(padded, 3, seed=padded[0] + padded[1], num_boxes=6)
    assert any((h * w) % 4 for h, w in ho.shapes)
    hip_vs_oracle(ho, "bayes_od", image, (image[0] * 2, image[1] * 2), seed=5, runs=3)


@pytest.mark.parametrize("K", [1, 3, 12])
def test_other_class_counts(K):
    ho = synthetic.planted_head_outputs((128, 160), 2, seed=40 + K, num_boxes=6, num_classes=K)
    hip_vs_oracle(ho, "anchor_statistics", (120, 150), (120, 150), seed=6, runs=2)


@pytest.mark.parametrize("runs", [2, 7, 17])
def test_run_counts(runs):
    """N = 2 (quirk: only run 0 is ever merged), odd N, N > one load batch."""
    ho = synthetic.planted_head_outputs((128, 160), runs, seed=70 + runs, num_boxes=6)
    hip_vs_oracle(ho, "standard_nms", (120, 150), (240, 300), seed=7, runs=runs)


def test_small_topk_truncates_every_level():
    ho = synthetic.planted_head_outputs((192, 256), 1, seed=90, num_boxes=10, mode="worst")
    hp, det, ref = hip_vs_oracle(ho, "standard_nms", (192, 256), (192, 256), seed=8, runs=1, topk=37)
    assert hp.sel_count.cpu().tolist() == [37, 37, 37, 37, 36]


def test_more_than_2048_candidates_per_level_radix_select_path():
    """worst-case scores: 6912 anchors of p3 all pass the threshold -> K2's radix-select path; max_detections truncation."""
    ho = synthetic.planted_head_outputs((192, 256), 1, seed=91, num_boxes=0, mode="worst", with_cls_var=False, with_reg_var=False)
    hp, det, ref = hip_vs_oracle(ho, "standard_nms", (192, 256), (192, 256), seed=9, runs=1)
    A, K = ho.num_anchors, ho.num_classes
    n_pass = int((torch.sigmoid(ho.cls[0][0]).view(A, K, -1).amax(1) > 0.05).sum())       # p3 anchors above the threshold
    assert n_pass > 2048 and hp.sel_count.cpu().tolist()[0] == 1000
    assert det.count() >= 90 and int(hp.n_keep.item()) == 100   # a few boxes clip to empty
    assert int(hp.counters.abs().sum().item()) == 0             # K2 consumed the per-level counts


def test_zero_candidates_every_head_type():
    """No anchor above the threshold: empty result for plain AND variance-head models (the reference raises for the
    latter, SURVEY Q12 -- documented deviation)."""
    for var in (False, True):
        ho = synthetic.planted_head_outputs((128, 160), 2, seed=3, num_boxes=0, with_cls_var=var, with_reg_var=var).to("cuda")
        hp = make_path(ho)
        for mode in ("standard_nms", "anchor_statistics") + (("bayes_od",) if var else ()):
            det = hp.run(mode, ho.cls, ho.delta, ho.cls_var, ho.reg_var, image_size=(120, 150), out_size=(120, 150))
            assert det.count() == 0


def test_degenerate_zero_area_centre_falls_back():
    """dw -> -inf collapses a box to zero width: self-IoU is 0, the BayesOD cluster is empty; the reference would hit a
    singular matrix, this build returns the centre's own estimate and K7 drops the empty box."""
    ho = synthetic.planted_head_outputs((128, 160), 1, seed=11, num_boxes=4)
    # force the highest-scoring anchor of p4 to a huge negative dw in every run
    lvl = 1
    a_k = ho.cls[lvl][0]
    flat = int(a_k.reshape(-1).argmax())
    plane, hw = divmod(flat, a_k.shape[1] * a_k.shape[2])
    a = plane // ho.num_classes
    h, w = divmod(hw, a_k.shape[2])
    ho.delta[lvl][:, a * 4 + 2, h, w] = -80.0
    hd = ho.to("cuda")
    hp = make_path(ho)
    det = hp.run("bayes_od", hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=(120, 150), out_size=(120, 150),
                 eps_fn=synthetic.SeededNormals(1))
    m = det.count()
    assert m > 0 and bool(torch.isfinite(det.boxes[:m]).all()) and bool(torch.isfinite(det.cov[:m]).all())
    assert bool(((det.boxes[:m, 2] - det.boxes[:m, 0]) > 0).all())


def test_covariances_are_symmetric_positive_definite_at_full_size():
    """BASELINE size (R = 193374, N = 10), native RNG: <= 100 detections, boxes inside the frame, SPD covariances."""
    ho = synthetic.planted_head_outputs((768, 1344), 10, seed=123, num_boxes=24, device="cuda")
    hp = make_path(ho)
    for mode in ("bayes_od", "anchor_statistics", "standard_nms"):
        det = hp.run(mode, ho.cls, ho.delta, ho.cls_var, ho.reg_var, image_size=(750, 1333), out_size=(720, 1280))
        m = det.count()
        assert 0 < m <= 100
        b, c = det.boxes[:m], det.cov[:m]
        assert bool((b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 2] <= 1280).all() and (b[:, 3] <= 720).all())
        assert float((c - c.transpose(1, 2)).abs().max()) <= 1e-5 * float(c.abs().max())
        assert float(torch.linalg.eigvalsh(c.double().cpu()).min()) > 0
        s = det.scores[:m]
        assert bool((s[:-1] >= s[1:]).all()) or mode == "anchor_statistics"      # sorted by NMS score


def test_output_scale_equivariance():
    """Rescaling the output resolution by (sx, sy) scales boxes by S and covariances by S cov S (IU:394-424)."""
    ho = synthetic.planted_head_outputs((192, 256), 4, seed=21, num_boxes=8)
    hd = ho.to("cuda")
    hp = make_path(ho)
    kw = dict(image_size=(180, 250), eps_fn=None, draw_id=1)      # the same Philox draws in both calls
    d1 = hp.run("bayes_od", hd.cls, hd.delta, hd.cls_var, hd.reg_var, out_size=(180, 250), **kw)
    b1, c1, m1 = d1.boxes.clone(), d1.cov.clone(), d1.count()
    d2 = hp.run("bayes_od", hd.cls, hd.delta, hd.cls_var, hd.reg_var, out_size=(540, 500), **kw)
    assert d2.count() == m1
    s = torch.tensor([2.0, 3.0, 2.0, 3.0], device="cuda")
    inside = (b1[:m1, 0] > 0) & (b1[:m1, 1] > 0) & (b1[:m1, 2] < 250) & (b1[:m1, 3] < 180)     # clipping is not equivariant
    assert torch.allclose(d2.boxes[:m1][inside], b1[:m1][inside] * s, rtol=1e-6, atol=1e-4)
    assert torch.allclose(d2.cov[:m1], c1[:m1] * s[:, None] * s[None, :], rtol=1e-5, atol=1e-7)


def test_threshold_monotonicity():
    """Raising the score threshold can only remove candidates (native mode, same Philox draws)."""
    ho = synthetic.planted_head_outputs((192, 256), 3, seed=22, num_boxes=8, mode="worst").to("cuda")
    keys = []
    for thr in (0.05, 0.5, 0.9):
        params = hotpath.PathParams(score_thresh=thr)
        hp = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=3, has_cls_var=True, cov_dims=4, device="cuda")
        hp.candidates(ho.cls, ho.delta, ho.cls_var, ho.reg_var, None)
        n = int(hp.n_total.item())
        keys.append(set(zip(hp.cand_level[:n].cpu().tolist(), hp.cand_anchor_idx[:n].cpu().tolist())))
    assert keys[2] <= keys[1] <= keys[0] and len(keys[2]) < len(keys[0])
