"""Pins the CPU oracle (oracle/pod_oracle.py) against outputs of the reference itself
(tests/golden, written by oracle/make_golden.py in the build container)."""
import json

import numpy as np
import pytest
import torch

from oracle import pod_oracle as po
from pod_compare_amd import synthetic
from tests.helpers import GOLDEN, Golden, assert_close, fixture_id, fixture_paths

ALL = fixture_paths()
SMALL = [p for p in ALL if "/full_" not in p]


def params_of(g: Golden) -> po.PathParams:
    return po.PathParams(topk_candidates=g.meta["topk"])


def oracle_run(g: Golden):
    s = g.spec
    ho = g.head_outputs()
    rec = po.RecordEps(g.eps_source())
    p = params_of(g)
    img, out = tuple(g.meta["image"]), tuple(g.meta["out"])
    runs = [synthetic.to_reference_layout(ho, r) for r in range(s["runs"])]
    if s.get("post_nms"):
        det = po.predict_post_nms_ensemble(p, img, out, runs, rec)
    elif s["runs"] > 1:
        det = po.predict(s["mode"], p, img, out, run_outputs=runs, eps_fn=rec,
                         box_merge_mode=s.get("box_merge", "bayesian_inference"), cls_merge_mode=s.get("cls_merge", "max_score"))
    else:
        det = po.predict(s["mode"], p, img, out, outputs=runs[0], eps_fn=rec,
                         box_merge_mode=s.get("box_merge", "bayesian_inference"), cls_merge_mode=s.get("cls_merge", "max_score"))
    return det, rec


@pytest.mark.parametrize("path", ALL, ids=fixture_id)
def test_oracle_reproduces_reference(path):
    g = Golden(path)
    det, rec = oracle_run(g)
    g.check_eps(rec.tensors)
    # integer / index outputs: exact
    assert torch.equal(det.pred_classes, g.t("pred_classes"))
    assert det.pred_boxes.shape == g.t("pred_boxes").shape
    # same torch/numpy ops in the same order: bit-identical on the host that wrote the fixture; torch's
    # SIMD exp differs in the last bit between AVX2 and AVX-512 hosts, hence a 2-ulp bound, not torch.equal
    assert_close(det.scores, g.t("scores"), "scores", rtol=3e-7, atol=1e-9)
    assert_close(det.pred_cls_probs, g.t("pred_cls_probs"), "probs", rtol=3e-7, atol=1e-9)
    assert_close(det.pred_boxes, g.t("pred_boxes"), "boxes", rtol=1e-6, atol=1e-6)
    assert_close(det.pred_boxes_covariance, g.t("pred_boxes_covariance"), "cov", rtol=1e-5, atol=1e-6)
    cat_map = {i: i + 1 for i in range(7)}
    js = po.detections_to_json(det, g.meta["seed"], cat_map)
    ref = json.loads(str(g.z["json"]))
    assert len(js) == len(ref)
    for a, b in zip(js, ref):
        assert a["image_id"] == b["image_id"] and a["category_id"] == b["category_id"]
        assert_close(a["bbox"], b["bbox"], "json bbox", 1e-6, 1e-6)
        assert_close(a["bbox_covar"], b["bbox_covar"], "json cov", 1e-5, 1e-6)


@pytest.mark.parametrize("path", [p for p in ALL if "post_nms" not in p], ids=fixture_id)
def test_oracle_indices_match_reference(path):
    """top-k anchor index sequences, candidate lists and NMS keep lists: exact."""
    g = Golden(path)
    s = g.spec
    ho = g.head_outputs()
    runs = [synthetic.to_reference_layout(ho, r) for r in range(s["runs"])]
    p = params_of(g)
    aw = po.anchorwise_inference(None if s["runs"] > 1 else runs[0], p, runs if s["runs"] > 1 else None, g.eps_source())
    # per-level top-k indices (before the score-threshold filter) are prefixes of the recorded topk
    off = 0
    for lvl, cnt in enumerate(aw.level_counts):
        ref_top = g.t("topk_%d" % lvl)
        assert torch.equal(aw.anchor_idx[off:off + cnt], ref_top[:cnt]), "level %d" % lvl
        off += cnt
    assert torch.equal(aw.classes, g.t("aw0_cls"))
    assert_close(aw.scores, g.t("aw0_prob"), "aw scores", rtol=3e-7, atol=1e-9)
    assert_close(aw.boxes, g.t("aw0_boxes"), "aw boxes", 1e-6, 1e-6)
    if aw.cov is not None:
        assert_close(aw.cov, g.t("aw0_cov"), "aw cov", 1e-5, 1e-6)
    keep = po.class_aware_nms(aw.boxes, aw.scores, aw.classes, p.nms_thresh)
    assert torch.equal(keep, g.t("nms_keep_0"))


def test_unit_functions():
    """Separately callable reference functions; inputs stored verbatim in the fixture.  Floats are held to
    a few ulp (bit-identical on the host that wrote the fixture), integers exactly."""
    z = np.load(GOLDEN + "/unit_functions.npz")
    t = lambda k: torch.from_numpy(z[k])
    tight = dict(rtol=2e-6, atol=1e-7)
    assert_close(po.cholesky_from_head(t("chol_in4")), t("chol_out4"), "chol4", **tight)
    assert_close(po.cholesky_from_head(t("chol_in10")), t("chol_out10"), "chol10", **tight)
    m, c = po.mean_covariance(t("mc_samples"))
    assert_close(m, t("mc_mean"), "mean", **tight)
    assert_close(c, t("mc_cov"), "cov", rtol=1e-5, atol=1e-6)
    m, c = po.mean_covariance(list(t("mcl_samples")))
    assert_close(m, t("mcl_mean"), "mean(list)", **tight)
    assert_close(c, t("mcl_cov"), "cov(list)", rtol=1e-5, atol=1e-6)
    assert_close(po.decode_sample_boxes(t("sd_deltas"), t("sd_anchors")), t("sd_out"), "sample decode", **tight)
    for mode in ("bayesian_inference", "covariance_intersection"):
        fm, fc = po.bayes_fuse(z["bf_means"], z["bf_covs"], mode)
        assert_close(np.squeeze(fm), z["bf_mean_" + mode], "fused mean", rtol=1e-5, atol=1e-5)
        assert_close(fc, z["bf_cov_" + mode], "fused cov", rtol=1e-5, atol=1e-6)
    det = po.Detections((180, 250), t("pp_boxes"), t("pp_scores"), t("pp_classes"), t("pp_probs"), t("pp_cov"))
    out = po.finalize(det, 173, 240)
    assert torch.equal(out.pred_classes, t("pp_out_classes"))
    assert_close(out.pred_boxes, t("pp_out_boxes"), "pp boxes", **tight)
    assert_close(out.scores, t("pp_out_scores"), "pp scores", **tight)
    assert_close(out.pred_cls_probs, t("pp_out_probs"), "pp probs", **tight)
    assert_close(out.pred_boxes_covariance, t("pp_out_cov"), "pp cov", **tight)
    assert_close(po.cov_xyxy_to_xywh(out.pred_boxes_covariance), t("pp_xywh_cov"), "xywh cov", rtol=1e-5, atol=1e-6)
    js = po.detections_to_json(out, 77, {i: i + 1 for i in range(6)})
    ref = json.loads(str(z["pp_json"]))
    assert [r["category_id"] for r in js] == [r["category_id"] for r in ref]
    for a, b in zip(js, ref):
        assert_close(a["bbox"], b["bbox"], "json bbox", **tight)
        assert_close(a["bbox_covar"], b["bbox_covar"], "json cov", rtol=1e-5, atol=1e-6)


def test_merge_quirk():
    """1..10 -> 4.6 with the reference's loop, 5.5 for the true mean."""
    xs = [torch.full((3,), float(i)) for i in range(1, 11)]
    assert torch.allclose(po.merge_runs(xs, True), torch.full((3,), 4.6))
    assert torch.allclose(po.merge_runs(xs, False), torch.full((3,), 5.5))
