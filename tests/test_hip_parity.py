"""GPU parity: the HIP hot path (through the C ABI) against the golden vectors produced by the
reference and against the CPU oracle on the same seeded inputs, in eps-replay mode.

Bar (BASELINE.json north_star): anchor indices / NMS keep lists bit-exact; box means and
covariances within 1e-4 (|a-b| <= 1e-4 * max(1, |b|))."""
import numpy as np
import pytest
import torch

from oracle import pod_oracle as po
from pod_compare_amd import hotpath, synthetic
from tests.helpers import Golden, assert_close, fixture_id, fixture_paths

pytestmark = pytest.mark.gpu

PRE_NMS = [p for p in fixture_paths() if "post_nms" not in p]
SMALL = [p for p in PRE_NMS if "/full_" not in p]


def make_path(ho: synthetic.HeadOutputs, topk=1000, quirk=True) -> hotpath.HotPath:
    params = hotpath.PathParams(num_classes=ho.num_classes, num_anchors=ho.num_anchors, topk_candidates=topk, merge_quirk=quirk)
    cov_dims = 0 if ho.reg_var is None else ho.reg_var[0].shape[1] // ho.num_anchors
    return hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=ho.num_runs, has_cls_var=ho.cls_var is not None,
                           cov_dims=cov_dims, device="cuda")


def canonical_perm(idx: torch.Tensor, score: torch.Tensor, counts) -> torch.Tensor:
    """torch.topk leaves the order of EXACTLY equal fp32 scores unspecified; this build's
    stated convention is lower anchor index first.  Returns `perm` such that ref[perm] lists the
    reference's candidates in that convention: a permutation inside groups of bit-equal scores of one
    level, the identity everywhere else."""
    perm = torch.arange(idx.numel())
    off = 0
    for cnt in counts:
        i = off
        while i < off + cnt:
            j = i + 1
            while j < off + cnt and score[j] == score[i]:
                j += 1
            if j - i > 1:
                perm[i:j] = i + torch.sort(idx[i:j])[1]
            i = j
        off += cnt
    return perm


def run_hip(g: Golden):
    ho = g.head_outputs()
    hp = make_path(ho, g.meta["topk"])
    hd = ho.to("cuda")
    s = g.spec
    det = hp.run(s["mode"], hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=tuple(g.meta["image"]),
                 out_size=tuple(g.meta["out"]), eps_fn=g.eps_source(),
                 box_merge_mode=s.get("box_merge", "bayesian_inference"), cls_merge_mode=s.get("cls_merge", "max_score"))
    torch.cuda.synchronize()
    return hp, det


@pytest.mark.parametrize("path", PRE_NMS, ids=fixture_id)
def test_hip_matches_reference_golden(path):
    g = Golden(path)
    hp, det = run_hip(g)
    m = det.count()
    ref_boxes = g.t("pred_boxes")
    assert m == ref_boxes.shape[0]
    assert torch.equal(det.classes[:m].cpu().long(), g.t("pred_classes"))
    assert_close(det.scores[:m].cpu(), g.t("scores"), "scores", rtol=2e-6, atol=1e-7)
    assert_close(det.probs[:m].cpu(), g.t("pred_cls_probs"), "probs", rtol=2e-6, atol=1e-7)
    assert_close(det.boxes[:m].cpu(), ref_boxes, "boxes")
    assert_close(det.cov[:m].cpu(), g.t("pred_boxes_covariance"), "cov")


@pytest.mark.parametrize("path", SMALL, ids=fixture_id)
def test_hip_indices_bit_exact(path):
    """top-k anchor index sequence per level and the NMS keep list equal the reference's."""
    g = Golden(path)
    hp, det = run_hip(g)
    n = int(hp.n_total.item())
    counts = hp.sel_count.cpu().tolist()
    assert n == g.t("aw0_boxes").shape[0]
    ref_top = torch.cat([g.t("topk_%d" % lvl)[:cnt] for lvl, cnt in enumerate(counts)])
    perm = canonical_perm(ref_top, g.t("aw0_prob"), counts)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    assert torch.equal(hp.cand_anchor_idx[:n].cpu().long(), ref_top[perm]), "top-k anchor index sequence"
    assert torch.equal(hp.cand_class[:n].cpu().long(), g.t("aw0_cls")[perm])
    assert_close(hp.cand_score[:n].cpu(), g.t("aw0_prob")[perm], "cand scores", rtol=2e-6, atol=1e-7)
    assert_close(hp.cand_probs[:n].cpu(), g.t("aw0_pvec")[perm], "cand probs", rtol=2e-6, atol=1e-7)
    # replayed eps columns belong to candidate POSITIONS, so members of a re-ordered tie group saw each
    # other's normal draws: compare boxes/covariances on the positions the convention left in place
    same = perm == torch.arange(n)
    assert int(same.sum()) >= n - 16
    assert_close(hp.boxes[:n].cpu()[same], g.t("aw0_boxes")[same], "candidate boxes")
    if g.t("aw0_cov").numel():
        assert_close(hp.cov[:n].cpu()[same], g.t("aw0_cov")[same], "candidate cov")
    nk = int(hp.n_keep.item())
    ref_keep = inv[g.t("nms_keep_0")[:100]]
    assert torch.equal(hp.keep[:nk].cpu().long(), ref_keep)


@pytest.mark.parametrize("seed,runs,mode,quirk", [(7, 1, "bayes_od", True), (8, 6, "bayes_od", True), (9, 6, "bayes_od", False),
                                                  (10, 5, "anchor_statistics", True), (11, 2, "standard_nms", True)])
def test_hip_matches_oracle_fresh_seeds(seed, runs, mode, quirk):
    """Beyond the fixtures: HIP vs the CPU oracle on new seeds (incl. the true-mean, quirk-off merge)."""
    ho = synthetic.planted_head_outputs((192, 256), runs, seed=seed, num_boxes=10)
    hp = make_path(ho, quirk=quirk)
    hd = ho.to("cuda")
    det = hp.run(mode, hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=(180, 250), out_size=(360, 500),
                 eps_fn=synthetic.SeededNormals(seed + 99))
    p = po.PathParams(merge_quirk=quirk)
    rl = [synthetic.to_reference_layout(ho, r) for r in range(runs)]
    ref = po.predict(mode, p, (180, 250), (360, 500), outputs=rl[0] if runs == 1 else None,
                     run_outputs=rl if runs > 1 else None, eps_fn=synthetic.SeededNormals(seed + 99))
    m = det.count()
    assert m == len(ref)
    assert torch.equal(det.classes[:m].cpu().long(), ref.pred_classes)
    assert_close(det.scores[:m].cpu(), ref.scores, "scores", rtol=2e-6, atol=1e-7)
    assert_close(det.boxes[:m].cpu(), ref.pred_boxes, "boxes")
    assert_close(det.cov[:m].cpu(), ref.pred_boxes_covariance, "cov")
    nk = int(hp.n_keep.item())
    if ref.keep is not None and len(ref.keep) == nk:
        assert torch.equal(hp.keep[:nk].cpu().long(), ref.keep)


def test_records_match_json_of_oracle():
    """K7's fixed-stride records == instances_to_json (XYWH box, T cov T^T)."""
    g = Golden([p for p in SMALL if "cfg2_bayes_od" in p][0])
    hp, det = run_hip(g)
    m = det.count()
    import json
    ref = json.loads(str(g.z["json"]))
    rec = det.records[:m].cpu()
    K = 7
    assert len(ref) == m
    for i, r in enumerate(ref):
        assert_close(rec[i, 0:4], r["bbox"], "json bbox")
        assert_close(rec[i, 4], r["score"], "json score", 2e-6, 1e-7)
        assert int(rec[i, 5]) + 1 == r["category_id"]
        assert_close(rec[i, 6:6 + K], r["cls_prob"], "json probs", 2e-6, 1e-7)
        assert_close(rec[i, 6 + K:].reshape(4, 4), r["bbox_covar"], "json cov")


def test_reg_nll_matches_scoring_rule():
    """NLL parity half of the metric (scoring_rules.py:68-74), |dNLL| <= 1e-3."""
    from pod_compare_amd import hip
    rng = synthetic.SeededNormals(5)
    n = 257
    means = 300 + 50 * rng.randn(n, 4)
    l = rng.randn(n, 4, 4)
    covs = torch.matmul(l, l.transpose(1, 2)) + 0.5 * torch.eye(4)
    gt = means + 2.0 * rng.randn(n, 4)
    ref = po.reg_nll(means, covs, gt)
    lib = hip.load()
    d = [t.cuda().contiguous() for t in (means, covs, gt)]
    out = torch.empty(n, device="cuda")
    hip.check(lib.pod_reg_nll(hip.ptr(d[0]), hip.ptr(d[1]), hip.ptr(d[2]), n, hip.ptr(out), hip.current_stream()), "pod_reg_nll")
    assert float((out.cpu() - ref).abs().max()) <= 1e-3
