"""Exact check of the PRODUCT kernels (native-RNG mode: k1_prune_stream + k1b_score_maybe + k2 + fused k23_gather_decode +
K4..K7 enqueued by pod_run_image) against the CPU oracle.

The product path draws its normals in-kernel (Philox4x32-10 + 16-bit Box-Muller) and never stores them, so the oracle
cannot be given "the same seed".  Instead the draws themselves are dumped (pod_dump_cls_normals / pod_dump_box_normals
evaluate the same counter -> normal maps, in the reference's (S, R_l, K) / (1000, n, 4) layouts) and fed to
`po.predict(..., eps_fn=ReplayEps(...))`: the oracle then runs the reference's exact arithmetic on exactly the draws the
product kernels used.  What may still differ is the native mode's arithmetic: v_exp / v_rcp approximations in the class
probabilities (pod_device.h: sigmoid_fast), a few ulp on a score; the box decode / moment code is the replay mode's.

Bar: candidate sets identical except for anchors whose score lies within 2e-6 (relative) of the threshold; candidate
order identical except inside groups of scores closer than 2e-6; NMS keep list and final classes identical; scores within
2e-6; boxes / covariances within 1e-4 * max(1, |ref|)."""
import os

import pytest
import torch

from oracle import pod_oracle as po
from pod_compare_amd import synthetic
from tests.helpers import GOLDEN, Golden, assert_close
from tests.test_hip_parity import make_path

pytestmark = pytest.mark.gpu

CASES = ["cfg2_bayes_od_regclsvar_s21", "cfg3_bayes_od_mc10_s31", "cfg3_bayes_od_mc10_s32", "standard_nms_regclsvar_s91",
         "anchor_stats_regclsvar_s61", "bayes_od_ci_clsbayes_s71", "full_cov_standard_nms_s121", "mc3_standard_nms_regclsvar_s131",
         "cfg5_ensembles_pre_nms_s51", "worst_bayes_od_mc4_s111", "full_cfg3_bayes_od_mc10_s1001"]
SCORE_RTOL = 2e-6


def assert_cov_close(a, b, what):
    """|a - b| <= 1e-4 * max(1, |b|) element-wise -- except that an entry which is tiny only by cancellation (an off-diagonal
    of a covariance whose diagonal is 1e4: the "worst" inputs' 0.3-sigma deltas on 800-pixel anchors) is held to 1e-6 of
    its matrix's largest entry instead: fp32 sums of 1000 products of that size differ by more than 1e-4 absolute between
    any two summation orders."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape
    if a.numel() == 0:
        return
    scale = b.abs().reshape(b.shape[0], -1).max(dim=1)[0].reshape(-1, 1, 1)
    bound = torch.maximum(1e-4 * b.abs().clamp(min=1.0), 1e-6 * scale)
    err = (a - b).abs()
    assert bool((err <= bound).all()), "{}: {} elements off, worst excess {:.3e}".format(what, int((err > bound).sum()), float((err - bound).max()))


class _ClsThenZeros:
    """eps source for the oracle's first pass: the dumped classification normals, zeros for the box draws (the candidate
    selection does not depend on them)."""

    def __init__(self, eps_cls):
        self.eps_cls, self.pos = list(eps_cls), 0

    def __call__(self, shape):
        if self.pos < len(self.eps_cls):
            t = self.eps_cls[self.pos]
            self.pos += 1
            assert tuple(t.shape) == tuple(shape)
            return t
        return torch.zeros(tuple(shape))


def _oracle_inputs(ho, runs):
    if runs == 1:
        return dict(outputs=synthetic.to_reference_layout(ho, 0))
    return dict(run_outputs=[synthetic.to_reference_layout(ho, r) for r in range(runs)])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("draw_id", [0, 12345])
def test_product_kernels_equal_oracle_on_their_own_draws(name, draw_id):
    g = Golden(os.path.join(GOLDEN, name + ".npz"))
    ho = g.head_outputs()
    if draw_id != 0 and sum(int(t.shape[1] * t.shape[2] * t.shape[3]) for t in ho.delta) // 4 > 100000:
        pytest.skip("one draw id is enough at full size (R = 193 374 anchors)")
    hd = ho.to("cuda")
    s = g.spec
    hp = make_path(ho, g.meta["topk"])
    image, out = tuple(g.meta["image"]), tuple(g.meta["out"])
    modes = dict(box_merge_mode=s.get("box_merge", "bayesian_inference"), cls_merge_mode=s.get("cls_merge", "max_score"))

    # ---- the product path: one C call, in-kernel draws ---------------------------------------------------------
    det = hp.run(s["mode"], hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=image, out_size=out, draw_id=draw_id, **modes)
    m = det.count()
    n = int(hp.n_total.item())
    counts = hp.sel_count.cpu().tolist()
    nat = dict(level=hp.cand_level[:n].cpu().long(), idx=hp.cand_anchor_idx[:n].cpu().long(), score=hp.cand_score[:n].cpu(),
               cls=hp.cand_class[:n].cpu().long(), probs=hp.cand_probs[:n].cpu(), boxes=hp.boxes[:n].cpu().clone(), cov=hp.cov[:n].cpu().clone(),
               keep=hp.keep[:int(hp.n_keep.item())].cpu().long())
    final = dict(boxes=det.boxes[:m].cpu().clone(), cov=det.cov[:m].cpu().clone(), scores=det.scores[:m].cpu().clone(),
                 classes=det.classes[:m].cpu().long(), probs=det.probs[:m].cpu().clone())

    # ---- the same draws, written out ------------------------------------------------------------------------
    eps_cls = [t.cpu() for t in hp.dump_cls_normals(draw_id)]
    assert max(float(t.abs().max()) for t in eps_cls) < 4.9          # the sampler's hard bound K1 prunes with
    params = po.PathParams(num_classes=ho.num_classes, topk_candidates=g.meta["topk"])
    kw = _oracle_inputs(ho, s["runs"])
    aw = po.anchorwise_inference(kw.get("outputs"), params, run_outputs=kw.get("run_outputs"), eps_fn=_ClsThenZeros(eps_cls))
    base = torch.tensor(hp.anchor_base)
    ref_level = torch.repeat_interleave(torch.arange(len(aw.level_counts)), torch.tensor(aw.level_counts))
    gids = base[ref_level] + aw.anchor_idx
    eps_prop = hp.dump_box_normals(draw_id, gids).cpu()
    ref = po.predict(s["mode"], params, image, out, eps_fn=po.ReplayEps(eps_cls + [eps_prop]), **kw, **modes)
    aw = po.anchorwise_inference(kw.get("outputs"), params, run_outputs=kw.get("run_outputs"), eps_fn=po.ReplayEps(eps_cls + [eps_prop]))

    # ---- candidate sets -------------------------------------------------------------------------------------
    thr = params.score_thresh
    near_thr = lambda sc: abs(float(sc) - thr) <= SCORE_RTOL * thr
    nat_key = {(int(l), int(i)): k for k, (l, i) in enumerate(zip(nat["level"], nat["idx"]))}
    ref_key = {(int(l), int(i)): k for k, (l, i) in enumerate(zip(ref_level, aw.anchor_idx))}
    for key in set(nat_key) ^ set(ref_key):
        sc = nat["score"][nat_key[key]] if key in nat_key else aw.scores[ref_key[key]]
        full_level = (key in ref_key and aw.level_counts[key[0]] == g.meta["topk"]) or (key in nat_key and counts[key[0]] == g.meta["topk"])
        assert near_thr(sc) or full_level, ("candidate only on one side, not at the threshold", key, float(sc))
    common = [key for key in nat_key if key in ref_key]
    assert len(common) >= max(len(nat_key), len(ref_key)) - 4 and len(common) > 0
    a = torch.tensor([nat_key[k] for k in common])
    b = torch.tensor([ref_key[k] for k in common])
    assert_close(nat["score"][a], aw.scores[b], "candidate scores", rtol=SCORE_RTOL, atol=1e-7)
    assert_close(nat["probs"][a], aw.probs[b], "candidate prob vectors", rtol=SCORE_RTOL, atol=1e-7)
    assert torch.equal(nat["cls"][a], aw.classes[b])
    # order: positions may only differ inside groups of near-equal scores
    moved = (a != b).nonzero().squeeze(1)
    for k in moved.tolist():
        lo, hi = sorted((int(a[k]), int(b[k])))
        span = nat["score"][lo:hi + 1]
        assert float(span.max() - span.min()) <= 2 * SCORE_RTOL * float(span.max()), "candidate order differs outside a near-tie"
    assert_close(nat["boxes"][a], aw.boxes[b], "candidate boxes")
    assert_cov_close(nat["cov"][a], aw.cov[b], "candidate covariances")

    # ---- detections -----------------------------------------------------------------------------------------
    assert m == len(ref) and m > 0
    if len(moved) == 0 and len(common) == len(nat_key) == len(ref_key):
        assert torch.equal(final["classes"], ref.pred_classes)
        assert_close(final["scores"], ref.scores, "scores", rtol=SCORE_RTOL, atol=1e-7)
        assert_close(final["probs"], ref.pred_cls_probs, "probs", rtol=SCORE_RTOL, atol=1e-7)
        assert_close(final["boxes"], ref.pred_boxes, "boxes")
        assert_cov_close(final["cov"], ref.pred_boxes_covariance, "cov")
    else:   # a near-tie moved: match detections by box, then the same bars
        d = (final["boxes"][:, None, :] - ref.pred_boxes[None, :, :]).abs().sum(-1)
        match = d.argmin(1)
        assert sorted(match.tolist()) == list(range(m))
        assert torch.equal(final["classes"], ref.pred_classes[match])
        assert_close(final["boxes"], ref.pred_boxes[match], "boxes")
        assert_cov_close(final["cov"], ref.pred_boxes_covariance[match], "cov")

    # ---- and the HIP eps-replay kernels on the same draws: the decode / moment code is shared, so with an identical
    # candidate list boxes and covariances must be BIT-identical between the two K1/K2b/K3 implementations --------
    if len(moved) == 0 and len(common) == len(nat_key) == len(ref_key):
        feed = po.ReplayEps([t for t in eps_cls] + [eps_prop])
        hp2 = make_path(ho, g.meta["topk"])
        hp2.run(s["mode"], hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=image, out_size=out, eps_fn=feed, **modes)
        n2 = int(hp2.n_total.item())
        if n2 == n and torch.equal(hp2.cand_anchor_idx[:n].cpu().long(), nat["idx"]):
            assert torch.equal(hp2.boxes[:n].cpu(), nat["boxes"]) and torch.equal(hp2.cov[:n].cpu(), nat["cov"])
            assert torch.equal(hp2.keep[:int(hp2.n_keep.item())].cpu().long(), nat["keep"])


def test_draw_ids_give_independent_draws_and_default_is_fresh():
    """every image (and every member of a post-NMS ensemble) must see its own normals, as the reference's
    per-call rsample does (PI:291-294, 351-356).  Same draw id = same draws; different ids = uncorrelated draws; the
    default (no id) advances a per-workspace counter."""
    ho = synthetic.planted_head_outputs((192, 256), 2, seed=4, num_boxes=6)
    hp = make_path(ho)
    a0 = hp.dump_cls_normals(7)[0].cpu()
    a1 = hp.dump_cls_normals(7)[0].cpu()
    b = hp.dump_cls_normals(8)[0].cpu()
    assert torch.equal(a0, a1) and not torch.equal(a0, b)
    corr = float(torch.corrcoef(torch.stack((a0.reshape(-1), b.reshape(-1))))[0, 1])
    assert abs(corr) < 0.02 and abs(float(a0.mean())) < 0.01 and abs(float(a0.std()) - 1.0) < 0.01
    hd = ho.to("cuda")
    kw = dict(image_size=(180, 250), out_size=(180, 250))
    d1 = hp.run("standard_nms", hd.cls, hd.delta, hd.cls_var, hd.reg_var, **kw)
    c1 = d1.cov[:d1.count()].clone()
    d2 = hp.run("standard_nms", hd.cls, hd.delta, hd.cls_var, hd.reg_var, **kw)
    assert d2.count() == d1.count() and not torch.equal(d2.cov[:d2.count()], c1)          # fresh draws by default
