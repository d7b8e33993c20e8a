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
    assert_close(det.cov[:m].cpu(), g.t("pred_boxes_covariance"), "cov", matrix_scale="/full_" in path)


@pytest.mark.parametrize("path", PRE_NMS, ids=fixture_id)
def test_hip_indices_bit_exact(path):
    """top-k anchor index sequence per level and the NMS keep list equal the reference's (PI:300-311, PI:554-566, IU:78-92) -- on the
    small frames AND at BASELINE's own size (R = 193 374: the five configs planted, configs[2] also on the adversarial distribution
    where every level is truncated at 1000: n = 4 594, the radix-select path of K2)."""
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
        # (full size: per MATRIX -- tests/helpers.py; the small frames keep the element-wise bar they have always met)
        assert_close(hp.cov[:n].cpu()[same], g.t("aw0_cov")[same], "candidate cov", matrix_scale="/full_" in path)
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


# ---------------------------------------------------------------------------------------------------
# native-RNG mode (in-kernel Philox): no bit-level comparison with the CPU is possible; check the
# exactness of K1's pruning bound, determinism, and statistical agreement with the replay path.
# ---------------------------------------------------------------------------------------------------

def _native_candidates(hp, hd, prune):
    from pod_compare_amd import hip
    P, st, lib = hip.ptr, hip.current_stream(), hp.lib
    lv = hp._levels(hd.cls, hd.delta, hd.cls_var, hd.reg_var, None)
    hip.check(lib.pod_reset_counters(P(hp.counters), 8, st), "reset")
    hip.check(lib.pod_mc_merge_score(hp.cfg, lv, P(hp.mean_cls), P(hp.mean_cls_var), P(hp.mean_delta), P(hp.mean_reg_var),
                                     P(hp.cand_keys), P(hp.cand_count), P(hp.maybe_bits) if prune else None, st), "k1")
    n_maybe = 0
    if prune:
        n_maybe = sum(bin(int(x) & (2 ** 64 - 1)).count("1") for x in hp.maybe_bits.cpu().tolist())   # K1b clears the bitmap
        hip.check(lib.pod_score_maybe(hp.cfg, lv, P(hp.mean_cls), P(hp.mean_cls_var), P(hp.maybe_bits),
                                      P(hp.cand_keys), P(hp.cand_count), P(hp.probs_dense), st), "k1b")
    torch.cuda.synchronize()
    counts = hp.cand_count.cpu().tolist()
    hip.check(lib.pod_reset_counters(P(hp.counters), 8, st), "reset")      # leave the workspace clean for the next image
    keys = [torch.sort(hp.cand_keys[b:b + c].cpu())[0] for b, c in zip(hp.anchor_base, counts)]
    assert not prune or int(hp.maybe_bits.abs().sum().item()) == 0      # left zeroed for the next image
    return counts, keys, [n_maybe]


@pytest.mark.parametrize("mode,runs,K", [("planted", 10, 7), ("worst", 3, 7), ("planted", 1, 7), ("planted", 2, 12), ("worst", 2, 3)])
def test_prune_bound_is_exact(mode, runs, K):
    """K1 prune mode + K1b must emit exactly the candidate keys of the dense in-kernel scoring (same Philox
    draws), on planted data (almost everything pruned) and on worst-case data (nothing pruned)."""
    ho = synthetic.planted_head_outputs((384, 512), runs, seed=77, num_boxes=12, mode=mode, num_classes=K).to("cuda")
    hp = make_path(ho)
    c0, k0, _ = _native_candidates(hp, ho, prune=False)
    c1, k1, maybe = _native_candidates(hp, ho, prune=True)
    assert c0 == c1 and sum(c0) > 0
    for a, b in zip(k0, k1):
        assert torch.equal(a, b)
    if mode == "planted":
        assert sum(maybe) < 0.10 * hp.R      # the bound really prunes (more classes = more chances per anchor)
    else:
        assert sum(maybe) > 0.9 * hp.R


def _fused_candidates(hp, hd, store_planes):
    from pod_compare_amd import hip
    P, st, lib = hip.ptr, hip.current_stream(), hp.lib
    lv = hp._levels(hd.cls, hd.delta, hd.cls_var, hd.reg_var, None)
    hip.check(lib.pod_reset_counters(P(hp.counters), 8, st), "reset")
    hip.check(lib.pod_merge_score_fused(hp.cfg, lv, P(hp.mean_cls) if store_planes else None,
                                        P(hp.mean_cls_var) if store_planes and hp.has_cls_var else None, P(hp.cand_keys), P(hp.cand_count),
                                        P(hp.probs_dense) if hp.has_cls_var else None, st), "k1f")
    torch.cuda.synchronize()
    counts = hp.cand_count.cpu().tolist()
    hip.check(lib.pod_reset_counters(P(hp.counters), 8, st), "reset")
    keys = [torch.sort(hp.cand_keys[b:b + c].cpu())[0] for b, c in zip(hp.anchor_base, counts)]
    return counts, keys


@pytest.mark.parametrize("mode,runs,K,padded,cls_var", [
    ("planted", 10, 7, (384, 512), True), ("worst", 3, 7, (384, 512), True), ("planted", 1, 7, (384, 512), True),
    ("planted", 2, 12, (384, 512), True), ("worst", 2, 3, (160, 224), True),
    ("planted", 5, 7, (96, 352), True),                  # ragged maps: H*W % 4 != 0 on the small levels (scalar path)
    ("planted", 1, 7, (384, 512), False), ("planted", 4, 7, (160, 224), False), ("worst", 1, 7, (96, 352), False),    # no variance head
])
@pytest.mark.parametrize("quirk", [True, False])
def test_fused_merge_score_equals_the_two_launch_form(mode, runs, K, padded, cls_var, quirk):
    """pod_merge_score_fused (one streaming launch: the form pod_run_image enqueues) against pod_mc_merge_score + pod_score_maybe:
    the same candidate keys on every level, the same stored class probabilities at the emitted anchors, the same merged planes when
    they are asked for -- bit for bit (same Philox key, same functions)."""
    ho = synthetic.planted_head_outputs(padded, runs, seed=77 + runs, num_boxes=12, mode=mode, num_classes=K, with_cls_var=cls_var).to("cuda")
    params = hotpath.PathParams(num_classes=ho.num_classes, num_anchors=ho.num_anchors, merge_quirk=quirk)
    hp = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=runs, has_cls_var=cls_var, cov_dims=4, device="cuda")
    hp._begin_draw(3)
    planes = [t for t in (hp.mean_cls, hp.mean_cls_var) if t is not None]
    if cls_var:
        hp.probs_dense.fill_(-1.0)
    c0, k0, _ = _native_candidates(hp, ho, prune=cls_var)
    planes0 = [t.clone() for t in planes]
    probs0 = hp.probs_dense.clone() if cls_var else None
    if cls_var:
        hp.probs_dense.fill_(-1.0)
    for t in planes:
        t.fill_(float("nan"))
    c1, k1 = _fused_candidates(hp, ho, store_planes=runs > 1)
    assert c0 == c1 and sum(c0) > 0
    for a, b in zip(k0, k1):
        assert torch.equal(a, b)
    if cls_var:
        assert torch.equal(probs0, hp.probs_dense)
    for a, b in zip(planes0, planes):
        assert torch.equal(a, b)
    # without the planes: the same keys again
    c2, k2 = _fused_candidates(hp, ho, store_planes=False)
    assert c2 == c0 and all(torch.equal(a, b) for a, b in zip(k0, k2))


def test_native_mode_is_deterministic_and_close_to_replay():
    g = Golden([p for p in SMALL if "cfg3_bayes_od_mc10_s31" in p][0])
    ho = g.head_outputs()
    hp = make_path(ho)
    hd = ho.to("cuda")
    kw = dict(image_size=tuple(g.meta["image"]), out_size=tuple(g.meta["out"]), draw_id=0)      # same Philox key: same draws
    d1 = hp.run("bayes_od", hd.cls, hd.delta, hd.cls_var, hd.reg_var, **kw)
    b1, c1, m1 = d1.boxes.clone(), d1.cov.clone(), d1.count()
    d2 = hp.run("bayes_od", hd.cls, hd.delta, hd.cls_var, hd.reg_var, **kw)
    assert d2.count() == m1 and torch.equal(d2.boxes[:m1], b1[:m1]) and torch.equal(d2.cov[:m1], c1[:m1])
    # against the reference (torch normals): the same set of detections (order by score may swap between
    # near-equal scores because the draws differ), means within sampling error, variances within ~30 %
    ref_b, ref_c, ref_cls = g.t("pred_boxes"), g.t("pred_boxes_covariance"), g.t("pred_classes")
    assert m1 == ref_b.shape[0]
    nb, nc, ncls = b1[:m1].cpu(), c1[:m1].cpu(), d1.classes[:m1].cpu().long()
    dist = (nb[:, None, :] - ref_b[None, :, :]).abs().sum(-1)
    match = dist.argmin(1)
    assert sorted(match.tolist()) == list(range(m1))          # one-to-one
    assert torch.equal(ncls, ref_cls[match])
    rb, rc = ref_b[match], ref_c[match]
    sd = rc.diagonal(dim1=1, dim2=2).sqrt()
    ratio = (nb - rb).abs() / (sd + 1e-2)
    # (a cluster member entering/leaving at IoU ~ 0.9 moves a fused box by a fraction of a sigma: also true of the
    # reference from one seed to the next)
    assert float(ratio.max()) <= 1.0, (ratio.max(), ratio.argmax(), nb.reshape(-1)[ratio.argmax()], rb.reshape(-1)[ratio.argmax()], sd.reshape(-1)[ratio.argmax()])
    rel = (nc.diagonal(dim1=1, dim2=2) - rc.diagonal(dim1=1, dim2=2)).abs() / rc.diagonal(dim1=1, dim2=2)
    assert float(rel.max()) < 2.0      # fused covariances scale with the member count; moments are checked per candidate below


def test_native_candidate_moments_match_replay_statistically():
    """Before any clustering: per-candidate sample means / covariances from in-kernel Philox (1000 bounded 16-bit
    Box-Muller draws) against the eps-replay values; mean error <= 5 sigma/sqrt(1000), variance within 25 %."""
    g = Golden([p for p in SMALL if "cfg2_bayes_od_regclsvar_s22" in p][0])
    ho = g.head_outputs()
    hd = ho.to("cuda")
    hp = make_path(ho)
    hp.run("standard_nms", hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=tuple(g.meta["image"]), out_size=tuple(g.meta["out"]),
           eps_fn=g.eps_source())
    n0 = int(hp.n_total.item())
    idx0, b0, c0 = hp.cand_anchor_idx[:n0].cpu(), hp.boxes[:n0].cpu(), hp.cov[:n0].cpu()
    lvl0 = hp.cand_level[:n0].cpu()
    hp.run("standard_nms", hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=tuple(g.meta["image"]), out_size=tuple(g.meta["out"]))
    n1 = int(hp.n_total.item())
    idx1, b1, c1, lvl1 = hp.cand_anchor_idx[:n1].cpu(), hp.boxes[:n1].cpu(), hp.cov[:n1].cpu(), hp.cand_level[:n1].cpu()
    key0 = {(int(l), int(i)): k for k, (l, i) in enumerate(zip(lvl0, idx0))}
    pairs = [(key0[(int(l), int(i))], k) for k, (l, i) in enumerate(zip(lvl1, idx1)) if (int(l), int(i)) in key0]
    assert len(pairs) > 0.9 * n0
    a = torch.tensor([p[0] for p in pairs])
    b = torch.tensor([p[1] for p in pairs])
    sd = c0[a].diagonal(dim1=1, dim2=2).sqrt()
    z = (b1[b] - b0[a]).abs() / (sd / 1000 ** 0.5 * 2 ** 0.5)      # both sides are 1000-sample means
    assert float(z.max()) < 6.0 and float((z > 3).float().mean()) < 0.02
    rel = (c1[b].diagonal(dim1=1, dim2=2) / c0[a].diagonal(dim1=1, dim2=2) - 1).abs()
    assert float(rel.max()) < 0.25


# ---------------------------------------------------------------------------------------------------
# K1's dense outputs: the merged planes are the reference's PI:211-270 tensors, bit for bit
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("quirk", [True, False])
@pytest.mark.parametrize("runs,padded,prune", [(10, (384, 512), True), (5, (384, 512), False), (3, (96, 352), True), (2, (160, 224), False)])
def test_dense_merge_planes_equal_reference_merge(runs, padded, prune, quirk):
    """pod_mc_merge_score with every mean_* output requested (HotPath(dense_box_merge=True)) against the oracle's
    left-to-right merge (PI:216-222: x0 twice, last run never, one divide), in both kernels (streaming prune kernel /
    LDS class-per-wave kernel) and on ragged maps (scalar tails).  Layout: level-major, NCHW inside a level."""
    from pod_compare_amd import hip
    ho = synthetic.planted_head_outputs(padded, runs, seed=31 + runs, num_boxes=8)
    params = hotpath.PathParams(num_classes=ho.num_classes, num_anchors=ho.num_anchors, merge_quirk=quirk)
    hp = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=runs, has_cls_var=True, cov_dims=4, device="cuda", dense_box_merge=True)
    hd = ho.to("cuda")
    P, st, lib = hip.ptr, hip.current_stream(), hp.lib
    lv = hp._levels(hd.cls, hd.delta, hd.cls_var, hd.reg_var, None)
    for t in (hp.mean_cls, hp.mean_cls_var, hp.mean_delta, hp.mean_reg_var):
        t.fill_(float("nan"))
    hip.check(lib.pod_reset_counters(P(hp.counters), 8, st), "reset")
    hip.check(lib.pod_mc_merge_score(hp.cfg, lv, P(hp.mean_cls), P(hp.mean_cls_var), P(hp.mean_delta), P(hp.mean_reg_var),
                                     P(hp.cand_keys), P(hp.cand_count), P(hp.maybe_bits) if prune else None, st), "k1")
    torch.cuda.synchronize()
    if prune:
        hp.maybe_bits.zero_()
    for name, out, src in (("cls", hp.mean_cls, ho.cls), ("cls_var", hp.mean_cls_var, ho.cls_var),
                           ("delta", hp.mean_delta, ho.delta), ("reg_var", hp.mean_reg_var, ho.reg_var)):
        off = 0
        for l, x in enumerate(src):
            ref = po.merge_runs([x[r] for r in range(runs)], quirk=quirk)           # (A*C, H, W)
            got = out[off:off + ref.numel()].cpu().view_as(ref)
            assert torch.equal(got, ref), (name, l, float((got - ref).abs().max()))
            off += ref.numel()
        assert off == out.numel()


def test_product_path_skips_the_dense_box_merge_and_gets_the_same_detections():
    """The default HotPath leaves box_delta / box_reg_var to K2b (merged at the candidates only); requesting the dense
    planes as well must not change a single output bit."""
    ho = synthetic.planted_head_outputs((384, 512), 6, seed=5, num_boxes=10)
    hd = ho.to("cuda")
    outs = []
    for dense in (False, True):
        params = hotpath.PathParams(num_classes=ho.num_classes, num_anchors=ho.num_anchors)
        hp = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=6, has_cls_var=True, cov_dims=4, device="cuda", dense_box_merge=dense)
        assert (hp.mean_delta is not None) == dense
        det = hp.run("bayes_od", hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=(380, 500), out_size=(760, 1000))
        outs.append((det.count(), det.boxes.clone(), det.cov.clone(), det.scores.clone(), det.classes.clone()))
    assert outs[0][0] == outs[1][0] > 0
    m = outs[0][0]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(a[:m], b[:m])
