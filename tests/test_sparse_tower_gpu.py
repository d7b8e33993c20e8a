"""Sparse bbox tower (csrc/k15_sparse_blocks.hip, pod_compare_amd/sparse.py): PI:310-331 reads box_delta / box_reg_var only at the candidates
of PI:300-308, so bbox_subnet / bbox_pred / bbox_cov (PR:518-537) are evaluated over the blocks that can reach one -- and the detections
must be those of the dense evaluation."""
import numpy as np
import pytest
import torch

from pod_compare_amd import hip, hotpath, modeling, sparse, synthetic
from pod_compare_amd.wino import block_table

pytestmark = pytest.mark.gpu


def build(plain=False, **kw):
    torch.manual_seed(0)
    heads = {} if plain else dict(cls_var_loss="loss_attenuation", cls_var_num_samples=10, bbox_cov_loss="negative_log_likelihood")
    m = modeling.ProbabilisticRetinaNet(**heads, **kw).cuda().eval()
    modeling.fold_frozen_bn(m)
    for q in m.parameters():
        q.requires_grad_(False)
    return generic_deltas(m)


def generic_deltas(m):
    """A random-init bbox_pred (std 0.01) leaves every box ON its anchor, where IoUs between neighbouring anchors sit exactly on the NMS /
    affinity thresholds and a last-bit difference flips a keep decision: give the deltas a generic size (std 0.25)."""
    with torch.no_grad():
        probe = torch.randint(0, 256, (3, 128, 192), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(77))
        std = float(torch.cat([t.reshape(-1) for t in m(probe).delta]).std())
        m.head.bbox_pred.weight.mul_(0.25 / max(std, 1e-12))
    return m


def planted(padded, runs, seed, mode="planted"):
    return synthetic.planted_head_outputs(padded, runs, seed=seed, num_boxes=6, mode=mode).to("cuda")


def reach_reference(hp, n_total, cat_keys, cat_level):
    """needed(j) = candidates (+) box(2 j rows, 4 j columns), brute force."""
    A = hp.p.num_anchors
    out = []
    for l, (h, w) in enumerate(hp.shapes):
        r = np.full((h, w), 255, dtype=np.int64)
        idx = [(0xFFFFFFFF - (int(k) & 0xFFFFFFFF)) // A for k, lv in zip(cat_keys[:n_total], cat_level[:n_total]) if lv == l]
        ys, xs = np.arange(h)[:, None], np.arange(w)[None, :]
        for c in set(idx):
            cy, cx = divmod(c, w)
            j = np.maximum((np.abs(ys - cy) + 1) // 2, (np.abs(xs - cx) + 3) // 4)
            r = np.minimum(r, np.where(j <= 5, j, 255))
        out.append(r.reshape(-1))
    return np.concatenate(out)


@pytest.mark.parametrize("mode,runs", [("planted", 3), ("worst", 1)])
def test_reach_map_and_live_lists_equal_a_brute_force_evaluation(mode, runs):
    padded = (256, 384)
    ho = planted(padded, runs, seed=5, mode=mode)
    hp = hotpath.HotPath(ho.shapes, ho.anchors, hotpath.PathParams(), n_runs=runs, has_cls_var=True, cov_dims=4, device="cuda:0")
    hp.select(ho.cls, ho.cls_var, draw_id=3)
    live = sparse.LiveBlocks(hp)
    n = int(hp.n_total.item())
    assert n > 0
    want = reach_reference(hp, n, hp.cat_keys.cpu().tolist(), hp.cat_level.cpu().tolist())
    got = live.reach.cpu().numpy().astype(np.int64)
    assert np.array_equal(got, want)
    levels = [tuple(s) for s in hp.shapes]
    for copies, reach, in_reach in ((1, 4, 255), (5, 2, 3), (5, 0, 1)):
        table = block_table(levels, copies, "cuda")
        lst = live(table, reach, in_reach).cpu().numpy()
        ents = lst[sparse.LIVE_HEAD:sparse.LIVE_HEAD + sparse.LIVE_STRIDE * lst[0]].reshape(-1, sparse.LIVE_STRIDE)
        recs, rl = table.cpu().numpy(), table.pod_rec_level.cpu().numpy()
        base = np.cumsum([0] + [h * w for h, w in levels])
        expect = {}
        for r, (d, l) in enumerate(zip(recs, rl)):
            gcols, H, W, n_img = (d[2] >> 24) & 0xFF, (d[2] >> 12) & 0xFFF, d[2] & 0xFFF, (d[3] >> 24) & 0xFF
            by, bx = (d[3] >> 12) & 0xFFF, d[3] & 0xFFF

            def cells_of(vy, vx):
                m_, gy, n_, gx = vy // (H + 1), vy % (H + 1), vx // (W + 1), vx % (W + 1)
                ok = (vy >= 0) & (vx >= 0) & (gy < H) & (gx < W) & (n_ < gcols) & (m_ * gcols + n_ < n_img)
                return ok, base[l] + np.where(ok, gy * W + gx, 0)
            ok, cells = cells_of(*np.meshgrid(16 * by + np.arange(16), 16 * bx + np.arange(16), indexing="ij"))
            if bool((ok & (want[cells] <= reach)).any()):
                # need bits: the 18 x 18 input patch of the block (canvas rows / columns -1 .. 16), row-major
                okp, cp = cells_of(*np.meshgrid(16 * by - 1 + np.arange(18), 16 * bx - 1 + np.arange(18), indexing="ij"))
                expect[r] = (okp & (want[cp] <= in_reach)).reshape(-1)
        assert lst[0] == len(expect) and set(ents[:, 0].tolist()) == set(expect)
        assert set(live.records(table, reach, in_reach).cpu().tolist()) == set(expect)
        for e in ents:
            words = e[1:].astype(np.uint32)
            bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[:324].astype(bool)
            assert np.array_equal(bits, expect[int(e[0])]), int(e[0])
        if mode == "planted" and reach == 0:
            assert lst[0] < 0.6 * len(recs)                    # a handful of objects: most blocks are dead


@pytest.mark.parametrize("dropout,runs", [(0.2, 4), (0.0, 1)])
def test_sparse_tower_gives_the_dense_towers_detections(dropout, runs):
    """The whole model twice on one frame, dense and sparse, with the SAME planted class tensors selecting the candidates (a random-init
    head has none): identical candidates and masks, box deltas / variances at the candidates equal to the last bits (the abs-max record
    of a sparse launch covers the live blocks only: another binade moves the f16 splits' roundings), identical detections."""
    m = build(dropout_rate=dropout)
    frame = torch.randint(0, 256, (3, 256, 384), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    pl = planted((256, 384), runs, seed=9)
    mc = dropout > 0
    kw = dict(num_mc_dropout_runs=runs if mc else -1, mc_dropout=mc, skip_unused_last_run=mc)
    m.head._drop_calls = 0
    dense = m(frame, **kw)
    hp = hotpath.HotPath(pl.shapes, pl.anchors, hotpath.PathParams(), n_runs=runs if mc else 1, has_cls_var=True, cov_dims=4, device="cuda:0")
    cls = pl.cls if mc else [t[:1] for t in pl.cls]
    cls_var = pl.cls_var if mc else [t[:1] for t in pl.cls_var]
    want = hp.run_image("bayes_od", cls, dense.delta, cls_var, dense.reg_var, (256, 384), (256, 384), draw_id=11)
    n = int(hp.n_total.item())
    ref = [t[:n].clone() for t in (hp.cand_delta, hp.cand_reg_var, hp.cand_anchor_idx, hp.boxes, hp.cov)]
    shares = {}

    def hook(partial):
        hp.select(cls, cls_var, draw_id=11)
        lb = sparse.LiveBlocks(hp)
        shares["lb"] = lb
        return lb

    m.head._drop_calls = 0
    sp = m(frame, sparse_bbox=hook, **kw)
    for a, b in zip(sp.cls + sp.cls_var, dense.cls + dense.cls_var):
        # the cls side is the dense one (MIOpen's p6 / p7 kernels accumulate with atomics: two forwards agree to rounding, not bit for bit)
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))
    got = hp.finish("bayes_od", cls, sp.delta, cls_var, sp.reg_var, (256, 384), (256, 384))
    assert int(hp.n_total.item()) == n and n > 20
    assert torch.equal(hp.cand_anchor_idx[:n], ref[2])
    for name, a, b in (("delta", hp.cand_delta[:n], ref[0]), ("reg_var", hp.cand_reg_var[:n], ref[1])):
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), name
    assert float((hp.boxes[:n] - ref[3]).abs().max()) <= 1e-4 * max(1.0, float(ref[3].abs().max()))
    k = got.count()
    assert k == want.count() and k > 0
    assert torch.equal(got.classes[:k], want.classes[:k])
    assert float((got.boxes[:k] - want.boxes[:k]).abs().max()) <= 1e-4 * max(1.0, float(want.boxes[:k].abs().max()))
    assert float((got.cov[:k] - want.cov[:k]).abs().max()) <= 1e-4 * max(1.0, float(want.cov[:k].abs().max()))
    # and the tower really was sparse
    lv = [tuple(s) for s in pl.shapes]
    assert shares["lb"].fraction(block_table(lv, 1, "cuda"), 0) < 0.7


def test_predictor_switch_gives_the_same_instances():
    """build_predictor(...)(input_im) with and without `sparse_bbox_tower`, a model whose class bias lets every level fill its top-k
    (all blocks live: the switch must not change a thing beyond the last bits)."""
    import os
    from pod_compare_amd import config
    from pod_compare_amd.probabilistic_inference import build_predictor
    root = os.path.join(os.path.dirname(os.path.abspath(config.__file__)), "configs")
    cfg = config.setup_config(os.path.join(root, "BDD-Detection", "retinanet", "retinanet_R_50_FPN_1x_reg_cls_var.yaml"),
                              os.path.join(root, "Inference", "bayes_od.yaml"))
    cfg.MODEL.DEVICE = "cuda"
    m = build()
    with torch.no_grad():
        m.head.cls_score.weight.mul_(40.0)
        m.head.cls_score.bias.fill_(-2.5)
    frame = torch.randint(0, 256, (3, 200, 300), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    inp = [{"image": frame, "height": 200, "width": 300, "image_id": 7}]
    outs = []
    for on in (False, True):
        p = build_predictor(cfg, model=m)
        p.sparse_bbox_tower = on
        outs.append(p(inp))
    a, b = outs
    assert len(a) == len(b) and len(a) > 0
    assert torch.equal(a.pred_classes, b.pred_classes)
    assert float((a.pred_boxes.tensor - b.pred_boxes.tensor).abs().max()) <= 1e-4 * max(1.0, float(a.pred_boxes.tensor.abs().max()))
    assert float((a.pred_boxes_covariance - b.pred_boxes_covariance).abs().max()) <= 1e-4 * max(1.0, float(a.pred_boxes_covariance.abs().max()))


def test_ensemble_members_share_one_set_of_live_blocks():
    """BASELINE configs[4] in process (PI:495-505): every member's cls side first, ONE candidate selection on the merged class outputs, every
    member's bbox side over the same live blocks -- the same Instances as the dense members."""
    import os
    from pod_compare_amd import config
    from pod_compare_amd.probabilistic_inference import build_predictor
    root = os.path.join(os.path.dirname(os.path.abspath(config.__file__)), "configs")
    cfg = config.setup_config(os.path.join(root, "BDD-Detection", "retinanet", "retinanet_R_50_FPN_1x_reg_cls_var.yaml"),
                              os.path.join(root, "Inference", "ensembles_pre_nms.yaml"))
    cfg.MODEL.DEVICE = "cuda"
    members = []
    for seed in (0, 1000, 2000):
        torch.manual_seed(seed)
        m = modeling.ProbabilisticRetinaNet(cls_var_loss="loss_attenuation", cls_var_num_samples=10, bbox_cov_loss="negative_log_likelihood").cuda().eval()
        modeling.fold_frozen_bn(m)
        with torch.no_grad():
            m.head.cls_score.weight.mul_(40.0)
            m.head.cls_score.bias.fill_(-2.5)
        members.append(generic_deltas(m))
    frame = torch.randint(0, 256, (3, 200, 300), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    inp = [{"image": frame, "height": 200, "width": 300, "image_id": 3}]
    outs = []
    for on in (False, True):
        p = build_predictor(cfg, model=members[0], model_list=members)
        p.sparse_bbox_tower = on
        outs.append(p(inp))
    a, b = outs
    assert len(a) == len(b) and len(a) > 0
    assert torch.equal(a.pred_classes, b.pred_classes)
    assert float((a.pred_boxes.tensor - b.pred_boxes.tensor).abs().max()) <= 1e-4 * max(1.0, float(a.pred_boxes.tensor.abs().max()))
    assert float((a.pred_boxes_covariance - b.pred_boxes_covariance).abs().max()) <= 1e-4 * max(1.0, float(a.pred_boxes_covariance.abs().max()))


@pytest.mark.parametrize("dropout,runs", [(0.0, 1), (0.2, 4)])
def test_sparse_results_do_not_depend_on_the_images_before(dropout, runs):
    """Round 6: the sparse tower is a function of the image alone.  A frame whose activations are 50 x larger, with its
    candidates elsewhere, runs first -- it leaves loud values in the blocks that are dead for the NEXT frame, inside live blocks' patches,
    in the tower's re-used buffers.  The quiet frame after it must give, BIT FOR BIT, what it gives in a session that never saw the loud
    one: box deltas / variances at the candidates, boxes, covariances, keep list, detections.  (A sparse launch reads the cells the layer
    below did not compute for this image as 0.0 and every launch max'es into a fresh abs-max record; round 5 read the stale values and kept
    one never-zeroed record per buffer.)  And it still equals the dense tower's detections to the parity bar."""
    m = build(dropout_rate=dropout)
    mc = dropout > 0
    kw = dict(num_mc_dropout_runs=runs if mc else -1, mc_dropout=mc, skip_unused_last_run=mc)
    n_runs = runs if mc else 1
    pl = planted((256, 384), n_runs, seed=9)
    other = planted((256, 384), n_runs, seed=21)         # the loud frame's candidates lie elsewhere: its live blocks are the quiet frame's dead ones
    hp = hotpath.HotPath(pl.shapes, pl.anchors, hotpath.PathParams(), n_runs=n_runs, has_cls_var=True, cov_dims=4, device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(3)
    quiet = torch.randint(0, 256, (3, 256, 384), dtype=torch.uint8, device="cuda", generator=g).float()
    loud = quiet * 50.0

    def run(frame, sparse_on, who):
        m.head._drop_calls = 0                            # the same dropout masks for every evaluation of the frame
        if m.head._epoch is not None:
            m.head._epoch.zero_()

        def hook(partial):
            hp.select(who.cls, who.cls_var, draw_id=5)
            return sparse.LiveBlocks(hp)
        if sparse_on:
            ho = m(frame, sparse_bbox=hook, **kw)
            det = hp.finish("bayes_od", who.cls, ho.delta, who.cls_var, ho.reg_var, (256, 384), (256, 384))
        else:
            ho = m(frame, **kw)
            det = hp.run_image("bayes_od", who.cls, ho.delta, who.cls_var, ho.reg_var, (256, 384), (256, 384), draw_id=5)
        n, k = int(hp.n_total.item()), det.count()
        nk = int(hp.n_keep.item())
        return {"n": n, "k": k, "cand_delta": hp.cand_delta[:n].clone(), "cand_reg_var": hp.cand_reg_var[:n].clone(), "cand_boxes": hp.boxes[:n].clone(),
                "cand_cov": hp.cov[:n].clone(), "keep": hp.keep[:nk].clone(), "boxes": det.boxes[:k].clone(), "cov": det.cov[:k].clone(),
                "classes": det.classes[:k].clone(), "scores": det.scores[:k].clone(), "finite": all(bool(torch.isfinite(t).all()) for t in ho.delta + ho.reg_var)}

    alone = run(quiet, True, pl)                          # a fresh model: the quiet frame is the first image the tower ever sees
    run(loud, True, other)
    after = run(quiet, True, pl)
    assert after["finite"] and alone["n"] == after["n"] > 20 and alone["k"] == after["k"] > 0
    for name in ("cand_delta", "cand_reg_var", "cand_boxes", "cand_cov", "keep", "boxes", "cov", "classes", "scores"):
        assert torch.equal(alone[name], after[name]), name
    # ... twice more, the other way round (quiet, quiet): still the same bits
    again = run(quiet, True, pl)
    for name in ("cand_delta", "cand_reg_var", "boxes", "cov", "keep"):
        assert torch.equal(alone[name], again[name]), name
    want = run(quiet, False, pl)
    assert want["k"] == after["k"] and torch.equal(want["classes"], after["classes"]) and torch.equal(want["keep"], after["keep"])
    assert float((after["boxes"] - want["boxes"]).abs().max()) <= 1e-4 * max(1.0, float(want["boxes"].abs().max()))
    assert float((after["cov"] - want["cov"]).abs().max()) <= 1e-4 * max(1.0, float(want["cov"].abs().max()))


def test_no_sparse_launch_reads_what_an_earlier_image_left():
    """The same property at the source: poison the tower's re-used buffers with NaN between two evaluations of a frame -- every value the
    second evaluation's candidates read is computed from this image (needed cells) or read as 0.0 (need bits), so nothing changes."""
    m = build(dropout_rate=0.0)
    pl = planted((256, 384), 1, seed=9)
    hp = hotpath.HotPath(pl.shapes, pl.anchors, hotpath.PathParams(), n_runs=1, has_cls_var=True, cov_dims=4, device="cuda:0")
    frame = torch.randint(0, 256, (3, 256, 384), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))

    def run():
        def hook(partial):
            hp.select(pl.cls, pl.cls_var, draw_id=5)
            return sparse.LiveBlocks(hp)
        ho = m(frame, sparse_bbox=hook)
        det = hp.finish("bayes_od", pl.cls, ho.delta, pl.cls_var, ho.reg_var, (256, 384), (256, 384))
        n, k = int(hp.n_total.item()), det.count()
        return hp.cand_delta[:n].clone(), hp.cand_reg_var[:n].clone(), det.boxes[:k].clone(), det.cov[:k].clone()

    first = run()
    pools = list(m.head._sparse_pool.values())
    assert pools and all(len(p) > 0 for p in pools)
    for pool in pools:
        for t in pool.values():
            t.fill_(float("nan"))
    second = run()
    assert len(first[2]) > 0
    for a, b in zip(first, second):
        assert bool(torch.isfinite(b).all()) and torch.equal(a, b)


def test_plain_model_without_variance_heads_takes_the_sparse_order_too():
    """BASELINE configs[3]'s model (retinanet_R_50_FPN_1x: no cls_var / bbox_cov heads) under anchor_statistics: dense against sparse."""
    m = build(plain=True)
    frame = torch.randint(0, 256, (3, 256, 384), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(8))
    pl = synthetic.planted_head_outputs((256, 384), 1, seed=13, num_boxes=6, with_cls_var=False, with_reg_var=False).to("cuda")
    hp = hotpath.HotPath(pl.shapes, pl.anchors, hotpath.PathParams(), n_runs=1, has_cls_var=False, cov_dims=0, device="cuda:0")
    dense = m(frame)
    assert dense.cls_var is None and dense.reg_var is None
    want = hp.run_image("anchor_statistics", pl.cls, dense.delta, None, None, (256, 384), (256, 384), draw_id=2)

    def hook(partial):
        assert partial.cls_var is None
        hp.select(pl.cls, None, draw_id=2)
        return sparse.LiveBlocks(hp)

    sp = m(frame, sparse_bbox=hook)
    got = hp.finish("anchor_statistics", pl.cls, sp.delta, None, None, (256, 384), (256, 384))
    k = got.count()
    assert k == want.count() and k > 0 and torch.equal(got.classes[:k], want.classes[:k])
    assert float((got.boxes[:k] - want.boxes[:k]).abs().max()) <= 1e-4 * max(1.0, float(want.boxes[:k].abs().max()))
    assert float((got.cov[:k] - want.cov[:k]).abs().max()) <= 1e-4 * max(1.0, float(want.cov[:k].abs().max()))


def test_an_image_without_candidates_runs_no_block_and_gives_no_detections():
    """The default order of apply_net must survive an empty image: no candidate (PI:300-308 selects nothing) -> every live list is empty, every
    launch of the bbox side exits at once, the path finishes with zero detections -- as the dense order does."""
    m = build(dropout_rate=0.0)
    frame = torch.randint(0, 256, (3, 256, 384), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(12))
    pl = synthetic.planted_head_outputs((256, 384), 1, seed=31, num_boxes=0).to("cuda")
    hp = hotpath.HotPath(pl.shapes, pl.anchors, hotpath.PathParams(), n_runs=1, has_cls_var=True, cov_dims=4, device="cuda:0")
    seen = {}

    def hook(partial):
        hp.select(pl.cls, pl.cls_var, draw_id=4)
        seen["lb"] = sparse.LiveBlocks(hp)
        return seen["lb"]

    sp = m(frame, sparse_bbox=hook)
    got = hp.finish("bayes_od", pl.cls, sp.delta, pl.cls_var, sp.reg_var, (256, 384), (256, 384))
    assert int(hp.n_total.item()) == 0 and got.count() == 0
    lv = [tuple(s) for s in pl.shapes]
    assert seen["lb"].fraction(block_table(lv, 1, "cuda"), 0) == 0.0 and seen["lb"].fraction(block_table(lv, 1, "cuda"), 4) == 0.0
    dense = m(frame)
    want = hp.run_image("bayes_od", pl.cls, dense.delta, pl.cls_var, dense.reg_var, (256, 384), (256, 384), draw_id=4)
    assert want.count() == 0
