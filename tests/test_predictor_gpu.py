"""The reference's plugin surface on the GPU: build_predictor(cfg) -> predictor(input_im) -> Instances, driven by
the five BASELINE configs' YAMLs with a fake model that returns the golden fixtures' head tensors."""
import json
import os

import pytest
import torch

from pod_compare_amd import config, inference_utils, probabilistic_inference as pinf
from tests.helpers import GOLDEN, Golden, assert_close

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pod_compare_amd", "configs")
M = CFG + "/BDD-Detection/retinanet/"
I = CFG + "/Inference/"

CASES = [  # (fixture, model yaml, inference yaml)  = BASELINE.json configs[0..4]
    ("cfg1_standard_nms_plain_s11", "retinanet_R_50_FPN_1x.yaml", "standard_nms.yaml"),
    ("cfg2_bayes_od_regclsvar_s21", "retinanet_R_50_FPN_1x_reg_cls_var.yaml", "bayes_od.yaml"),
    ("cfg3_bayes_od_mc10_s31", "retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", "bayes_od_mc_dropout.yaml"),
    ("cfg4_anchor_stats_plain_s41", "retinanet_R_50_FPN_1x.yaml", "anchor_statistics.yaml"),
    ("cfg5_ensembles_pre_nms_s51", "retinanet_R_50_FPN_1x_reg_cls_var.yaml", "ensembles_pre_nms.yaml"),
    ("mc_dropout_plain_pre_nms_s81", "retinanet_R_50_FPN_1x_dropout.yaml", "mc_dropout_ensembles_pre_nms.yaml"),
    # post-NMS merges (README "Black Box" rows)
    ("post_nms_ensembles_s141", "retinanet_R_50_FPN_1x_reg_cls_var.yaml", "ensembles_post_nms.yaml"),
    ("post_nms_mc_dropout_s151", "retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", "mc_dropout_ensembles_post_nms.yaml"),
]


class FakeModel:
    """Stands in for ProbabilisticRetinaNet: returns stored head tensors (one run or all runs)."""

    def __init__(self, ho, runs):
        self.ho, self.runs = ho, runs
        self.cls_var_num_samples, self.test_topk_candidates, self.test_score_thresh = 10, 1000, 0.05
        self.test_nms_thresh, self.max_detections_per_image = 0.5, 100

    def __call__(self, image, num_mc_dropout_runs=-1):
        from pod_compare_amd.synthetic import HeadOutputs
        h = self.ho
        sel = (lambda lst: None if lst is None else [t[self.runs].contiguous() for t in lst])
        return HeadOutputs(sel(h.cls), sel(h.delta), sel(h.cls_var), sel(h.reg_var), h.anchors, h.shapes, h.num_anchors,
                           h.num_classes, h.image_size)


@pytest.mark.parametrize("fixture,model_yaml,inf_yaml", CASES, ids=[c[0] for c in CASES])
def test_predictor_matches_reference(fixture, model_yaml, inf_yaml):
    g = Golden(os.path.join(GOLDEN, fixture + ".npz"))
    cfg = config.setup_config(M + model_yaml, I + inf_yaml)
    ho = g.head_outputs().to("cuda")
    n = g.spec["runs"]
    if cfg.PROBABILISTIC_INFERENCE.MC_DROPOUT.ENABLE:
        cfg.PROBABILISTIC_INFERENCE.MC_DROPOUT.NUM_RUNS = n      # the fixtures use fewer runs than the YAML's 10
    if cfg.PROBABILISTIC_INFERENCE.INFERENCE_MODE == "ensembles":
        members = [FakeModel(ho, [r]) for r in range(n)]
        pred = pinf.build_predictor(cfg, model=members[0], model_list=members)
    else:
        pred = pinf.build_predictor(cfg, model=FakeModel(ho, list(range(n))))
    assert isinstance(pred, pinf.RetinaNetProbabilisticPredictor)
    pred.eps_fn = g.eps_source()
    h, w = g.meta["image"]
    input_im = [{"image": torch.zeros((3, h, w), device="cuda"), "height": g.meta["out"][0], "width": g.meta["out"][1],
                 "image_id": g.meta["seed"]}]
    res = pred(input_im)
    assert res.image_size == tuple(g.meta["out"])
    assert res.pred_classes.dtype == torch.int64
    assert torch.equal(res.pred_classes.cpu(), g.t("pred_classes"))
    assert_close(res.pred_boxes.tensor.cpu(), g.t("pred_boxes"), "boxes")
    assert_close(res.pred_boxes_covariance.cpu(), g.t("pred_boxes_covariance"), "cov")
    assert_close(res.scores.cpu(), g.t("scores"), "scores", 2e-6, 1e-7)
    assert_close(res.pred_cls_probs.cpu(), g.t("pred_cls_probs"), "probs", 2e-6, 1e-7)
    js = inference_utils.instances_to_json(res, g.meta["seed"], {i: i + 1 for i in range(7)})
    ref = json.loads(str(g.z["json"]))
    assert [d["category_id"] for d in js] == [d["category_id"] for d in ref]
    for a, b in zip(js, ref):
        assert_close(a["bbox"], b["bbox"], "json bbox")
        assert_close(a["bbox_covar"], b["bbox_covar"], "json cov")
    # the device-side records (multi-GPU gather payload) carry the same JSON
    det = pred.last_detections
    js2 = inference_utils.records_to_json(det.records, det.count(), g.meta["seed"], 7, {i: i + 1 for i in range(7)})
    assert len(js2) == len(ref)
    for a, b in zip(js2, ref):
        assert a["category_id"] == b["category_id"]
        assert_close(a["bbox"], b["bbox"], "rec bbox")
        assert_close(a["bbox_covar"], b["bbox_covar"], "rec cov")


def test_anchorwise_surface_returns_reference_tuple():
    """retinanet_probabilistic_inference (PI:178-388) returns (boxes, cov, prob, classes int64, prob vectors)."""
    g = Golden(os.path.join(GOLDEN, "cfg1_standard_nms_plain_s11.npz"))
    cfg = config.setup_config(M + "retinanet_R_50_FPN_1x.yaml", I + "standard_nms.yaml")
    pred = pinf.build_predictor(cfg, model=FakeModel(g.head_outputs().to("cuda"), [0]))
    h, w = g.meta["image"]
    boxes, cov, prob, cls, pvec = pred.retinanet_probabilistic_inference([{"image": torch.zeros((3, h, w), device="cuda")}])
    assert cov == [] and cls.dtype == torch.int64                    # plain model: PI:381
    assert_close(boxes.cpu(), g.t("aw0_boxes"), "boxes")
    assert torch.equal(cls.cpu(), g.t("aw0_cls"))
    assert_close(prob.cpu(), g.t("aw0_prob"), "prob", 2e-6, 1e-7)
    assert pvec.shape == (boxes.shape[0], 7)


def test_end_to_end_with_the_torch_model():
    """build_predictor(cfg) with the real (random-init) ResNet-50-FPN + probabilistic head: runs, returns well-formed
    Instances (random-init weights give few or no detections)."""
    cfg = config.setup_config(M + "retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", I + "bayes_od_mc_dropout.yaml")
    cfg.PROBABILISTIC_INFERENCE.MC_DROPOUT.NUM_RUNS = 3
    cfg.MODEL.WEIGHTS, cfg.OUTPUT_DIR = "", ""          # explicit random init (the yaml's ImageNet URL cannot be fetched)
    torch.manual_seed(0)
    pred = pinf.build_predictor(cfg)
    img = torch.randint(0, 256, (3, 180, 250), dtype=torch.uint8, device="cuda")
    res = pred([{"image": img, "height": 360, "width": 500, "image_id": 5}])
    m = len(res)
    assert res.image_size == (360, 500) and m <= 100
    assert res.pred_boxes.tensor.shape == (m, 4) and res.pred_boxes_covariance.shape == (m, 4, 4)
    assert res.pred_cls_probs.shape == (m, 7) and res.pred_classes.dtype == torch.int64
    assert bool(torch.isfinite(res.pred_boxes.tensor).all())
    inference_utils.instances_to_json(res, 5, {i: i + 1 for i in range(7)})


def _calibrated_checkpoint(tmp_path, seed):
    """A detectron2-named checkpoint whose FrozenBN statistics are the actual activation statistics of one frame (what
    training leaves behind): every BN output is ~N(0, 1), so -- unlike identity statistics on random-init convs, which let
    activations explode through ResNet-50 -- the head sees bounded features and emits its prior (p ~ 0.01, PR:454-455)."""
    from pod_compare_amd import checkpoint, modeling
    torch.manual_seed(seed)
    src = modeling.ProbabilisticRetinaNet(dropout_rate=0.2, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                          bbox_cov_loss="negative_log_likelihood").eval()

    def calibrate(bn, inputs):
        x = inputs[0]
        bn.running_mean.copy_(x.mean(dim=(0, 2, 3)))
        bn.running_var.copy_(x.var(dim=(0, 2, 3), unbiased=False).clamp(min=1e-6))

    hooks = [m.register_forward_pre_hook(calibrate) for m in src.modules() if isinstance(m, modeling.FrozenBatchNorm2d)]
    frame = torch.randint(0, 256, (3, 96, 160), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
    src(frame)
    for h in hooks:
        h.remove()
    out_dir = str(tmp_path / "BDD-Detection" / "retinanet" / "retinanet_R_50_FPN_1x_reg_cls_var_dropout" / "random_seed_0")
    os.makedirs(out_dir)
    torch.save({"model": checkpoint.to_detectron2_state_dict(src), "iteration": 89999}, os.path.join(out_dir, "model_final.pth"))
    with open(os.path.join(out_dir, "last_checkpoint"), "w") as f:
        f.write("model_final.pth")
    return src, frame


def test_loaded_checkpoint_gives_prior_scores_and_matches_the_cpu_model(tmp_path):
    """PI:78-84 on the GPU: build_predictor(cfg) loads `<OUTPUT_DIR>/last_checkpoint` (detectron2 names), folds the
    non-identity FrozenBN statistics, and the MIOpen forward of the loaded model equals the CPU forward of the source model
    (unfolded conv -> BN) -- the numeric pin of row a1.  With sane statistics the scores are the head's prior, not 1.0."""
    src, frame = _calibrated_checkpoint(tmp_path, 21)
    cfg = config.setup_config(M + "retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", I + "bayes_od_mc_dropout.yaml", random_seed=0,
                              data_dir=str(tmp_path), is_testing=True)
    pred = pinf.build_predictor(cfg)
    assert pred.model.loaded_from.endswith("random_seed_0/model_final.pth")
    want = src(frame)                                             # CPU, eval (no dropout), unfolded
    got = pred.model(frame.cuda())                                # GPU, folded, fused conv tails
    for name in ("cls", "delta", "cls_var", "reg_var"):
        for a, b in zip(getattr(want, name), getattr(got, name)):
            scale = max(1.0, float(a.abs().max()))
            assert float((a - b.cpu()).abs().max()) <= 2e-3 * scale, name      # different conv algorithms (MIOpen vs mkldnn), fp32
    p = torch.sigmoid(torch.cat([t.reshape(-1) for t in got.cls]))
    assert 0.002 < float(p.min()) and float(p.max()) < 0.05 and abs(float(p.median()) - 0.01) < 0.003
    assert abs(float(torch.cat([t.reshape(-1) for t in got.cls_var]).median()) + 10.0) < 0.5          # PR:458-470 bias -10
    res = pred([{"image": frame.cuda(), "height": 96, "width": 160, "image_id": 3}])                # MC dropout, N = 10
    assert len(res) == 0                                          # every score is below 0.05: an empty, well-formed Instances
    assert res.pred_boxes_covariance.shape == (0, 4, 4)


class Fp64Conv:
    """Stand-in for wino.WinoConv with the same call signature and NO arithmetic in common with it: decodes the block table into
    its images (so the table's meaning is checked on the way), evaluates each with F.conv2d in fp64 on the CPU, rounds to fp32 and
    applies bias + ReLU + dropout with pod_bias_act on the same channels-last buffer -- the same Philox fields as the fused store."""

    def __init__(self, conv):
        self.w, self.b = conv.weight.detach().double().cpu(), conv.bias.detach().float()
        self.K, self.C = int(conv.weight.shape[0]), int(conv.weight.shape[1])
        self.Kpad = (self.K + 63) // 64 * 64
        self.split = False          # (no replicas() here: the head then takes conv + pod_expand_dropout, the same masks as the fused store pass)

    def __call__(self, src, dst, table, relu=False, dropout_p=0.0, seed=0, offset=0, planes=False, epoch=None):   # (epoch: 0 in an eager forward, i.e. the plain seed: the masks pod_bias_act draws below)
        import torch.nn.functional as F
        from pod_compare_amd import hip
        rec = table.cpu().long()
        canvases = {(int(r[0]), int(r[1]), int(r[2]), int(r[3]) >> 24) for r in rec}
        x_all = src.cpu().double()
        if not planes:
            dst.zero_()
        for xin, yout, z, n in sorted(canvases):
            H, W = (z >> 12) & 0xFFF, z & 0xFFF
            for i in range(n):
                x = x_all[xin + i * H * W:xin + (i + 1) * H * W].view(1, H, W, self.C).permute(0, 3, 1, 2)
                y = F.conv2d(x, self.w, None, padding=1)[0].float()                         # (K, H, W)
                o = yout + i * H * W
                if planes:
                    dst.view(-1)[o * self.K:(o + H * W) * self.K].view(self.K, H, W).copy_(y + self.b.cpu().view(-1, 1, 1))
                else:
                    dst[o:o + H * W, :self.K].copy_(y.permute(1, 2, 0).reshape(H * W, self.K))
        if not planes:
            bias = torch.zeros(self.Kpad, device=dst.device)
            bias[:self.K].copy_(self.b)
            hip.check(hip.load().pod_bias_act(dst.data_ptr(), bias.data_ptr(), None, None, dst.numel(), self.Kpad, 1, 1 if relu else 0,
                                              float(dropout_p), seed, offset, hip.current_stream()), "pod_bias_act")
        return dst


@pytest.mark.parametrize("split", [True, False], ids=["K12-f16x3", "K11-fp32-mfma"])
def test_detections_through_the_winograd_head_equal_those_through_an_fp64_head(tmp_path, split, monkeypatch):
    """(both convolution kernels: the production split kernel and the fp32-MFMA kernel)
    Detection level, model -> hot path: a calibrated checkpoint with the head sharpened so that detections exist, N = 10 MC
    dropout runs, the same dropout masks and the same hot-path draws; once with every head convolution on pod_wino_conv3x3 and
    once with every head convolution evaluated in fp64 (Fp64Conv).  Same detections: classes and order identical, scores,
    boxes and covariances within the parity tolerance -- fp32 Winograd in the head moves no detection."""
    from pod_compare_amd import wino
    monkeypatch.setattr(wino, "SPLIT_BF16", split)
    src, frame = _calibrated_checkpoint(tmp_path, 33)
    cfg = config.setup_config(M + "retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", I + "bayes_od_mc_dropout.yaml", random_seed=0,
                              data_dir=str(tmp_path), is_testing=True)
    pred = pinf.build_predictor(cfg)
    head = pred.model.head
    with torch.no_grad():
        for conv in list(head.cls_subnet) + list(head.bbox_subnet):
            conv.weight.mul_(3.0)                                   # std-0.01 filters would shrink the activations to nothing
        spread = float(torch.cat([t.reshape(-1) for t in pred.model(frame.cuda()).cls]).std())
        head.cls_score.weight.mul_(1.5 / spread)                    # logits ~ N(-5, 1.5): a few per cent of the (anchor, class) pairs
        head.cls_score.bias.fill_(-5.0)                             # pass the 0.05 threshold instead of all sitting at the prior 0.01
        head.bbox_pred.weight.mul_(3.0)
    feats = [f.clone() for f in pred.model.fpn(pred.model.bottom_up(pred.model.preprocess_image(frame.cuda())))]
    pred.model.fpn.forward = lambda _: [f.clone() for f in feats]   # one set of features for both heads (MIOpen's backbone kernels
    inp = [{"image": frame.cuda(), "height": 96, "width": 160, "image_id": 7}]      # accumulate with atomics: not bit-reproducible)
    head._drop_calls = 0
    got = pred(inp)
    assert modeling_wino_split(head) == split
    real = head._wino
    try:
        head._wino = lambda conv: Fp64Conv(conv)
        head._drop_calls = 0
        want = pred(inp)
    finally:
        head._wino = real
    assert len(want) >= 5, "the sharpened head must produce detections for this test to mean anything"
    assert len(got) == len(want)
    assert torch.equal(got.pred_classes.cpu(), want.pred_classes.cpu())
    assert_close(got.scores.cpu(), want.scores.cpu(), "scores")
    bscale = want.pred_boxes.tensor.cpu().abs().amax(dim=1, keepdim=True).clamp(min=1.0)                # per detection: a corner near 0 of a
    assert_close(got.pred_boxes.tensor.cpu() / bscale, want.pred_boxes.tensor.cpu() / bscale, "boxes")   # 300-pixel box moves with ITS box
    scale = want.pred_boxes_covariance.cpu().abs().amax(dim=(1, 2), keepdim=True).clamp(min=1.0)       # per detection: an off-diagonal
    assert_close(got.pred_boxes_covariance.cpu() / scale, want.pred_boxes_covariance.cpu() / scale, "cov")   # entry is small against ITS matrix
    assert_close(got.pred_cls_probs.cpu(), want.pred_cls_probs.cpu(), "probs")


def modeling_wino_split(head) -> bool:
    from pod_compare_amd import modeling
    return modeling.wino_of(head.cls_subnet[1]).split


def test_eval_mode_trunk_sharing_on_the_gpu():
    """SURVEY f-4 on the GPU: without dropout the mean and variance branches of a subnet see the same trunk activation, so
    the head evaluates each trunk once (PR:518-523 runs it twice); the outputs equal two separate evaluations."""
    from pod_compare_amd import modeling
    torch.manual_seed(5)
    model = modeling.ProbabilisticRetinaNet(dropout_rate=0.0, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                            bbox_cov_loss="negative_log_likelihood").cuda().eval()
    modeling.fold_frozen_bn(model)
    for q in model.parameters():
        q.requires_grad_(False)
    feats = [torch.randn(1, 256, h, w, device="cuda") for h, w in ((24, 32), (12, 16), (6, 8))]
    calls = {"n": 0}
    real = model.head._trunk

    def counting(convs, feature, copies, dropout, level=0):
        calls["n"] += 1
        return real(convs, feature, copies, dropout, level)

    real_all = model.head._trunk_all_levels

    def counting_all(convs, x0, levels, copies, dropout):
        calls["n"] += len(levels)
        return real_all(convs, x0, levels, copies, dropout)

    real_grouped = model.head._trunks_grouped

    def counting_grouped(x0, levels, copies_c, copies_b, dropout):          # (both subnets, layer by layer in one launch each)
        calls["n"] += 2 * len(levels)
        return real_grouped(x0, levels, copies_c, copies_b, dropout)

    model.head._trunk = counting
    model.head._trunk_all_levels = counting_all
    model.head._trunks_grouped = counting_grouped
    cls, delta, cls_var, reg_var = model.head(feats, 1, mc_dropout=False)
    assert calls["n"] == 2 * len(feats)                           # one cls trunk + one box trunk per level, not four
    model.head._trunk, model.head._trunk_all_levels, model.head._trunks_grouped = real, real_all, real_grouped
    for l, f in enumerate(feats):
        tc, tb = f, f
        for conv in model.head.cls_subnet:
            tc = torch.relu(conv(tc))
        for conv in model.head.bbox_subnet:
            tb = torch.relu(conv(tb))
        for got, want in ((cls[l], model.head.cls_score(tc)), (cls_var[l], model.head.cls_var(tc)),
                          (delta[l], model.head.bbox_pred(tb)), (reg_var[l], model.head.bbox_cov(tb))):
            assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_ensemble_members_in_packed_strided_buffer_match_reference():
    """Config 5 data path on one GPU: members packed into the (M, packed) exchange buffer and handed to K1 as
    run-strided views (no re-layout) give the reference's result."""
    from pod_compare_amd import ensemble_dist
    g = Golden(os.path.join(GOLDEN, "cfg5_ensembles_pre_nms_s51.npz"))
    cfg = config.setup_config(M + "retinanet_R_50_FPN_1x_reg_cls_var.yaml", I + "ensembles_pre_nms.yaml")
    ho = g.head_outputs().to("cuda")
    n = g.spec["runs"]
    lay = ensemble_dist.MemberLayout.of(ho)
    stacked = torch.stack([lay.pack(pinf.run_slice(ho, r)) for r in range(n)])
    views = lay.views(stacked, ho)
    assert views.cls[0].stride(0) == lay.total
    pred = pinf.build_predictor(cfg, model=FakeModel(ho, [0]), model_list=[object()] * n)
    pred.eps_fn = g.eps_source()
    h, w = g.meta["image"]
    input_im = [{"image": torch.zeros((3, h, w), device="cuda"), "height": g.meta["out"][0], "width": g.meta["out"][1], "image_id": 1}]
    res = pred._run("standard_nms", input_im, views)
    assert torch.equal(res.pred_classes.cpu(), g.t("pred_classes"))
    assert_close(res.pred_boxes.tensor.cpu(), g.t("pred_boxes"), "boxes")
    assert_close(res.pred_boxes_covariance.cpu(), g.t("pred_boxes_covariance"), "cov")


def test_fused_relu_dropout_statistics():
    """pod_relu_dropout == dropout(relu(x), p): zeros where x <= 0, survivors scaled by 1/(1-p), keep rate 1-p,
    different masks for different counter offsets, deterministic for equal (seed, offset)."""
    from pod_compare_amd import hip
    lib = hip.load()
    x = torch.randn(3, 256, 37, 41, device="cuda")          # numel % 4 != 0 exercises the tail
    p = 0.2
    def run(off):
        y = x.clone()
        hip.check(lib.pod_relu_dropout(y.data_ptr(), y.numel(), p, 1234, off, hip.current_stream()), "pod_relu_dropout")
        return y
    y, y_again, y_other = run(0), run(0), run(1 << 34)
    assert torch.equal(y, y_again) and not torch.equal(y, y_other)
    pos = x > 0
    assert bool((y[~pos] == 0).all())
    kept = y[pos] != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 5e-3
    assert torch.allclose(y[pos][kept], x[pos][kept] / (1 - p), rtol=1e-6, atol=0)
    # neighbouring elements are not correlated (one Philox call feeds 4 elements)
    k = (y != 0).float().reshape(-1)[: (x.numel() // 4) * 4].reshape(-1, 4)
    m = pos.reshape(-1)[: (x.numel() // 4) * 4].reshape(-1, 4).all(1)
    c = torch.corrcoef(k[m].t())
    assert float((c - torch.eye(4, device="cuda")).abs().max()) < 0.02


def test_bn_folding_is_the_same_affine_map():
    from pod_compare_amd import modeling
    torch.manual_seed(3)
    net = modeling.ResNet50().cuda().eval()
    for mod in net.modules():                                   # non-trivial frozen statistics
        if isinstance(mod, modeling.FrozenBatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.2, 0.2)
            mod.running_mean.uniform_(-0.1, 0.1); mod.running_var.uniform_(0.5, 1.5)
    x = torch.randn(1, 3, 96, 128, device="cuda")
    with torch.no_grad():
        ref = net(x)
        assert modeling.fold_frozen_bn(net) == 53
        out = net(x)
    for a, b in zip(out, ref):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()))


@pytest.mark.parametrize("shape", [(3, 16, 12, 20), (2, 7, 6, 11), (1, 5, 3, 3)])     # H*W % 4 == 0 / != 0 / numel % 4 != 0
def test_bias_act_equals_the_torch_ops_it_replaces(shape):
    """pod_bias_act == dropout(relu((x + b[c]) + (r + rb[c])), p), every stage optional; with neither bias nor residual and the
    same counter offset it reproduces pod_relu_dropout's mask exactly."""
    from pod_compare_amd import hip
    lib = hip.load()
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x, r = torch.randn(shape, device="cuda", generator=g), torch.randn(shape, device="cuda", generator=g)
    b, rb = torch.randn(shape[1], device="cuda", generator=g), torch.randn(shape[1], device="cuda", generator=g)
    C, HW = shape[1], shape[2] * shape[3]

    def run(bias, res, res_bias, relu, p, off=0):
        y = x.clone()
        hip.check(lib.pod_bias_act(y.data_ptr(), hip.ptr(bias), hip.ptr(res), hip.ptr(res_bias), y.numel(), C, HW, relu, p, 99, off,
                                   hip.current_stream()), "pod_bias_act")
        return y

    v = lambda t: t.view(1, -1, 1, 1)
    assert torch.equal(run(b, None, None, 0, 0.0), x + v(b))
    assert torch.equal(run(b, None, None, 1, 0.0), torch.relu(x + v(b)))
    assert torch.equal(run(b, r, None, 1, 0.0), torch.relu((x + v(b)) + r))
    assert torch.equal(run(b, r, rb, 1, 0.0), torch.relu((x + v(b)) + (r + v(rb))))
    assert torch.equal(run(None, r, None, 0, 0.0), x + r)
    p = 0.25
    y = run(b, r, rb, 1, p, off=5 << 34)
    full = torch.relu((x + v(b)) + (r + v(rb)))
    kept = y != 0
    assert bool((y[full == 0] == 0).all())
    assert torch.allclose(y[kept], full[kept] / (1 - p), rtol=1e-6, atol=0)
    plain = x.clone()
    hip.check(lib.pod_relu_dropout(plain.data_ptr(), plain.numel(), p, 99, 5 << 34, hip.current_stream()), "pod_relu_dropout")
    assert torch.equal(run(None, None, None, 1, p, off=5 << 34), plain)
    # argument validation
    assert lib.pod_bias_act(x.data_ptr(), None, None, hip.ptr(rb), x.numel(), C, HW, 1, 0.0, 0, 0, hip.current_stream()) == -1
    assert lib.pod_bias_act(x.data_ptr(), None, None, None, x.numel(), C + 1, HW, 1, 0.0, 0, 0, hip.current_stream()) == -1


def test_fused_conv_tail_leaves_the_network_output_unchanged():
    """conv without bias + one pod_bias_act pass (bias, shortcut, ReLU) against torch's conv + add_ + add + clamp:
    the same head outputs through the whole folded ResNet-50-FPN without dropout, up to fp32 rounding (MIOpen may pick another
    solver for a bias-free conv, and ~60 layers of random weights amplify the last bit)."""
    from pod_compare_amd import modeling
    torch.manual_seed(11)
    net = modeling.ProbabilisticRetinaNet(num_classes=7, dropout_rate=0.2, cls_var_loss="loss_attenuation", bbox_cov_loss="negative_log_likelihood").cuda().eval()
    for mod in net.modules():
        if isinstance(mod, modeling.FrozenBatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.2, 0.2)
            mod.running_mean.uniform_(-0.1, 0.1); mod.running_var.uniform_(0.5, 1.5)
    assert modeling.fold_frozen_bn(net) > 40
    img = torch.rand(3, 200, 264, device="cuda") * 255
    outs = []
    with torch.no_grad():
        for fuse in (True, False):
            modeling.FUSE_CONV_TAIL = fuse
            try:
                outs.append(net(img, num_mc_dropout_runs=-1))
            finally:
                modeling.FUSE_CONV_TAIL = True
    for name in ("cls", "delta", "cls_var", "reg_var"):
        for a, b in zip(getattr(outs[0], name), getattr(outs[1], name)):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), name


def test_expand_dropout_statistics_and_independent_copies():
    from pod_compare_amd import hip
    lib = hip.load()
    src = torch.rand(1, 64, 24, 40, device="cuda") + 0.5
    copies, p = 7, 0.2
    dst = torch.empty((copies,) + tuple(src.shape[1:]), device="cuda")
    hip.check(lib.pod_expand_dropout(src.data_ptr(), dst.data_ptr(), src.numel(), copies, p, 5, 3 << 34, None, hip.current_stream()), "expand")
    kept = dst != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 5e-3
    assert torch.allclose(dst[kept], src.expand_as(dst)[kept] / (1 - p), rtol=1e-6, atol=0)
    masks = kept.reshape(copies, -1).float()
    c = torch.corrcoef(masks)
    assert float((c - torch.eye(copies, device="cuda")).abs().max()) < 0.02          # every copy has its own mask
    again = torch.empty_like(dst)
    hip.check(lib.pod_expand_dropout(src.data_ptr(), again.data_ptr(), src.numel(), copies, p, 5, 3 << 34, None, hip.current_stream()), "expand")
    assert torch.equal(dst, again)
    assert lib.pod_expand_dropout(src.data_ptr(), dst.data_ptr(), 6, 1, p, 0, 0, None, hip.current_stream()) == -1   # n % 4 != 0


def test_channels_last_trunk_matches_the_nchw_trunk():
    """Large maps run the head trunk channels-last (no MIOpen layout transposes); without dropout noise the outputs must be
    the NCHW trunk's, and the returned tensors must be NCHW planes either way."""
    from pod_compare_amd import modeling
    torch.manual_seed(5)
    head = modeling.ProbabilisticRetinaNetHead(256, 9, 7, 4, 0.01, 0.2, True, True, 4).cuda().eval()
    for mod in list(head.cls_subnet) + list(head.bbox_subnet):
        torch.nn.init.normal_(mod.weight, std=0.03); torch.nn.init.normal_(mod.bias, std=0.1)
    feats = [torch.randn(1, 256, 96, 100, device="cuda"), torch.randn(1, 256, 13, 17, device="cuda")]
    head.dropout_rate = 1e-9            # dropout path taken (MC mode), masks keep everything
    outs = []
    with torch.no_grad():
        for thr in (1, 10 ** 9):        # every level channels-last / none
            modeling.NHWC_TRUNK_MIN_CELLS = thr
            try:
                outs.append(head(feats, 3, mc_dropout=True))
            finally:
                modeling.NHWC_TRUNK_MIN_CELLS = 8192
    for a_list, b_list in zip(outs[0], outs[1]):
        for a, b in zip(a_list, b_list):
            assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max())   # different MIOpen solvers per layout (Winograd vs implicit GEMM)


def test_images_in_flight_on_several_streams_give_the_single_stream_results():
    """The image-sharded driver keeps consecutive images on different HIP streams; the predictor holds one workspace per
    stream.  Same head tensors, same Philox seeds: records must be bit-identical to the one-stream run."""
    g = Golden(os.path.join(GOLDEN, "cfg3_bayes_od_mc10_s31.npz"))
    cfg = config.setup_config(M + "retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", I + "bayes_od_mc_dropout.yaml")
    n = g.spec["runs"]
    cfg.PROBABILISTIC_INFERENCE.MC_DROPOUT.NUM_RUNS = n
    ho = g.head_outputs().to("cuda")
    pred = pinf.build_predictor(cfg, model=FakeModel(ho, list(range(n))))
    pred.return_device = True                                   # native Philox draws (no eps_fn): the production mode
    h, w = g.meta["image"]
    input_im = [{"image": torch.zeros((3, h, w), device="cuda"), "height": g.meta["out"][0], "width": g.meta["out"][1], "image_id": 1}]
    torch.cuda.synchronize()
    ref = pred(input_im)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    dets = []
    for j in range(9):
        with torch.cuda.stream(streams[j % 3]):
            dets.append(pred(input_im))
    torch.cuda.synchronize()
    assert len({d.buf.data_ptr() for d in dets}) == 9 and len(pred._paths) == 4        # default stream + 3
    m = ref.count()
    assert m > 0
    for d in dets:
        assert d.count() == m and torch.equal(d.records[:m], ref.records[:m])


@pytest.mark.parametrize("shape", [(3, 64, 12, 20), (2, 256, 7, 12), (1, 8, 2, 2), (2, 68, 9, 28)])     # full / partial tiles
def test_bias_act_to_nchw_is_transpose_then_bias_act(shape):
    from pod_compare_amd import hip
    lib = hip.load()
    N, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(N * C + H)
    x = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device="cuda", generator=g)
    for relu, p in ((1, 0.0), (0, 0.0), (1, 0.3)):
        out = torch.full(shape, float("nan"), device="cuda")
        hip.check(lib.pod_bias_act_to_nchw(x.data_ptr(), out.data_ptr(), b.data_ptr(), N, C, H * W, relu, p, 7, 2 << 34,
                                           hip.current_stream()), "to_nchw")
        ref = x.contiguous().clone()      # NCHW copy, then the in-place NCHW pass with the same counters
        hip.check(lib.pod_bias_act(ref.data_ptr(), b.data_ptr(), None, None, ref.numel(), C, H * W, relu, p, 7, 2 << 34,
                                   hip.current_stream()), "bias_act")
        assert out.is_contiguous() and torch.equal(out, ref), (relu, p)
    assert lib.pod_bias_act_to_nchw(x.data_ptr(), x.data_ptr(), None, N, C, H * W, 1, 0.0, 0, 0, hip.current_stream()) == -1
