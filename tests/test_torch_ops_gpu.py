"""torch.ops.pod_mi355x.* (SURVEY 8b): the registered operators give exactly what the ctypes path gives, refuse CPU tensors
and malformed arguments."""
import os

import pytest
import torch

import pod_compare_amd.torch_ops  # noqa: F401  (registers the library)
from oracle import pod_oracle as po
from tests.helpers import GOLDEN, Golden, assert_close
from tests.test_hip_parity import make_path
from tests.test_nms_gpu import clustered

pytestmark = pytest.mark.gpu


def test_predict_operator_equals_hotpath_and_reference_counts():
    g = Golden(os.path.join(GOLDEN, "cfg3_bayes_od_mc10_s31.npz"))
    ho = g.head_outputs().to("cuda")
    image, out = list(g.meta["image"]), list(g.meta["out"])
    b, c, s, k, p = torch.ops.pod_mi355x.predict(ho.cls, ho.delta, ho.cls_var, ho.reg_var, ho.anchors, "bayes_od", image, out, draw_id=4)
    hp = make_path(ho)
    det = hp.run("bayes_od", ho.cls, ho.delta, ho.cls_var, ho.reg_var, image_size=tuple(image), out_size=tuple(out), draw_id=4)
    m = det.count()
    assert b.shape == (m, 4) and c.shape == (m, 4, 4) and k.dtype == torch.int64 and p.shape == (m, 7)
    assert torch.equal(b, det.boxes[:m]) and torch.equal(c, det.cov[:m]) and torch.equal(s, det.scores[:m])
    assert torch.equal(k, det.classes[:m].long()) and torch.equal(p, det.probs[:m])
    assert abs(m - g.t("pred_boxes").shape[0]) <= 1          # native draws: the reference's detections up to a threshold case
    # a model without variance heads: empty lists
    g1 = Golden(os.path.join(GOLDEN, "cfg4_anchor_stats_plain_s41.npz"))
    h1 = g1.head_outputs().to("cuda")
    b1, c1, s1, k1, p1 = torch.ops.pod_mi355x.predict(h1.cls, h1.delta, [], [], h1.anchors, "anchor_statistics", list(g1.meta["image"]),
                                                      list(g1.meta["out"]), affinity_thresh=0.9)
    assert torch.equal(k1.cpu(), g1.t("pred_classes"))
    assert_close(b1.cpu(), g1.t("pred_boxes"), "boxes")
    assert_close(c1.cpu(), g1.t("pred_boxes_covariance"), "cov")


def test_predict_operator_follows_new_anchor_tensors_without_growing_its_cache():
    """detectron2's anchor generator returns NEW tensors on every forward.  The operator's workspace keeps a copy of the
    anchors: a call with other anchor tensors of the same geometry must decode against THOSE (not the cached copy), must not be
    fooled by a recycled address, and must not add a workspace per call."""
    from pod_compare_amd import torch_ops
    g = Golden(os.path.join(GOLDEN, "cfg2_bayes_od_regclsvar_s21.npz"))
    ho = g.head_outputs().to("cuda")
    image, out = list(g.meta["image"]), list(g.meta["out"])
    call = lambda anchors: torch.ops.pod_mi355x.predict(ho.cls, ho.delta, ho.cls_var, ho.reg_var, anchors, "bayes_od", image, out, draw_id=9)
    ref = call(ho.anchors)
    n_paths = len(torch_ops._PATHS)
    for rep in range(5):
        shifted = [a + 3.0 for a in ho.anchors]                  # new tensors (possibly at recycled addresses), other values
        got = call(shifted)
        assert torch.equal(got[3], ref[3]) and not torch.equal(got[0], ref[0])          # same classes, boxes moved with the anchors
        del shifted
        same = call([a.clone() for a in ho.anchors])             # new tensors again, the original values
        assert all(torch.equal(x, y) for x, y in zip(same, ref))
    assert len(torch_ops._PATHS) == n_paths
    assert len(torch_ops._PATHS) <= torch_ops._MAX_PATHS


def test_nms_and_nll_operators():
    boxes, scores, classes = clustered(1000, 5, (1344.0, 768.0))
    keep = torch.ops.pod_mi355x.nms_cluster(boxes.cuda(), scores.cuda(), classes.cuda(), 0.5, 100, 3)
    assert torch.equal(keep.cpu(), po.class_aware_nms(boxes, scores, classes.long(), 0.5)[:100])
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(50, 4, 4, generator=gen)
    cov = a @ a.transpose(1, 2) + 0.1 * torch.eye(4)
    means, gt = 100 * torch.rand(50, 4, generator=gen), 100 * torch.rand(50, 4, generator=gen)
    nll = torch.ops.pod_mi355x.reg_nll(means.cuda(), cov.cuda(), gt.cuda())
    assert_close(nll.cpu(), po.reg_nll(means, cov, gt), "nll", rtol=1e-5, atol=1e-4)


def test_operators_reject_bad_arguments():
    with pytest.raises(NotImplementedError):
        torch.ops.pod_mi355x.reg_nll(torch.zeros(2, 4), torch.zeros(2, 4, 4), torch.zeros(2, 4))
    with pytest.raises(RuntimeError):
        torch.ops.pod_mi355x.reg_nll(torch.zeros(2, 4, device="cuda"), torch.zeros(2, 3, 3, device="cuda"), torch.zeros(2, 4, device="cuda"))
    with pytest.raises(RuntimeError):
        torch.ops.pod_mi355x.nms_cluster(torch.zeros(4, 4, device="cuda", dtype=torch.float64), torch.zeros(4, device="cuda"),
                                         torch.zeros(4, device="cuda"), 0.5, 100, 3)


def test_wino_conv_operator_equals_conv2d():
    """torch.ops.pod_mi355x.wino_conv3x3: the head's convolution as an operator (channels-last and plane outputs)."""
    import torch.nn.functional as F
    from pod_compare_amd.wino import block_table, level_pixel_offsets
    levels, copies, C, K = [(23, 40), (6, 10)], 2, 64, 63
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(K, C, 3, 3, device="cuda", generator=g) * 0.06
    b = torch.zeros(64, device="cuda")
    b[:K] = torch.randn(K, device="cuda", generator=g)
    xs = [torch.randn(copies, C, h, wd, device="cuda", generator=g) for h, wd in levels]
    src = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).contiguous()
    U = torch.ops.pod_mi355x.wino_filter_transform(w)
    table = block_table(levels, copies, "cuda")
    offs = level_pixel_offsets(levels, copies)
    nhwc = torch.ops.pod_mi355x.wino_conv3x3(src, U, b, table, K, src.shape[0] * 64, relu=True)
    planes = torch.ops.pod_mi355x.wino_conv3x3(src, U, b, table, K, offs[-1] * K, planes=True)
    for i, (x, (h, wd)) in enumerate(zip(xs, levels)):
        want = F.conv2d(x, w, b[:K], padding=1)
        got_p = planes[offs[i] * K:offs[i + 1] * K].view(copies, K, h, wd)
        got_n = nhwc[offs[i]:offs[i + 1]].view(copies, h, wd, 64).permute(0, 3, 1, 2)[:, :K]
        scale = max(1.0, float(want.abs().max()))
        assert float((got_p - want).abs().max()) <= 2e-5 * scale and float((got_n - want.relu()).abs().max()) <= 2e-5 * scale
    with pytest.raises(Exception):
        torch.ops.pod_mi355x.wino_conv3x3(src.cpu(), U, b, table, K, src.shape[0] * 64)
    with pytest.raises(Exception):
        torch.ops.pod_mi355x.wino_conv3x3(src, U[:-1], b, table, K, src.shape[0] * 64)


def test_trunk_operators_equal_conv2d_and_max_pool():
    """torch.ops.pod_mi355x.{conv1x1_filter_split, conv1x1_split, stem7x7_filter_split, stem7x7_split, maxpool3x3s2_cl}: the channels-last
    trunk's kernels as operators, against torch on the same tensors."""
    import torch.nn.functional as F
    ops = torch.ops.pod_mi355x
    g = torch.Generator(device="cuda").manual_seed(8)
    frame = torch.randint(0, 256, (3, 70, 122), dtype=torch.uint8, device="cuda", generator=g)
    mean, std = torch.tensor([103.53, 116.28, 123.675], device="cuda"), torch.tensor([1.0, 57.0, 58.0], device="cuda")
    w7 = torch.randn(64, 3, 7, 7, device="cuda", generator=g) * 0.1
    b7 = torch.randn(64, device="cuda", generator=g)
    y = ops.stem7x7_split(frame, ops.stem7x7_filter_split(w7), b7, mean, std, 96, 128, relu=True)
    x = F.pad((frame.float() - mean.view(3, 1, 1)) / std.view(3, 1, 1), (0, 128 - 122, 0, 96 - 70)).unsqueeze(0)
    want = F.conv2d(x, w7, b7, stride=2, padding=3).relu()
    assert tuple(y.shape) == (48 * 64, 64)
    assert float((y.view(1, 48, 64, 64).permute(0, 3, 1, 2) - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    p = ops.maxpool3x3s2_cl(y, 48, 64)
    assert torch.equal(p.view(1, 24, 32, 64).permute(0, 3, 1, 2), F.max_pool2d(y.view(1, 48, 64, 64).permute(0, 3, 1, 2), 3, 2, 1))
    w1 = torch.randn(128, 64, 1, 1, device="cuda", generator=g) * 0.2
    b1 = torch.randn(128, device="cuda", generator=g)
    ws = ops.conv1x1_filter_split(w1)
    res = torch.randn(12 * 16, 128, device="cuda", generator=g)
    for splits in (1, 2):
        z = ops.conv1x1_split(p, ws, b1, res, 24, 32, 2, 128, relu=True, n_splits=splits)
        wantz = (F.conv2d(p.view(1, 24, 32, 64).permute(0, 3, 1, 2), w1, b1, stride=2) + res.view(1, 12, 16, 128).permute(0, 3, 1, 2)).relu()
        assert float((z.view(1, 12, 16, 128).permute(0, 3, 1, 2) - wantz).abs().max()) <= 2e-5 * max(1.0, float(wantz.abs().max()))
    w3 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05                     # FPN's p6 / p7 form: patch matrix + the 1x1 kernel
    cols = ops.im2col3x3s2_cl(p, 24, 32, relu=True)
    z3 = ops.conv1x1_split(cols, ops.conv1x1_filter_split(w3.permute(0, 2, 3, 1).reshape(64, 576, 1, 1).contiguous()), b7, None, 12, 16, 1, 64)
    want3 = F.conv2d(p.view(1, 24, 32, 64).permute(0, 3, 1, 2).relu(), w3, b7, stride=2, padding=1)
    assert float((z3.view(1, 12, 16, 64).permute(0, 3, 1, 2) - want3).abs().max()) <= 2e-5 * max(1.0, float(want3.abs().max()))
    for bad in (lambda: ops.im2col3x3s2_cl(p, 24, 31), lambda: ops.conv1x1_split(p.cpu(), ws, b1, None, 24, 32, 1, 128), lambda: ops.conv1x1_split(p, ws[:-1], b1, None, 24, 32, 1, 128),
                lambda: ops.conv1x1_split(p, ws, b1, None, 24, 32, 3, 128), lambda: ops.conv1x1_split(p, ws, b1, None, 24, 32, 1, 128, n_splits=3),
                lambda: ops.stem7x7_split(frame, ops.stem7x7_filter_split(w7), b7, mean, None, 96, 128), lambda: ops.stem7x7_split(frame, ops.stem7x7_filter_split(w7), b7, None, None, 64, 128),
                lambda: ops.maxpool3x3s2_cl(y, 48, 63)):
        with pytest.raises(Exception):
            bad()
