"""ground-truth matching + scoring rules.  CPU: oracle vs the reference's own output (golden);
GPU: HIP kernel + host mirror vs the same golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import pod_oracle as po
from tests.helpers import GOLDEN, assert_close

PARTS = ("true_positives", "duplicates", "false_positives", "false_negatives")


def load():
    z = np.load(os.path.join(GOLDEN, "eval_matching.npz"))
    t = lambda k: torch.from_numpy(z[k])
    pk, gk = z["pred_keys"].tolist(), z["gt_keys"].tolist()
    pb = {i: t("in_pb_%d" % i) for i in pk}
    pp = {i: t("in_pp_%d" % i) for i in pk}
    pc = {i: t("in_pc_%d" % i) for i in pk}
    gb = {i: t("in_gb_%d" % i) for i in gk}
    gc = {i: t("in_gc_%d" % i) for i in gk}
    return z, pb, pp, pc, gb, gc


def check(res, z, tight=True):
    for part in PARTS:
        for name, val in res[part].items():
            ref = torch.from_numpy(z["%s__%s" % (part, name)])
            val = val.cpu()
            assert tuple(val.shape) == tuple(ref.shape), (part, name, val.shape, ref.shape)
            if name == "gt_cat_idxs":
                assert torch.equal(val.float(), ref.float()), (part, name)
            else:
                assert_close(val, ref, part + "." + name, rtol=1e-6, atol=1e-6)


def test_oracle_matching_equals_reference():
    z, pb, pp, pc, gb, gc = load()
    res = po.match_predictions_to_groundtruth(pb, pp, pc, gb, gc, 0.1, 0.7)
    check(res, z)
    tp = res["true_positives"]
    ign, mse = po.compute_reg_scores(tp["predicted_box_means"], tp["predicted_box_covariances"], tp["gt_box_means"])
    assert abs(ign - float(z["tp_ignorance"])) < 1e-4 and abs(mse - float(z["tp_mse"])) < 1e-3
    fp = res["false_positives"]
    assert abs(po.compute_reg_entropy(fp["predicted_box_means"], fp["predicted_box_covariances"]) - float(z["fp_entropy"])) < 1e-4
    score = torch.gather(tp["predicted_cls_probs"], 1, (tp["gt_cat_idxs"].squeeze(1) - 1).long().unsqueeze(1)).squeeze(1)
    assert abs(po.retinanet_cls_ignorance(score) - float(z["tp_cls_ignorance"])) < 1e-5


@pytest.mark.gpu
def test_hip_matching_equals_reference():
    from pod_compare_amd import evaluation_utils as ev
    z, pb, pp, pc, gb, gc = load()
    res = ev.match_predictions_to_groundtruth(pb, pp, pc, gb, gc, 0.1, 0.7)
    check(res, z)
    tp = res["true_positives"]
    valid = torch.ones(tp["predicted_box_means"].shape[0], dtype=torch.bool, device="cuda")
    reg = ev.compute_reg_scores(tp, valid)
    assert abs(reg["ignorance_score_mean"] - float(z["tp_ignorance"])) <= 1e-3      # NLL parity bar
    assert abs(reg["mean_squared_error"] - float(z["tp_mse"])) <= 1e-3
    fp = res["false_positives"]
    fv = torch.ones(fp["predicted_box_means"].shape[0], dtype=torch.bool, device="cuda")
    assert abs(ev.compute_reg_scores_fn(fp, fv)["total_entropy_mean"] - float(z["fp_entropy"])) <= 1e-3
    tp = dict(tp)
    tp["predicted_score_of_gt_category"] = torch.gather(tp["predicted_cls_probs"], 1, (tp["gt_cat_idxs"].squeeze(1) - 1).long().unsqueeze(1)).squeeze(1)
    assert abs(ev.retinanet_compute_cls_scores(tp, valid)["ignorance_score_mean"] - float(z["tp_cls_ignorance"])) <= 1e-5


@pytest.mark.gpu
def test_hip_matching_empty_and_no_groundtruth():
    from pod_compare_amd import evaluation_utils as ev
    b = torch.tensor([[0., 0., 10., 10.], [5., 5., 20., 20.]])
    res = ev.match_predictions_to_groundtruth({3: b}, {3: torch.rand(2, 7)}, {3: torch.eye(4).repeat(2, 1, 1)}, {}, {})
    assert res["false_positives"]["predicted_box_means"].shape[0] == 2 and res["true_positives"]["predicted_box_means"].shape[0] == 0
    res = ev.match_predictions_to_groundtruth({}, {}, {}, {}, {})
    assert res["true_positives"]["predicted_box_means"].shape[0] == 0


def _json_instances():
    import json
    from tests.helpers import fixture_paths
    out = []
    for p in fixture_paths("cfg"):
        out += json.loads(str(np.load(p)["json"]))
    out.append(dict(out[0], category_id=-1))          # unmapped category: dropped unless is_odd
    return out


def test_result_file_roundtrip_matches_reference_preprocess():
    """f-2: the JSON the path writes (instances_to_json, IU:454-502) read back by eval_predictions_preprocess (EU:19-73):
    this build's vectorised mirror against the reference's own function on the reference's own JSON records."""
    from oracle.refimport import load_reference_evaluation, reference_available
    from pod_compare_amd import evaluation_utils as ev
    insts = _json_instances()
    mine = ev.eval_predictions_preprocess(insts, min_allowed_score=0.3)
    assert sum(v.shape[0] for v in mine["predicted_boxes"].values()) > 10
    if not reference_available():
        pytest.skip("reference tree not present")
    eu, _ = load_reference_evaluation()
    ref = eu.eval_predictions_preprocess(insts, min_allowed_score=0.3)
    assert sorted(mine["predicted_boxes"]) == sorted(ref["predicted_boxes"])
    for key in ref["predicted_boxes"]:
        for name in ("predicted_boxes", "predicted_cls_probs", "predicted_covar_mats"):
            assert_close(mine[name][key], ref[name][key], name, rtol=1e-6, atol=1e-5)
    gts = [{"image_id": 4, "bbox": [10.0, 20.0, 30.0, 40.0], "category_id": 3}, {"image_id": 4, "bbox": [1.0, 2.0, 3.0, 4.0], "category_id": 1}]
    g = ev.eval_gt_preprocess(gts)
    assert g["gt_boxes"][4].tolist() == [[10.0, 20.0, 40.0, 60.0], [1.0, 2.0, 4.0, 6.0]] and g["gt_cat_idxs"][4].tolist() == [[3.0], [1.0]]


def test_covariance_survives_the_json_roundtrip():
    """covar_xyxy_to_xywh (IU:428-451) followed by EU:58-66 is the identity on the covariance."""
    from pod_compare_amd import evaluation_utils as ev, inference_utils
    from pod_compare_amd.structures import Boxes, Instances
    inst = Instances((720, 1280))
    inst.pred_boxes = Boxes(torch.tensor([[10., 20., 110., 220.], [300., 40., 380., 90.]]))
    inst.scores = torch.tensor([0.9, 0.6])
    inst.pred_classes = torch.tensor([2, 5])
    inst.pred_cls_probs = torch.tensor([[0.1, 0.1, 0.9, 0.1, 0.1, 0.1, 0.1], [0.1, 0.1, 0.1, 0.1, 0.1, 0.6, 0.1]])
    l = torch.randn(2, 4, 4)
    inst.pred_boxes_covariance = l @ l.transpose(1, 2) + torch.eye(4)
    js = inference_utils.instances_to_json(inst, 9, {i: i + 1 for i in range(7)})
    back = ev.eval_predictions_preprocess(js)
    assert_close(back["predicted_boxes"][9], inst.pred_boxes.tensor, "boxes", 1e-6, 1e-5)
    assert_close(back["predicted_covar_mats"][9], inst.pred_boxes_covariance, "cov", 1e-5, 1e-5)
