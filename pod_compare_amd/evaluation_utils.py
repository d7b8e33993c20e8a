"""Ground-truth matching and proper scoring rules on the GPU, with the reference's names.

Mirrors core/evaluation_tools/evaluation_utils.py (EU) `match_predictions_to_groundtruth` and
core/evaluation_tools/scoring_rules.py (SR) `compute_reg_scores`, `compute_reg_scores_fn`,
`retinanet_compute_cls_scores`: same arguments (dicts keyed by image id), same nested result dict of tensors, same
row order (prediction-dict key order, then ground-truth order, true positive = highest max class probability).
The reference re-concatenates its result tensors inside a Python loop over every ground-truth box (EU:222-360);
here one kernel launch labels the whole data set and the partitions are gathered with index tensors on the device.
"""
import math
from typing import Dict

import torch

from . import hip

MATCH_MAX_DET = 128


def match_predictions_to_groundtruth(predicted_box_means: Dict, predicted_cls_probs: Dict, predicted_box_covariances: Dict,
                                     gt_box_means: Dict, gt_cat_idxs: Dict, iou_min: float = 0.1, iou_correct: float = 0.7,
                                     device="cuda"):
    """EU:191-367."""
    lib = hip.load()
    dev = torch.device(device)
    keys = list(predicted_box_means.keys())
    f = lambda t: t.to(dev, torch.float32)
    k = next(iter(predicted_cls_probs.values())).shape[1] if keys else 1
    pb = torch.cat([f(predicted_box_means[i]).reshape(-1, 4) for i in keys]) if keys else torch.zeros((0, 4), device=dev)
    pp = torch.cat([f(predicted_cls_probs[i]).reshape(-1, k) for i in keys]) if keys else torch.zeros((0, k), device=dev)
    pc = torch.cat([f(predicted_box_covariances[i]).reshape(-1, 4, 4) for i in keys]) if keys else torch.zeros((0, 4, 4), device=dev)
    nd = [int(predicted_box_means[i].shape[0]) for i in keys]
    ng = [int(gt_box_means[i].shape[0]) if i in gt_box_means else 0 for i in keys]
    if any(n > MATCH_MAX_DET for n in nd):
        raise hip.PodError("more than {} detections in one image".format(MATCH_MAX_DET))
    has_gt = [i for i in keys if i in gt_box_means]
    gb = torch.cat([f(gt_box_means[i]).reshape(-1, 4) for i in has_gt]) if has_gt else torch.zeros((0, 4), device=dev)
    gc = torch.cat([f(gt_cat_idxs[i]).reshape(-1, 1) for i in has_gt]) if has_gt else torch.zeros((0, 1), device=dev)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=dev)
    det_off = i32([0] + torch.tensor(nd).cumsum(0).tolist() if nd else [0])
    gt_off = i32([0] + torch.tensor(ng).cumsum(0).tolist() if ng else [0])
    det_img = torch.repeat_interleave(torch.arange(len(keys), dtype=torch.int32, device=dev), i32(nd).long()) if nd else i32([])
    gt_img = torch.repeat_interleave(torch.arange(len(keys), dtype=torch.int32, device=dev), i32(ng).long()) if ng else i32([])
    D, G = int(pb.shape[0]), int(gb.shape[0])
    gt_fn = torch.zeros(max(G, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(max(G, 1), dtype=torch.int32, device=dev)
    midx = torch.zeros((max(G, 1), MATCH_MAX_DET), dtype=torch.int32, device=dev)
    miou = torch.zeros((max(G, 1), MATCH_MAX_DET), dtype=torch.float32, device=dev)
    det_fp = torch.zeros(max(D, 1), dtype=torch.int32, device=dev)
    P = hip.ptr
    pb, pp, gb = pb.contiguous(), pp.contiguous(), gb.contiguous()
    hip.check(lib.pod_match_groundtruth(P(pb), P(pp), P(det_off), P(det_img.contiguous()), D, P(gb), P(gt_off), P(gt_img.contiguous()), G,
                                        k, float(iou_min), float(iou_correct), P(gt_fn), P(cnt), P(midx), P(miou), P(det_fp),
                                        hip.current_stream()), "pod_match_groundtruth")
    gt_fn, cnt, midx, miou, det_fp = gt_fn[:G], cnt[:G], midx[:G], miou[:G], det_fp[:D]
    rank = torch.arange(MATCH_MAX_DET, device=dev)[None, :]
    tp_g = (cnt > 0).nonzero().squeeze(1)
    tp_d = midx[tp_g, 0].long()
    dup_mask = (rank >= 1) & (rank < cnt[:, None])
    dup_g, dup_r = dup_mask.nonzero(as_tuple=True)                    # row-major: ground-truth order, then rank
    dup_d = midx[dup_g, dup_r].long()
    fp_d = det_fp.bool().nonzero().squeeze(1)
    fn_g = gt_fn.bool().nonzero().squeeze(1)
    part = lambda d, g, iou: {"predicted_box_means": pb[d], "predicted_box_covariances": pc[d], "predicted_cls_probs": pp[d],
                              "gt_box_means": gb[g], "gt_cat_idxs": gc[g], "iou_with_ground_truth": iou}
    return {"true_positives": part(tp_d, tp_g, miou[tp_g, 0]),
            "duplicates": part(dup_d, dup_g, miou[dup_g, dup_r]),
            "false_positives": {"predicted_box_means": pb[fp_d], "predicted_box_covariances": pc[fp_d], "predicted_cls_probs": pp[fp_d]},
            "false_negatives": {"gt_box_means": gb[fn_g], "gt_cat_idxs": gc[fn_g]}}


def compute_reg_scores(input_matches: Dict, valid_idxs: torch.Tensor) -> Dict:
    """SR:45-81: ignorance (NLL of N(mean, cov + 1e-2 I) at the ground truth, kernel pod_reg_nll) and MSE."""
    means = input_matches["predicted_box_means"][valid_idxs].contiguous()
    covs = input_matches["predicted_box_covariances"][valid_idxs].contiguous()
    gt = input_matches["gt_box_means"][valid_idxs].contiguous()
    if means.shape[0] == 0:
        return {"ignorance_score_mean": None, "mean_squared_error": None}
    lib = hip.load()
    nll = torch.empty(means.shape[0], dtype=torch.float32, device=means.device)
    hip.check(lib.pod_reg_nll(hip.ptr(means), hip.ptr(covs), hip.ptr(gt), means.shape[0], hip.ptr(nll), hip.current_stream()), "pod_reg_nll")
    return {"ignorance_score_mean": float(nll.mean()), "mean_squared_error": float(((means - gt) ** 2).mean())}


def compute_reg_scores_fn(false_positives: Dict, valid_idxs: torch.Tensor) -> Dict:
    """SR:84-114: mean differential entropy of N(mean, cov + 1e-2 I) = 0.5 log det(2 pi e Sigma)."""
    covs = false_positives["predicted_box_covariances"][valid_idxs]
    if covs.shape[0] == 0:
        return {"total_entropy_mean": None}
    sig = covs.double() + 1e-2 * torch.eye(4, dtype=torch.float64, device=covs.device)
    ent = 0.5 * torch.logdet(sig) + 2.0 * (1.0 + math.log(2.0 * math.pi))
    return {"total_entropy_mean": float(ent.mean())}


def retinanet_compute_cls_scores(input_matches: Dict, valid_idxs: torch.Tensor) -> Dict:
    """SR:6-42: mean of -log p(correct) in the multilabel formulation."""
    p = input_matches["predicted_score_of_gt_category"][valid_idxs]
    if p.shape[0] == 0:
        return {"ignorance_score_mean": None}
    return {"ignorance_score_mean": float((-torch.log(p)).mean())}


def eval_predictions_preprocess(predicted_instances, min_allowed_score: float = 0.0, is_odd: bool = False, device="cpu"):
    """EU:19-73 (SURVEY f-2, the on-disk result format read back): list of `instances_to_json` dicts -> per-image tensors.
    XYWH boxes back to XYXY; covariances back through T' = [[1,0,0,0],[0,1,0,0],[1,0,1,0],[0,1,0,1]] (the inverse of
    covar_xyxy_to_xywh's T), detections with category_id == -1 (unless `is_odd`) or max cls_prob < min_allowed_score
    dropped.  Vectorised per image instead of one torch.cat per detection."""
    from collections import defaultdict
    rows = defaultdict(list)
    for inst in predicted_instances:
        top = max(inst["cls_prob"])
        if (not is_odd and inst["category_id"] == -1) or top < min_allowed_score:
            continue
        rows[inst["image_id"]].append(inst)
    tinv = torch.tensor([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [1.0, 0, 1.0, 0], [0, 1.0, 0.0, 1.0]], dtype=torch.float64)
    boxes, probs, covs = {}, {}, {}
    for image_id, insts in rows.items():
        b = torch.tensor([i["bbox"] for i in insts], dtype=torch.float64)
        b[:, 2] += b[:, 0]
        b[:, 3] += b[:, 1]
        c = torch.tensor([i["bbox_covar"] for i in insts], dtype=torch.float64)
        boxes[image_id] = b.to(torch.float32).to(device)
        probs[image_id] = torch.tensor([i["cls_prob"] for i in insts], dtype=torch.float32, device=device)
        covs[image_id] = (tinv @ c @ tinv.t()).to(torch.float32).to(device)
    return {"predicted_boxes": boxes, "predicted_cls_probs": probs, "predicted_covar_mats": covs}


def eval_gt_preprocess(gt_instances, device="cpu"):
    """EU:76-92: COCO annotations -> per-image XYXY boxes and (n,1) category ids."""
    from collections import defaultdict
    rows = defaultdict(list)
    for g in gt_instances:
        rows[g["image_id"]].append(g)
    boxes, cats = {}, {}
    for image_id, gs in rows.items():
        b = torch.tensor([g["bbox"] for g in gs], dtype=torch.float64)
        b[:, 2] += b[:, 0]
        b[:, 3] += b[:, 1]
        boxes[image_id] = b.to(torch.float32).to(device)
        cats[image_id] = torch.tensor([[g["category_id"]] for g in gs], dtype=torch.float32, device=device)
    return {"gt_boxes": boxes, "gt_cat_idxs": cats}


def planted_ground_truth(planted_boxes: torch.Tensor, planted_classes: torch.Tensor, image_size, out_size):
    """Planted boxes (network-input pixels, synthetic.planted_head_outputs) as ground truth at the output resolution:
    scaled like the detections (IU:394-403) and clipped to the frame; category ids are class + 1 (BDD ids 1..7)."""
    sx, sy = out_size[1] / image_size[1], out_size[0] / image_size[0]
    b = planted_boxes.to(torch.float32) * torch.tensor([sx, sy, sx, sy], dtype=torch.float32, device=planted_boxes.device)
    b[:, 0::2] = b[:, 0::2].clamp(0.0, float(out_size[1]))
    b[:, 1::2] = b[:, 1::2].clamp(0.0, float(out_size[0]))
    return b, (planted_classes.to(torch.float32) + 1.0).reshape(-1, 1)


def score_against_planted(detections, heads, image_size, out_size, iou_min: float = 0.1, iou_correct: float = 0.7, device="cuda") -> Dict:
    """End-to-end "NLL parity" number of the metric: detections of the path (`DeviceDetections`, one per image) matched to
    the planted ground truth of their inputs (EU:191-367, the offline evaluation's defaults iou_min 0.1 / iou_correct 0.7),
    then SR:68-74 on the true positives.  Returns {"nll", "mse", true_positives, duplicates, false_positives,
    false_negatives}."""
    pb, pp, pc, gb, gc = {}, {}, {}, {}, {}
    for i, (det, h) in enumerate(zip(detections, heads)):
        m = det.count()
        pb[i], pp[i], pc[i] = det.boxes[:m], det.probs[:m], det.cov[:m]
        gb[i], gc[i] = planted_ground_truth(h.planted_boxes, h.planted_classes, image_size, out_size)
    res = match_predictions_to_groundtruth(pb, pp, pc, gb, gc, iou_min, iou_correct, device=device)
    tp = res["true_positives"]
    valid = torch.ones(tp["predicted_box_means"].shape[0], dtype=torch.bool, device=tp["predicted_box_means"].device)
    reg = compute_reg_scores(tp, valid)
    return {"nll": reg["ignorance_score_mean"], "mse": reg["mean_squared_error"],
            "true_positives": int(tp["predicted_box_means"].shape[0]), "duplicates": int(res["duplicates"]["predicted_box_means"].shape[0]),
            "false_positives": int(res["false_positives"]["predicted_box_means"].shape[0]),
            "false_negatives": int(res["false_negatives"]["gt_box_means"].shape[0])}
