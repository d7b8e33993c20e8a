"""Probabilistic detection metrics from a result file: the consumer of the "NLL parity" half of the metric.

Mirrors /root/reference/src/offline_evaluation/compute_probabilistic_metrics.py (PM): results + ground truth ->
ground-truth matching (EU:191-367, kernel pod_match_groundtruth) -> per-class classification / regression ignorance
scores (SR:6-114; the regression NLL on the kernel pod_reg_nll) -> PM's summary table.  Same arithmetic and the same
defaults (iou_min 0.1, iou_correct 0.7, classes [1, 3], nan-mean over the classes, PM:134-178); what is left out is the
reference's detectron2 plumbing (MetadataCatalog lookups, on-disk caches of the intermediate tensors, `launch`).

    python -m pod_compare_amd.compute_probabilistic_metrics --results coco_instances_results.json --gt val_coco_format.json
    python -m pod_compare_amd.compute_probabilistic_metrics --binary-results results.podr --gt val_coco_format.json

The matching and scoring functions come from `pod_compare_amd.evaluation_utils` (HIP); `ev=` lets a test substitute another
implementation of the same five functions.
"""
import argparse
import json
from typing import Dict, Optional, Sequence

import numpy as np
import torch

BDD_DATASET_ID_TO_CONTIGUOUS = {i + 1: i for i in range(7)}       # core/datasets/metadata.py: BDD ids 1..7


def probabilistic_metrics(predicted_instances: Sequence[dict], gt_instances: Sequence[dict],
                          cat_mapping_dict: Optional[Dict[int, int]] = None, iou_min: float = 0.1, iou_correct: float = 0.7,
                          min_allowed_score: float = 0.0, classes: Sequence[int] = (1, 3), device="cuda", ev=None) -> dict:
    """PM:81-178.  predicted_instances: the dicts of coco_instances_results.json; gt_instances: COCO `annotations`;
    cat_mapping_dict: dataset category id -> contiguous id (PM:69-79).  Returns {"counts": {...}, "average": {...},
    "per_class": {...}} with PM's key names."""
    if ev is None:
        from . import evaluation_utils as ev
    cat_mapping_dict = BDD_DATASET_ID_TO_CONTIGUOUS if cat_mapping_dict is None else cat_mapping_dict
    pred = ev.eval_predictions_preprocess(predicted_instances, min_allowed_score, device=device)          # EU:19-73
    gt = ev.eval_gt_preprocess(gt_instances, device=device)                                               # EU:76-92
    matched = ev.match_predictions_to_groundtruth(pred["predicted_boxes"], pred["predicted_cls_probs"], pred["predicted_covar_mats"],
                                                  gt["gt_boxes"], gt["gt_cat_idxs"], iou_min, iou_correct, device=device)   # EU:191-367
    dev = torch.device(device)
    for part in matched.values():                                                                         # PM:88-114
        if "gt_cat_idxs" in part:
            conv = torch.as_tensor([cat_mapping_dict[int(c)] for c in part["gt_cat_idxs"].reshape(-1).cpu().tolist()],
                                   dtype=torch.int64, device=dev)
            part["gt_converted_cat_idxs"] = conv
            if "predicted_cls_probs" in part:
                part["predicted_score_of_gt_category"] = torch.gather(part["predicted_cls_probs"], 1, conv.unsqueeze(1)).squeeze(1)
            part["gt_cat_idxs"] = conv
        else:   # false positives: the correct category is background = 1 - score of the predicted category
            probs, idx = part["predicted_cls_probs"].max(1) if part["predicted_cls_probs"].shape[0] else \
                (torch.zeros(0, device=dev), torch.zeros(0, dtype=torch.int64, device=dev))
            part["predicted_score_of_gt_category"] = 1.0 - probs
            part["predicted_cat_idxs"] = idx
    tp, fn, fp = matched["true_positives"], matched["false_negatives"], matched["false_positives"]
    per_class = []
    for class_idx in classes:                                                                             # PM:123-146
        tp_valid = tp["gt_converted_cat_idxs"] == class_idx
        fp_valid = fp["predicted_cat_idxs"] == class_idx
        per_class.append({"true_positives_cls_analysis": ev.retinanet_compute_cls_scores(tp, tp_valid),
                          "true_positives_reg_analysis": ev.compute_reg_scores(tp, tp_valid),
                          "false_positives_cls_analysis": ev.retinanet_compute_cls_scores(fp, fp_valid),
                          "false_positives_reg_analysis": ev.compute_reg_scores_fn(fp, fp_valid)})
    average = {}
    for key in per_class[0]:                                                                              # PM:148-178
        average[key] = {}
        for inner in per_class[0][key]:
            vals = np.array([c[key][inner] for c in per_class if c[key][inner] is not None], dtype=np.float64)
            average[key][inner] = float(np.nanmean(vals)) if vals.size else float("nan")
    return {"counts": {"true_positives": int(tp["predicted_box_means"].shape[0]), "duplicates": int(matched["duplicates"]["predicted_box_means"].shape[0]),
                       "false_positives": int(fp["predicted_box_means"].shape[0]), "false_negatives": int(fn["gt_box_means"].shape[0])},
            "average": average, "per_class": {int(c): d for c, d in zip(classes, per_class)}}


def format_table(res: dict) -> str:
    """PM:179-213 without prettytable."""
    a, c = res["average"], res["counts"]
    rows = [("Output Type", "Number of Instances", "Cls Ignorance Score", "Reg Ignorance Score"),
            ("True Positives:", c["true_positives"], "%.4f" % a["true_positives_cls_analysis"]["ignorance_score_mean"],
             "%.4f" % a["true_positives_reg_analysis"]["ignorance_score_mean"]),
            ("False Positives:", c["false_positives"], "%.4f" % a["false_positives_cls_analysis"]["ignorance_score_mean"],
             "%.4f" % a["false_positives_reg_analysis"]["total_entropy_mean"]),
            ("False Negatives:", c["false_negatives"], "-", "-")]
    w = [max(len(str(r[i])) for r in rows) for i in range(4)]
    line = "+" + "+".join("-" * (x + 2) for x in w) + "+"
    body = ["| " + " | ".join(str(r[i]).center(w[i]) for i in range(4)) + " |" for r in rows]
    return "\n".join([line, body[0], line] + body[1:] + [line])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--results", default="", help="coco_instances_results.json written by apply_net (AN:100-102)")
    ap.add_argument("--binary-results", default="", help="the binary sidecar instead (inference_utils.write_binary_results)")
    ap.add_argument("--gt", required=True, help="COCO-format ground truth json (its `annotations`)")
    ap.add_argument("--iou-min", type=float, default=0.1)
    ap.add_argument("--iou-correct", type=float, default=0.7)
    ap.add_argument("--min-allowed-score", type=float, default=0.0)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    if args.binary_results:
        from .apply_net import BDD_CAT_MAP
        from .inference_utils import binary_results_to_json
        predicted = binary_results_to_json(args.binary_results, BDD_CAT_MAP)
    else:
        with open(args.results, "r") as f:
            predicted = json.load(f)
    with open(args.gt, "r") as f:
        gt = json.load(f)["annotations"]
    res = probabilistic_metrics(predicted, gt, iou_min=args.iou_min, iou_correct=args.iou_correct, min_allowed_score=args.min_allowed_score,
                                device=args.device)
    print(format_table(res))
    return res


if __name__ == "__main__":
    main()
