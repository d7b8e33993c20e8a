"""Calibration and minimum-uncertainty errors from matched results: the tail of SURVEY 8 f-4.

Mirrors /root/reference/src/offline_evaluation/compute_calibration_errors.py (CE:19-302): results + ground truth -> ground-truth matching
(EU:191-367, kernel pod_match_groundtruth, as compute_probabilistic_metrics does) -> over every class of the mapping

  * regression expected / maximum calibration error (CE:206-261): per box coordinate, the Normal cdf of the ground truth under the
    predicted marginal, against 14 cumulative bins of width 1/15 ("Accurate uncertainties for deep learning using calibrated regression");
  * classification / regression minimum uncertainty error (CE:160-178, CE:263-292): detections sorted by entropy, the best threshold's
    balanced error between true positives and the rest;
  * classification marginal calibration error (CE:117-136): the flattened class probabilities of all detections against one-hot labels
    (false positives: all zero) handed to `calibration.get_calibration_error` -- the third-party package `uncertainty-calibration`
    (requirements.txt:16, >= 0.0.7), which is not installed in this image.  `marginal_calibration_error` below restates its published
    default (p = 2, debiased, 15 equal-mass bins for continuous scores: Kumar, Liang, Ma, "Verified Uncertainty Calibration", NeurIPS 2019)
    -- PARITY UNPINNED for that one function; what the reference hands to it is pinned (tests/test_calibration_cpu.py).

Everything else is pinned against the reference's own `main` run on seeded partitions (oracle/make_golden_calib.py,
tests/golden/calib_errors.npz).  Host-side torch: this is offline evaluation, not the images/sec path.

    python -m pod_compare_amd.compute_calibration_errors --results coco_instances_results.json --gt val_coco_format.json
"""
import argparse
import json
from typing import Dict, Optional

import numpy as np
import torch


def marginal_calibration_error(probs: np.ndarray, labels: np.ndarray, num_bins: int = 15) -> float:
    """`calibration.get_calibration_error(probs, labels)` for 1-D scores in [0, 1] and 0 / 1 labels, the package's defaults: L2, debiased
    (`unbiased_l2_ce`), `num_bins` equal-mass bins with upper edges half-way between neighbouring buckets, the last edge 1.0; scores with
    enough duplicates (fewer than n / 4 distinct values) are binned by their distinct values instead.  Restated from the paper and the
    package's documentation -- not checked against the package (absent here)."""
    probs, labels = np.asarray(probs, dtype=np.float64).reshape(-1), np.asarray(labels).reshape(-1)
    if probs.shape != labels.shape or probs.size == 0:
        raise ValueError("probs and labels: equally many, at least one")
    if not np.issubdtype(labels.dtype, np.integer) or labels.min() < 0 or labels.max() > 1:
        raise ValueError("labels: integers in {0, 1}")
    distinct = np.unique(probs)
    if distinct.size < probs.size / 4.0:                               # discrete scores: one bin per value
        edges = np.concatenate([(distinct[:-1] + distinct[1:]) / 2.0, [1.0]])
    else:
        parts = np.array_split(np.sort(probs), min(num_bins, probs.size))
        edges = [(parts[i][-1] + parts[i + 1][0]) / 2.0 for i in range(len(parts) - 1)] + [1.0]
        edges = np.array(sorted(set(edges)))
    which = np.searchsorted(edges, probs)                              # bin i: (edge[i-1], edge[i]]
    total = 0.0
    for b in range(len(edges)):
        sel = which == b
        n = int(sel.sum())
        if n < 2:
            continue
        mean_label = float(labels[sel].mean())
        err = (mean_label - float(probs[sel].mean())) ** 2 - mean_label * (1.0 - mean_label) / (n - 1.0)
        total += n / probs.size * err
    return float(max(total, 0.0) ** 0.5)


def default_marginal_fn():
    """The reference's own function when its dependency is installed (`calibration.get_calibration_error`, requirements.txt:16:
    `uncertainty-calibration`; CE:117-136 calls it with the package's defaults), else the restatement above -- and which one it is, for the
    table: the restatement is unpinned against the package (absent from this image), so a number produced with it says so."""
    try:
        import calibration as cal                                     # noqa: F401  (third-party: not in this image)
        return (lambda probs, labels: float(cal.get_calibration_error(np.asarray(probs), np.asarray(labels)))), "calibration.get_calibration_error"
    except Exception:
        return marginal_calibration_error, "restated (package `uncertainty-calibration` not installed: unpinned)"


def _min_uncertainty_error(entropy: torch.Tensor, is_tp: torch.Tensor) -> torch.Tensor:
    """CE:160-178 / CE:279-292: shuffle (ties), sort by entropy, 0.5 * (TP above the threshold / TP) + 0.5 * (non-TP below it / non-TP), min."""
    perm = torch.randperm(entropy.shape[0])
    entropy, is_tp = entropy[perm], is_tp[perm]
    _, order = entropy.sort()
    tp = is_tp[order]
    fp = 1.0 - tp
    errs = 0.5 * (tp.sum(0) - torch.cumsum(tp, 0)) / tp.sum(0) + 0.5 * torch.cumsum(fp, 0) / fp.sum(0)
    return errs.min() if errs.numel() else torch.tensor(float("nan"), dtype=torch.float64)


def calibration_errors(matched: dict, cat_mapping_dict: Dict[int, int], marginal_fn=marginal_calibration_error) -> dict:
    """CE:86-297 on the partitions `match_predictions_to_groundtruth` returns (true_positives, duplicates, false_positives; tensors on any
    device, moved to the CPU here).  Draws two `torch.randperm` per class, in the reference's order (seed torch to reproduce its ties)."""
    parts = {k: {n: (t.detach().cpu() if torch.is_tensor(t) else t) for n, t in v.items()} for k, v in matched.items()
             if k in ("true_positives", "duplicates", "false_positives")}
    for part in parts.values():                                                                       # CE:86-103
        if "gt_cat_idxs" in part:
            conv = torch.as_tensor([cat_mapping_dict[int(c)] for c in part["gt_cat_idxs"].reshape(-1).tolist()], dtype=torch.int64)
            part["gt_converted_cat_idxs"] = part["gt_cat_idxs"] = conv
        probs, idx = part["predicted_cls_probs"][:, :-1].max(1)
        part["predicted_cat_idxs"], part["output_logits"] = idx, probs
    tp, dup, fp = parts["true_positives"], parts["duplicates"], parts["false_positives"]
    k1 = tp["predicted_cls_probs"].shape[1]
    scores = torch.cat((tp["predicted_cls_probs"].flatten(), dup["predicted_cls_probs"].flatten(), fp["predicted_cls_probs"].flatten()), 0)   # CE:117-131
    onehot = torch.cat((torch.nn.functional.one_hot(tp["gt_cat_idxs"], k1).flatten(), torch.nn.functional.one_hot(dup["gt_cat_idxs"], k1).flatten(),
                        torch.zeros(fp["predicted_cls_probs"].numel(), dtype=torch.int64)), 0)
    out = {"cls_marginal_calibration_error": float(marginal_fn(scores.numpy(), onehot.numpy())),
           "cls_marginal_inputs": (scores.numpy(), onehot.numpy())}
    cls_min_u, reg_min_u, reg_ece, reg_mce = [], [], [], []
    for class_idx in cat_mapping_dict.values():                                                       # CE:138-292
        tpv, dv, fv = tp["gt_converted_cat_idxs"] == class_idx, dup["gt_converted_cat_idxs"] == class_idx, fp["predicted_cat_idxs"] == class_idx
        is_tp = torch.cat((torch.ones(int(tpv.sum())), torch.zeros(int(dv.sum())), torch.zeros(int(fv.sum()))), 0).double()
        cls_entropy = -torch.log(torch.cat((tp["output_logits"][tpv], dup["output_logits"][dv], fp["output_logits"][fv]), 0))
        cls_min_u.append(_min_uncertainty_error(cls_entropy, is_tp).double())
        means = torch.cat((tp["predicted_box_means"][tpv], dup["predicted_box_means"][dv]), 0)        # CE:180-261: matched detections only
        var = torch.diagonal(torch.cat((tp["predicted_box_covariances"][tpv], dup["predicted_box_covariances"][dv]), 0), dim1=1, dim2=2)
        gt = torch.cat((tp["gt_box_means"][tpv], dup["gt_box_means"][dv]), 0)
        ece_c, mce_c = [], []
        step = 1 / 15.0
        for d in range(gt.shape[1]):
            cdf = torch.distributions.Normal(means[:, d], scale=torch.sqrt(var[:, d])).cdf(gt[:, d])
            errs = []
            for i in torch.arange(0.0, 1.0 - step, step):
                frac = (cdf < (i + step)).float().sum() / cdf.shape[0]
                errs.append((frac - (i + step)) ** 2)
            errs = torch.stack(errs)
            mce_c.append(errs.max())
            ece_c.append(errs.mean())
        reg_mce.append(torch.stack(mce_c))
        reg_ece.append(torch.stack(ece_c))
        covs = torch.cat((tp["predicted_box_covariances"][tpv], dup["predicted_box_covariances"][dv], fp["predicted_box_covariances"][fv]), 0)   # CE:263-292
        ent = torch.distributions.multivariate_normal.MultivariateNormal(torch.zeros(covs.shape[0:2]), covs + 1e-4 * torch.eye(covs.shape[2])).entropy()
        reg_min_u.append(_min_uncertainty_error(ent, is_tp).double())

    def nanmean(ts):
        t = torch.stack(ts, 0).double()
        return float(t[~torch.isnan(t)].mean())
    out.update({"reg_expected_calibration_error": nanmean(reg_ece), "reg_maximum_calibration_error": nanmean(reg_mce),
                "cls_minimum_uncertainty_error": nanmean(cls_min_u), "reg_minimum_uncertainty_error": nanmean(reg_min_u)})
    return out


def format_table(res: dict) -> str:
    """CE:294-318 without prettytable."""
    names = ("Cls Marginal Calibration Error", "Reg Expected Calibration Error", "Reg Maximum Calibration Error", "Cls Minimum Uncertainty Error",
             "Reg Minimum Uncertainty Error")
    vals = ["%.4f" % res[k] for k in ("cls_marginal_calibration_error", "reg_expected_calibration_error", "reg_maximum_calibration_error",
                                      "cls_minimum_uncertainty_error", "reg_minimum_uncertainty_error")]
    w = [max(len(a), len(b)) for a, b in zip(names, vals)]
    line = "+" + "+".join("-" * (x + 2) for x in w) + "+"
    row = lambda r: "| " + " | ".join(str(r[i]).center(w[i]) for i in range(5)) + " |"
    return "\n".join([line, row(names), line, row(vals), line])


def main(argv=None):
    from .compute_probabilistic_metrics import BDD_DATASET_ID_TO_CONTIGUOUS
    from . import evaluation_utils as ev
    ap = argparse.ArgumentParser()
    ap.add_argument("--results", required=True, help="coco_instances_results.json written by apply_net (AN:100-102)")
    ap.add_argument("--gt", required=True, help="COCO-format ground truth json (its `annotations`)")
    ap.add_argument("--iou-min", type=float, default=0.1)
    ap.add_argument("--iou-correct", type=float, default=0.7)
    ap.add_argument("--min-allowed-score", type=float, default=0.0)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    with open(args.results, "r") as f:
        predicted = json.load(f)
    with open(args.gt, "r") as f:
        gt = json.load(f)["annotations"]
    pred = ev.eval_predictions_preprocess(predicted, args.min_allowed_score, device=args.device)
    g = ev.eval_gt_preprocess(gt, device=args.device)
    matched = ev.match_predictions_to_groundtruth(pred["predicted_boxes"], pred["predicted_cls_probs"], pred["predicted_covar_mats"],
                                                  g["gt_boxes"], g["gt_cat_idxs"], args.iou_min, args.iou_correct, device=args.device)
    fn, which = default_marginal_fn()
    res = calibration_errors(matched, BDD_DATASET_ID_TO_CONTIGUOUS, marginal_fn=fn)
    print(format_table(res))
    print("Cls Marginal Calibration Error computed by: " + which)       # (say it in the table's output, not only in a docstring)
    res["cls_marginal_calibration_error_source"] = which
    return res


if __name__ == "__main__":
    main()
