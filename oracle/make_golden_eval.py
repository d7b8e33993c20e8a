#!/usr/bin/env python3
"""Golden vectors for SURVEY row f-1 (ground-truth matching + scoring rules), produced by the REFERENCE's own
core/evaluation_tools/{evaluation_utils,scoring_rules}.py imported in this container.  Inputs are stored verbatim."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.refimport import load_reference_evaluation  # noqa: E402
from pod_compare_amd.synthetic import SeededNormals  # noqa: E402


def make_inputs(seed=77, n_images=9, k=7):
    rng = SeededNormals(seed)
    pb, pp, pc, gb, gc = {}, {}, {}, {}, {}
    for img in range(n_images):
        g = int(rng.randint(6, 1)[0]) + (0 if img % 4 == 3 else 1)           # some frames have no ground truth at all
        if img == 5:
            g = 0
        boxes = []
        for _ in range(g):
            x, y = float(rng.rand(1)) * 900, float(rng.rand(1)) * 500
            w, h = 40 + float(rng.rand(1)) * 200, 30 + float(rng.rand(1)) * 150
            boxes.append([x, y, x + w, y + h])
        gts = torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4)
        dets = []
        for b in gts:                                                          # 0..3 detections per object, mixed quality
            for _ in range(int(rng.randint(4, 1)[0])):
                jitter = (2.0, 12.0, 60.0)[int(rng.randint(3, 1)[0])]
                dets.append(b + jitter * rng.randn(4))
        for _ in range(int(rng.randint(4, 1)[0])):                             # clutter
            x, y = float(rng.rand(1)) * 1100, float(rng.rand(1)) * 600
            dets.append(torch.tensor([x, y, x + 50 + 80 * float(rng.rand(1)), y + 40 + 60 * float(rng.rand(1))]))
        if img == 7:
            dets = dets[:0]                                                    # frame with objects but no detection: never visited
        if len(dets):
            d = torch.stack(dets).float()
            probs = 0.02 + 0.1 * rng.rand(d.shape[0], k)
            probs[torch.arange(d.shape[0]), rng.randint(k, d.shape[0])] = 0.3 + 0.69 * rng.rand(d.shape[0])
            l = rng.randn(d.shape[0], 4, 4)
            pb[img], pp[img], pc[img] = d, probs, torch.matmul(l, l.transpose(1, 2)) + 4.0 * torch.eye(4)
        if g > 0:
            gb[img] = gts
            gc[img] = (rng.randint(k, g) + 1).float().reshape(-1, 1)
    return pb, pp, pc, gb, gc


def main():
    eu, sr = load_reference_evaluation()
    import io
    import contextlib
    pb, pp, pc, gb, gc = make_inputs()
    with contextlib.redirect_stderr(io.StringIO()):
        res = eu.match_predictions_to_groundtruth(pb, pp, pc, gb, gc, iou_min=0.1, iou_correct=0.7)
    out = {}
    for img in pb:
        out["in_pb_%d" % img], out["in_pp_%d" % img], out["in_pc_%d" % img] = pb[img].numpy(), pp[img].numpy(), pc[img].numpy()
    for img in gb:
        out["in_gb_%d" % img], out["in_gc_%d" % img] = gb[img].numpy(), gc[img].numpy()
    out["pred_keys"] = np.array(list(pb.keys()))
    out["gt_keys"] = np.array(list(gb.keys()))
    for part, d in res.items():
        for name, t in d.items():
            out["%s__%s" % (part, name)] = t.numpy()
    tp = res["true_positives"]
    valid = torch.ones(tp["predicted_box_means"].shape[0], dtype=torch.bool)
    reg = sr.compute_reg_scores(tp, valid)
    out["tp_ignorance"] = np.float64(reg["ignorance_score_mean"])
    out["tp_mse"] = np.float64(reg["mean_squared_error"])
    fp = dict(res["false_positives"])
    fpv = torch.ones(fp["predicted_box_means"].shape[0], dtype=torch.bool)
    out["fp_entropy"] = np.float64(sr.compute_reg_scores_fn(fp, fpv)["total_entropy_mean"])
    gt_idx = (tp["gt_cat_idxs"].squeeze(1) - 1).long()
    tp2 = dict(tp)
    tp2["predicted_score_of_gt_category"] = torch.gather(tp["predicted_cls_probs"], 1, gt_idx.unsqueeze(1)).squeeze(1)
    out["tp_cls_ignorance"] = np.float64(sr.retinanet_compute_cls_scores(tp2, valid)["ignorance_score_mean"])
    path = os.path.join(ROOT, "tests", "golden", "eval_matching.npz")
    np.savez_compressed(path, **out)
    print({k: v.shape for k, v in out.items() if "__" in k})
    print("ignorance %.6f mse %.4f fp entropy %.6f cls %.6f -> %s (%.1f KB)" % (out["tp_ignorance"], out["tp_mse"], out["fp_entropy"],
                                                                          out["tp_cls_ignorance"], path, os.path.getsize(path) / 1024))


def pm_composition(eu, sr, predicted_instances, gt_instances, cat_map, iou_min=0.1, iou_correct=0.7, min_allowed_score=0.0, classes=(1, 3)):
    """offline_evaluation/compute_probabilistic_metrics.py:81-178 as a composition of the REFERENCE's own functions (`eu`, `sr`
    are its evaluation_utils / scoring_rules modules): the script itself cannot be imported here (detectron2 MetadataCatalog /
    launch, prettytable, on-disk caches), its arithmetic can."""
    pred = eu.eval_predictions_preprocess(predicted_instances, min_allowed_score)
    gt = eu.eval_gt_preprocess(gt_instances)
    import contextlib
    import io
    with contextlib.redirect_stderr(io.StringIO()):
        matched = eu.match_predictions_to_groundtruth(pred["predicted_boxes"], pred["predicted_cls_probs"], pred["predicted_covar_mats"],
                                                      gt["gt_boxes"], gt["gt_cat_idxs"], iou_min, iou_correct)
    for part in matched.values():                                                        # PM:88-114
        if "gt_cat_idxs" in part:
            conv = torch.as_tensor([cat_map[c] for c in part["gt_cat_idxs"].squeeze(1).tolist()], dtype=torch.int64)
            part["gt_converted_cat_idxs"] = conv
            if "predicted_cls_probs" in part:
                part["predicted_score_of_gt_category"] = torch.gather(part["predicted_cls_probs"], 1, conv.unsqueeze(1)).squeeze(1)
            part["gt_cat_idxs"] = conv
        else:
            probs, idx = part["predicted_cls_probs"].max(1)
            part["predicted_score_of_gt_category"] = 1.0 - probs
            part["predicted_cat_idxs"] = idx
    tp, fp = matched["true_positives"], matched["false_positives"]
    per_class = []
    for class_idx in classes:                                                            # PM:123-146
        tv, fv = tp["gt_converted_cat_idxs"] == class_idx, fp["predicted_cat_idxs"] == class_idx
        per_class.append({"true_positives_cls_analysis": sr.retinanet_compute_cls_scores(tp, tv),
                          "true_positives_reg_analysis": sr.compute_reg_scores(tp, tv),
                          "false_positives_cls_analysis": sr.retinanet_compute_cls_scores(fp, fv),
                          "false_positives_reg_analysis": sr.compute_reg_scores_fn(fp, fv)})
    avg = {}
    for key in per_class[0]:                                                             # PM:148-178
        for inner in per_class[0][key]:
            vals = np.array([c[key][inner] for c in per_class if c[key][inner] is not None])
            avg[key + "/" + inner] = float(np.nanmean(vals))
    counts = [tp["predicted_box_means"].shape[0], matched["duplicates"]["predicted_box_means"].shape[0], fp["predicted_box_means"].shape[0],
              matched["false_negatives"]["gt_box_means"].shape[0]]
    return avg, counts, per_class


def main_metrics():
    """tests/golden/eval_metrics.npz: a result file + ground truth as JSON strings, and PM's numbers for them."""
    import json
    eu, sr = load_reference_evaluation()
    torch.Tensor.cuda = lambda self, *a, **k: self          # EU:88-91 calls .cuda() unconditionally; this container has no GPU
    pb, pp, pc, gb, gc = make_inputs(seed=91, n_images=14)
    t = torch.tensor([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [-1.0, 0, 1.0, 0], [0, -1.0, 0, 1.0]])
    predicted, gts = [], []
    for img in pb:
        for b, p, c in zip(pb[img], pp[img], pc[img]):
            xywh = [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]
            predicted.append({"image_id": int(img), "category_id": int(p.argmax()) + 1, "bbox": xywh, "score": float(p.max()),
                              "cls_prob": [float(x) for x in p], "bbox_covar": (t @ c @ t.t()).tolist()})
    for img in gb:
        for b, c in zip(gb[img], gc[img]):
            gts.append({"image_id": int(img), "category_id": int(c), "bbox": [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]})
    cat_map = {i + 1: i for i in range(7)}
    avg, counts, per_class = pm_composition(eu, sr, predicted, gts, cat_map)
    out = {"predicted_json": np.array(json.dumps(predicted)), "gt_json": np.array(json.dumps(gts)), "counts": np.array(counts),
           "avg_keys": np.array(sorted(avg)), "avg_vals": np.array([avg[k] for k in sorted(avg)], dtype=np.float64),
           "per_class_json": np.array(json.dumps(per_class))}
    path = os.path.join(ROOT, "tests", "golden", "eval_metrics.npz")
    np.savez_compressed(path, **out)
    print(counts, avg, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
    main_metrics()
