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


if __name__ == "__main__":
    main()
