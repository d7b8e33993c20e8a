"""Host-side helpers with the reference's names (inference_utils.py, IU).

The numeric post-processing of IU (NMS, clustering, Bayesian fusion, sample moments, rescale) lives in the HIP
library; what remains on the host is the result format: `instances_to_json` (IU:454-502) and
`covar_xyxy_to_xywh` (IU:428-451), plus the fixed-stride record form K7 emits for the multi-GPU gather.
"""
from typing import Dict, List, Optional

import torch

RECORD_HEAD = 6   # x, y, w, h, score, class


def record_width(num_classes: int) -> int:
    return RECORD_HEAD + num_classes + 16


def covar_xyxy_to_xywh(output_boxes_covariance: torch.Tensor) -> torch.Tensor:
    """IU:428-451: T cov T^T, T = [[1,0,0,0],[0,1,0,0],[-1,0,1,0],[0,-1,0,1]] (on the tensor's device)."""
    t = torch.as_tensor([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [-1.0, 0, 1.0, 0], [0, -1.0, 0, 1.0]],
                        device=output_boxes_covariance.device, dtype=output_boxes_covariance.dtype)
    return t @ output_boxes_covariance @ t.t()


def instances_to_json(instances, img_id, cat_mapping_dict: Optional[Dict[int, int]] = None) -> List[dict]:
    """IU:454-502: COCO-style dicts with keys image_id, category_id, bbox (XYWH), score, cls_prob, bbox_covar."""
    num_instance = len(instances)
    if num_instance == 0:
        return []
    boxes = instances.pred_boxes.tensor.detach().cpu().clone()
    boxes[:, 2] -= boxes[:, 0]
    boxes[:, 3] -= boxes[:, 1]
    boxes = boxes.tolist()
    scores = instances.scores.cpu().tolist()
    classes = instances.pred_classes.cpu().tolist()
    classes = [cat_mapping_dict[c] if c in cat_mapping_dict.keys() else -1 for c in classes]
    pred_cls_probs = instances.pred_cls_probs.cpu().tolist()
    covs = covar_xyxy_to_xywh(instances.pred_boxes_covariance).cpu().tolist() if instances.has("pred_boxes_covariance") else []
    return [{"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k], "cls_prob": pred_cls_probs[k],
             "bbox_covar": covs[k]} for k in range(num_instance) if classes[k] != -1]


def records_to_json(records: torch.Tensor, count: int, img_id, num_classes: int,
                    cat_mapping_dict: Optional[Dict[int, int]] = None) -> List[dict]:
    """Same output as `instances_to_json`, from K7's records (XYWH box and T cov T^T already applied on the GPU)."""
    rec = records[:count].detach().cpu()
    out = []
    for row in rec.tolist():
        c = int(row[5])
        cid = cat_mapping_dict[c] if cat_mapping_dict is not None and c in cat_mapping_dict else -1
        if cid == -1:
            continue
        cov = row[RECORD_HEAD + num_classes:]
        out.append({"image_id": img_id, "category_id": cid, "bbox": row[0:4], "score": row[4],
                    "cls_prob": row[RECORD_HEAD:RECORD_HEAD + num_classes], "bbox_covar": [cov[0:4], cov[4:8], cov[8:12], cov[12:16]]})
    return out
