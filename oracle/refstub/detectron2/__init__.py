"""Minimal detectron2 API stand-in, written for this repo (NOT detectron2 source).

Purpose: let `oracle/make_golden.py` import the reference's pure-Python
probabilistic_inference modules in a container without detectron2 /
torchvision, so that golden vectors come from the reference's own code.
Only the calls the reference makes on the hot path are provided; their
semantics restate the public detectron2 (v0.3/0.4) / torchvision
behaviour:

  * pairwise_iou: inter / (a1 + a2 - inter) with an `inter > 0` guard, areas
    (x2-x1)*(y2-y1), no +1.
  * batched_nms: torchvision "coordinate trick" -- plain NMS on
    boxes + class * (max_coord + 1); suppress when IoU > thr (strict);
    candidates visited in stable descending-score order.
  * Box2BoxTransform.apply_deltas: weights, dw/dh clamp at log(1000/16).
  * Boxes.scale / clip / nonempty, Instances field container + indexing.

This package is test infrastructure used in the build container only; it is
never imported by the product path and never needed on the GPU box.
"""
