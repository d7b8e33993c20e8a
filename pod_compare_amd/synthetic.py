"""Seeded synthetic head outputs ("planted objects"), SURVEY.md section 8(d).

Random-init RetinaNet weights give p ~= 0.01 < 0.05 (probabilistic_retinanet.py:454-455
vs probabilistic_inference.py:304), i.e. zero detections, so parity tests and the
benchmark drive the hot path with dense head tensors that contain planted boxes.

All tensors are produced in the product's HBM layout: per level, fp32
`(N_runs, A*C, H, W)` (what a conv head writes, NCHW). `to_reference_layout`
gives the `(1, H*W*A, C)` view the reference gets from `permute_to_N_HWA_K`
(probabilistic_retinanet.py:343-349).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import anchors as _anchors

SCALE_CLAMP = math.log(1000.0 / 16)


@dataclass
class HeadOutputs:
    """Dense anchor-wise head outputs of one image.

    cls/cls_var:  list over levels of (N, A*K, H, W) fp32
    delta/reg_var: list over levels of (N, A*4, H, W) / (N, A*D, H, W) fp32
    anchors: list over levels of (H*W*A, 4) fp32
    """
    cls: List[torch.Tensor]
    delta: List[torch.Tensor]
    cls_var: Optional[List[torch.Tensor]]
    reg_var: Optional[List[torch.Tensor]]
    anchors: List[torch.Tensor]
    shapes: List[Tuple[int, int]]
    num_anchors: int
    num_classes: int
    image_size: Tuple[int, int]
    planted_boxes: Optional[torch.Tensor] = None
    planted_classes: Optional[torch.Tensor] = None
    last_run_valid: bool = True      # False: run N-1 of cls / cls_var / reg_var was not computed (modeling: skip_unused_last_run)

    @property
    def num_runs(self) -> int:
        return self.cls[0].shape[0]

    def to(self, device) -> "HeadOutputs":
        mv = lambda lst: None if lst is None else [t.to(device) for t in lst]
        return HeadOutputs(mv(self.cls), mv(self.delta), mv(self.cls_var), mv(self.reg_var), mv(self.anchors),
                           self.shapes, self.num_anchors, self.num_classes, self.image_size,
                           None if self.planted_boxes is None else self.planted_boxes.to(device),
                           None if self.planted_classes is None else self.planted_classes.to(device), self.last_run_valid)


def nchw_from_anchor_major(x: torch.Tensor, h: int, w: int, a: int) -> torch.Tensor:
    """(N, H*W*A, C) -> (N, A*C, H, W)."""
    n, _, c = x.shape
    return x.view(n, h, w, a, c).permute(0, 3, 4, 1, 2).reshape(n, a * c, h, w).contiguous()


def anchor_major_from_nchw(x: torch.Tensor, c: int) -> torch.Tensor:
    """(N, A*C, H, W) -> (N, H*W*A, C); same as detectron2 permute_to_N_HWA_K."""
    n, ac, h, w = x.shape
    return x.view(n, ac // c, c, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, c)


def to_reference_layout(ho: HeadOutputs, run: Optional[int] = None) -> Dict[str, object]:
    """Raw-output dict of the reference model (probabilistic_retinanet.py:352-361) for one run
    (or, with run=None, requires N == 1)."""
    if run is None:
        assert ho.num_runs == 1
        run = 0
    k = ho.num_classes
    sel = lambda lst, c: None if lst is None else [anchor_major_from_nchw(t[run:run + 1], c).contiguous() for t in lst]
    dcov = None if ho.reg_var is None else ho.reg_var[0].shape[1] // ho.num_anchors
    return {"anchors": ho.anchors, "box_cls": sel(ho.cls, k), "box_delta": sel(ho.delta, 4),
            "box_cls_var": sel(ho.cls_var, k), "box_reg_var": sel(ho.reg_var, dcov)}


class SeededNormals:
    """Seeded random source.  On the CPU it is numpy's Philox bit generator (plain C, no SIMD
    dispatch: the same stream on every host, which is what lets golden fixtures store only a
    seed); on a GPU it is the device's torch generator (benchmark inputs, no parity claim)."""

    def __init__(self, seed: int, device="cpu"):
        self.device = torch.device(device)
        if self.device.type == "cpu":
            self.np = np.random.Generator(np.random.Philox(int(seed)))
            self.g = None
        else:
            self.np = None
            self.g = torch.Generator(device=self.device)
            self.g.manual_seed(int(seed))

    def randn(self, *shape) -> torch.Tensor:
        if self.np is not None:
            return torch.from_numpy(self.np.standard_normal(shape, dtype=np.float32))
        return torch.randn(*shape, generator=self.g, device=self.device, dtype=torch.float32)

    def rand(self, *shape) -> torch.Tensor:
        if self.np is not None:
            return torch.from_numpy(self.np.random(shape, dtype=np.float32))
        return torch.rand(*shape, generator=self.g, device=self.device, dtype=torch.float32)

    def randint(self, high: int, n: int) -> torch.Tensor:
        if self.np is not None:
            return torch.from_numpy(self.np.integers(0, high, size=(n,), dtype=np.int64))
        return torch.randint(0, high, (n,), generator=self.g, device=self.device)

    def __call__(self, shape) -> torch.Tensor:
        """eps source signature used by the oracle / predictor replay mode."""
        return self.randn(*tuple(shape))


def _exp(x: torch.Tensor) -> torch.Tensor:
    """exp/log through float64 numpy on the CPU: torch's fp32 SIMD exp/log (Sleef) differ in the last
    bit between AVX2 and AVX-512 hosts, which would break seed-only golden fixtures."""
    if x.device.type != "cpu":
        return torch.exp(x)
    return torch.from_numpy(np.exp(x.double().numpy())).float()


def _log(x: torch.Tensor) -> torch.Tensor:
    if x.device.type != "cpu":
        return torch.log(x)
    return torch.from_numpy(np.log(x.double().numpy())).float()


def _iou_rows(boxes: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    wh = torch.min(boxes[:, None, 2:], anchors[:, 2:]) - torch.max(boxes[:, None, :2], anchors[:, :2])
    inter = wh.clamp_(min=0).prod(dim=2)
    a1 = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    a2 = (anchors[:, 2] - anchors[:, 0]) * (anchors[:, 3] - anchors[:, 1])
    return inter / (a1[:, None] + a2 - inter)


def _deltas(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    tx, ty = dst[:, 0] + 0.5 * tw, dst[:, 1] + 0.5 * th
    return torch.stack(((tx - sx) / sw, (ty - sy) / sh, _log(tw / sw), _log(th / sh)), dim=1)


def planted_head_outputs(image_size: Tuple[int, int], num_runs: int = 1, *, seed: int = 0, num_boxes: int = 24,
                         num_classes: int = 7, with_cls_var: bool = True, with_reg_var: bool = True,
                         cov_dims: int = 4, mode: str = "planted", run_noise: float = 0.02,
                         match_iou: float = 0.5, device="cpu", strides: Sequence[int] = _anchors.FPN_STRIDES,
                         sizes=_anchors.ANCHOR_SIZES) -> HeadOutputs:
    """Dense head tensors for one image.

    mode="planted": `num_boxes` boxes (log-uniform side 24..400 px clipped to the frame, uniform
      class); anchors with IoU >= match_iou to a planted box get logit[class] ~ N(2, 0.5) and
      delta = get_deltas(anchor, box) + N(0, 0.05); all other logits ~ N(-4.6, 0.3).
    mode="worst": logits ~ N(-1, 1.5), deltas ~ N(0, 0.3): every level fills its top-k and
      clusters are singletons (the survey's CPU probe distribution).
    cls_var ~ N(-4, 0.5), reg_var ~ N(-5, 0.5); every run adds N(0, run_noise) to all tensors.
    """
    rng = SeededNormals(seed, device)
    H, W = image_size
    shapes = _anchors.level_shapes(H, W, strides)
    anchor_list = _anchors.grid_anchors(shapes, strides, sizes, device=device)
    A = len(sizes[0]) * len(_anchors.ASPECT_RATIOS)
    K = num_classes
    randn, rand = rng.randn, rng.rand

    boxes = classes = None
    if mode == "planted" and num_boxes > 0:
        side_lo, side_hi = math.log(24.0), math.log(min(400.0, 0.9 * min(H, W)))
        bw = _exp(side_lo + (side_hi - side_lo) * rand(num_boxes))
        bh = _exp(side_lo + (side_hi - side_lo) * rand(num_boxes))
        bx = rand(num_boxes) * (W - bw)
        by = rand(num_boxes) * (H - bh)
        boxes = torch.stack((bx, by, bx + bw, by + bh), dim=1)
        classes = rng.randint(K, num_boxes)

    cls, delta, cls_var, reg_var = [], [], [], []
    for (h, w), anc in zip(shapes, anchor_list):
        r = anc.shape[0]
        if mode == "planted":
            logits = -4.6 + 0.3 * randn(r, K)
            dl = 0.05 * randn(r, 4)
            pidx = torch.zeros((0,), dtype=torch.int64)
            if boxes is not None:
                best_iou, best = _iou_rows(boxes, anc).max(dim=0)      # over the G planted boxes
                pidx = (best_iou >= match_iou).nonzero().squeeze(1)
            if pidx.numel() > 0:
                pb = best[pidx]
                logits[pidx, classes[pb]] = 2.0 + 0.5 * randn(pidx.numel())
                dl[pidx] = dl[pidx] + _deltas(anc[pidx], boxes[pb])
        elif mode == "worst":
            logits = -1.0 + 1.5 * randn(r, K)
            dl = 0.3 * randn(r, 4)
        else:
            raise ValueError("unknown synthetic mode {}".format(mode))
        cv = -4.0 + 0.5 * randn(r, K)
        rv = -5.0 + 0.5 * randn(r, cov_dims)
        if cov_dims > 4:
            rv[:, 4:] = 0.02 * randn(r, cov_dims - 4)

        def runs(base):
            x = base.unsqueeze(0).expand(num_runs, -1, -1)
            if num_runs > 1:
                x = x + run_noise * randn(num_runs, *base.shape)
            return nchw_from_anchor_major(x.contiguous(), h, w, A)

        cls.append(runs(logits))
        delta.append(runs(dl))
        cls_var.append(runs(cv))
        reg_var.append(runs(rv))
    return HeadOutputs(cls, delta, cls_var if with_cls_var else None, reg_var if with_reg_var else None,
                       anchor_list, shapes, A, K, (H, W), boxes, classes)


def synthetic_frame(image_id: int, height: int = 720, width: int = 1280, device="cpu") -> torch.Tensor:
    """uint8 BGR (3, H, W) frame, uniform noise, seed 1234 + image_id."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + int(image_id))
    return torch.randint(0, 256, (3, height, width), generator=g, dtype=torch.uint8).to(device)
