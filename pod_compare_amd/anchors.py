"""RetinaNet anchor grid (detectron2 DefaultAnchorGenerator semantics, restated).

Reference call site: probabilistic_retinanet.py:101 (`self.anchor_generator(features)`),
config Base-RetinaNet.yaml:8 (sizes [[x, x*2^(1/3), x*2^(2/3)] for x in 32..512],
hard-coded here -- the YAML's `!!python/object/apply:eval` is never evaluated),
aspect ratios (0.5, 1, 2), offset 0, strides 8..128.

Anchor order inside a level matches `permute_to_N_HWA_K`: index r = (h*W + w)*A + a,
with a = size-major x ratio.
"""
import math
from typing import List, Sequence, Tuple

import torch

ANCHOR_SIZES: Tuple[Tuple[float, float, float], ...] = tuple(
    (float(x), x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)) for x in (32, 64, 128, 256, 512))
ASPECT_RATIOS: Tuple[float, ...] = (0.5, 1.0, 2.0)
FPN_STRIDES: Tuple[int, ...] = (8, 16, 32, 64, 128)
SIZE_DIVISIBILITY = 32


def cell_anchors(sizes: Sequence[float], aspect_ratios: Sequence[float] = ASPECT_RATIOS) -> torch.Tensor:
    """(A,4) XYXY anchors centred at the origin; float64 maths, stored fp32."""
    rows = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(rows, dtype=torch.float32)


def level_shapes(height: int, width: int, strides: Sequence[int] = FPN_STRIDES) -> List[Tuple[int, int]]:
    """Feature-map (H_l, W_l) of p3..p7 for a padded input of height x width.

    p3..p5 come from stride-2 3x3 convs/pools with padding 1 (ceil division), p6/p7
    from stride-2 3x3 convs on the previous level (also ceil division)."""
    shapes = []
    for s in strides:
        shapes.append((-(-height // s), -(-width // s)))
    return shapes


def padded_size(height: int, width: int, divisibility: int = SIZE_DIVISIBILITY) -> Tuple[int, int]:
    return (-(-height // divisibility) * divisibility, -(-width // divisibility) * divisibility)


def grid_anchors(shapes: Sequence[Tuple[int, int]], strides: Sequence[int] = FPN_STRIDES,
                 sizes=ANCHOR_SIZES, device="cpu") -> List[torch.Tensor]:
    """Per-level (H*W*A, 4) fp32 anchor tensors in (h, w, a) order."""
    out = []
    for (h, w), stride, sz in zip(shapes, strides, sizes):
        base = cell_anchors(sz).to(device)
        sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
        sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx = xx.reshape(-1)
        yy = yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
    return out


def resize_shortest_edge(height: int, width: int, min_size: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    """detectron2 ResizeShortestEdge output size (test transform, apply_net.py:83)."""
    scale = min_size * 1.0 / min(height, width)
    if height < width:
        newh, neww = min_size, scale * width
    else:
        newh, neww = scale * height, min_size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)
