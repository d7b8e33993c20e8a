"""detectron2 DefaultAnchorGenerator semantics (public behaviour, restated for the stub): per level A = |sizes| x |ratios| cell
anchors (size-major, w = sqrt(area / ratio), h = ratio * w, centred on 0), shifted over the grid with the level's stride and
`offset`; a level's anchors come out in (h, w, a) order -- the order permute_to_N_HWA_K gives the head outputs."""
import math

import torch

from detectron2.structures import Boxes


class DefaultAnchorGenerator:
    def __init__(self, sizes, aspect_ratios, strides, offset=0.0):
        self.strides, self.offset = list(strides), float(offset)
        n = len(self.strides)
        sizes = list(sizes) * n if len(sizes) == 1 else list(sizes)
        aspect_ratios = list(aspect_ratios) * n if len(aspect_ratios) == 1 else list(aspect_ratios)
        assert len(sizes) == n and len(aspect_ratios) == n
        self.cell_anchors = [self._cell(s, a) for s, a in zip(sizes, aspect_ratios)]

    @staticmethod
    def _cell(sizes, ratios):
        rows = []
        for size in sizes:
            area = size ** 2.0
            for r in ratios:
                w = math.sqrt(area / r)
                h = r * w
                rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(rows)

    @property
    def num_cell_anchors(self):
        return [len(c) for c in self.cell_anchors]

    def __call__(self, features):
        out = []
        for f, stride, base in zip(features, self.strides, self.cell_anchors):
            h, w = f.shape[-2:]
            sx = torch.arange(self.offset * stride, w * stride, step=stride, dtype=torch.float32)
            sy = torch.arange(self.offset * stride, h * stride, step=stride, dtype=torch.float32)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx, yy = xx.reshape(-1), yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            out.append(Boxes((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)))
        return out


def build_anchor_generator(cfg, input_shape):
    a = cfg.MODEL.ANCHOR_GENERATOR
    return DefaultAnchorGenerator(a.SIZES, a.ASPECT_RATIOS, [s.stride for s in input_shape], a.OFFSET)
