"""The three names probabilistic_retinanet.py takes from detectron2's RetinaNet (public behaviour, restated for the stub).

* permute_to_N_HWA_K: (N, A*K, H, W) -> (N, H*W*A, K): channel a*K + k of cell (h, w) becomes row (h*W + w)*A + a, column k.
* RetinaNetHead: the reference's ProbabilisticRetinaNetHead re-creates every layer itself after `super().__init__`
  (PR:403-484), so the base class only needs to be an nn.Module.
* RetinaNet: the attributes PR:20-108 reads -- num_classes, head_in_features, backbone (+ output_shape()), anchor_generator,
  device, preprocess_image.  The backbone is whatever `cfg.STUB_BACKBONE` holds: the fixtures feed FPN features directly, the
  ResNet-FPN is not part of the row this pins (a1: the head and the MC branch).
"""
import torch
from torch import nn

from detectron2.modeling.anchor_generator import build_anchor_generator


def permute_to_N_HWA_K(tensor, K):
    assert tensor.dim() == 4, tensor.shape
    N, _, H, W = tensor.shape
    tensor = tensor.view(N, -1, K, H, W)
    tensor = tensor.permute(0, 3, 4, 1, 2)
    return tensor.reshape(N, -1, K)


class RetinaNetHead(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()


class _Images:
    def __init__(self, tensor):
        self.tensor = tensor


class RetinaNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_classes = cfg.MODEL.RETINANET.NUM_CLASSES
        self.head_in_features = self.in_features = list(cfg.MODEL.RETINANET.IN_FEATURES)
        self.backbone = cfg.STUB_BACKBONE
        shapes = self.backbone.output_shape()
        self.anchor_generator = build_anchor_generator(cfg, [shapes[f] for f in self.head_in_features])
        self.register_buffer("pixel_mean", torch.zeros(3, 1, 1), persistent=False)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        return _Images(torch.stack([x["image"].float() for x in batched_inputs]))
