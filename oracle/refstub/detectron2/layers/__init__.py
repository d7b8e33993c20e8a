import torch


def cat(tensors, dim=0):
    """detectron2.layers.cat: torch.cat that skips the copy for a single tensor."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def _nms_sorted_greedy(boxes, scores, thr):
    """Greedy NMS, torchvision CPU-kernel arithmetic: visit in stable
    descending score order; area=(x2-x1)*(y2-y1); ovr=inter/(ai+aj-inter);
    suppress iff ovr > thr. Returns kept indices in visit order (int64)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order].float()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = torch.maximum(x1[i], x1[i + 1:])
        yy1 = torch.maximum(y1[i], y1[i + 1:])
        xx2 = torch.minimum(x2[i], x2[i + 1:])
        yy2 = torch.minimum(y2[i], y2[i + 1:])
        zero = torch.zeros((), dtype=b.dtype)
        w = torch.maximum(zero, xx2 - xx1)
        h = torch.maximum(zero, yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return order[torch.as_tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Class-aware NMS via the coordinate-offset trick (torchvision.ops.batched_nms)."""
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    boxes_for_nms = boxes + offsets[:, None]
    return _nms_sorted_greedy(boxes_for_nms, scores, iou_threshold)


class ShapeSpec:
    """detectron2.layers.ShapeSpec: the (channels, height, width, stride) note a backbone leaves for the heads."""

    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride
