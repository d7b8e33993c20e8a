"""One-seed-per-GPU ensembles (BASELINE config 5: `ensembles_pre_nms.yaml`, 5 seeds, 8 GPUs of one node).

The reference runs the 5 member models one after the other in one process (PI:495-505).  Here rank s < M holds
member s (its own weights), every member rank runs the conv net on the SAME image, and the dense pre-NMS head
tensors (17 MB per member at BASELINE size) meet on a merge rank before K1:

  * each member packs its level tensors into ONE contiguous fp32 buffer (`MemberLayout`, 16-B aligned segments);
  * the merge rank owns a `(M, packed)` buffer; member ranks send their row with point-to-point `isend`/`irecv`
    (RCCL over xGMI: M distinct links into one GPU in parallel -- a ring collective would be per-link bound,
    SURVEY 5/8e); the merge rank's own member (if any) is a local copy;
  * `MemberLayout.views` exposes that buffer as per-level `(M, A*C, H, W)` tensors whose run stride is the packed
    size -- exactly what K1 streams (`PodLevel.run_stride_*`), so there is no re-layout after the exchange;
  * the merge rank rotates with the image index so consecutive images pipeline across GPUs.

The exchange logic is backend-agnostic ("nccl" on GPUs, "gloo" in the CPU tests).
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .synthetic import HeadOutputs


class MemberLayout:
    """Offsets of every (tensor, level) segment inside a member's packed buffer."""

    def __init__(self, shapes: Sequence[Tuple[int, int]], num_anchors: int, num_classes: int, cov_dims: int, has_cls_var: bool):
        self.shapes, self.A, self.K, self.D, self.has_cls_var = [tuple(s) for s in shapes], num_anchors, num_classes, cov_dims, has_cls_var
        self.names = ["cls", "delta"] + (["cls_var"] if has_cls_var else []) + (["reg_var"] if cov_dims > 0 else [])
        chan = {"cls": num_classes, "cls_var": num_classes, "delta": 4, "reg_var": cov_dims}
        self.offsets, off = {}, 0
        for name in self.names:
            for l, (h, w) in enumerate(self.shapes):
                n = num_anchors * chan[name] * h * w
                self.offsets[(name, l)] = (off, num_anchors * chan[name], h, w)
                off += -(-n // 4) * 4           # keep every segment 16-byte aligned
        self.total = off

    @classmethod
    def of(cls, ho: HeadOutputs) -> "MemberLayout":
        d = 0 if ho.reg_var is None else ho.reg_var[0].shape[1] // ho.num_anchors
        return cls(ho.shapes, ho.num_anchors, ho.num_classes, d, ho.cls_var is not None)

    def pack(self, ho: HeadOutputs, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Member (N = 1) head tensors -> one flat fp32 buffer."""
        assert ho.num_runs == 1
        dev = ho.cls[0].device
        out = torch.zeros(self.total, dtype=torch.float32, device=dev) if out is None else out
        for name in self.names:
            for l, t in enumerate(getattr(ho, name)):
                off, c, h, w = self.offsets[(name, l)]
                out[off:off + c * h * w].copy_(t.reshape(-1))
        return out

    def views(self, stacked: torch.Tensor, like: HeadOutputs) -> HeadOutputs:
        """(M, total) buffer -> HeadOutputs with runs = members (strided views, no copy)."""
        m = stacked.shape[0]
        assert stacked.shape[1] == self.total and stacked.is_contiguous()

        def lvl(name):
            if name not in self.names:
                return None
            out = []
            for l in range(len(self.shapes)):
                off, c, h, w = self.offsets[(name, l)]
                out.append(stacked.as_strided((m, c, h, w), (self.total, h * w, w, 1), stacked.storage_offset() + off))
            return out

        return HeadOutputs(lvl("cls"), lvl("delta"), lvl("cls_var"), lvl("reg_var"), like.anchors, like.shapes, like.num_anchors,
                           like.num_classes, like.image_size)


def merge_rank(image_index: int, world: int) -> int:
    """The merge rank rotates with the image so consecutive images overlap on different GPUs."""
    return image_index % world


def exchange_members(packed: Optional[torch.Tensor], stacked: Optional[torch.Tensor], n_members: int, dst: int,
                     rank: int) -> None:
    """Point-to-point gather of the members' packed buffers onto rank `dst` (row s of `stacked` <- member s).
    Ranks >= n_members that are not `dst` do nothing.  Blocks until this rank's transfers completed."""
    reqs = []
    if rank == dst:
        for s in range(n_members):
            if s == rank:
                stacked[s].copy_(packed)
            else:
                reqs.append(dist.irecv(stacked[s], src=s))
    elif rank < n_members:
        reqs.append(dist.isend(packed, dst=dst))
    for r in reqs:
        r.wait()
