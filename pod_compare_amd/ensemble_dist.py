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
  * the merge rank rotates with the image index so consecutive images pipeline across GPUs;
  * `MemberPipeline` keeps two images in flight: image i's rows travel (on the collective library's own stream) while the
    member ranks already run the conv net on image i+1, and the merge rank only runs K1..K7 of image i after it has
    enqueued its own forward of image i+1 (`exchange_members` is the one-image-at-a-time form of the same exchange).

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


class MemberPipeline:
    """Double-buffered, software-pipelined form of `exchange_members` (replaces the sequential member loop PI:495-505).

    Every rank calls `post(i, member_outputs)` for every image i, in image order ("round" i):
      member rank s != dst(i): packs its head tensors into send slot i % depth and posts one `isend` to dst(i);
      dst(i): posts one `irecv` per other member into row s of `stacked[i % depth]` (its own member row is a local copy);
      other ranks: nothing.
    A round's operations are issued as ONE group (`batch_isend_irecv`: one fused RCCL kernel on the library's stream, so
    sends and receives of a round cannot block each other) and all ranks issue rounds in the same order, which makes the
    schedule deadlock-free.  Nothing here waits on the host with RCCL: `wait()` only orders streams.  `collect(i)` on
    dst(i) returns the `(M, packed)` buffer of image i once its rows have landed; call it after the NEXT image's forward
    has been enqueued (see `run`), then the transfer of image i overlaps that forward.  A slot is reused `depth` rounds
    later, after the work that last touched it has been waited for."""

    def __init__(self, layout: MemberLayout, n_members: int, rank: int, world: int, device, depth: int = 2):
        assert depth >= 2 and world >= 1 and n_members >= 1
        self.layout, self.M, self.rank, self.world, self.depth = layout, int(n_members), int(rank), int(world), int(depth)
        self.device = torch.device(device)
        self.packed = [torch.empty(layout.total, dtype=torch.float32, device=self.device) for _ in range(depth)] if rank < n_members else None
        self.stacked: List[Optional[torch.Tensor]] = [None] * depth        # allocated on first use as a merge rank
        self._send: List[list] = [[] for _ in range(depth)]
        self._recv = {}
        # RCCL moves device buffers directly (xGMI).  Any other backend (gloo: the functional check of this very code path
        # with all ranks on ONE GPU, and the CPU tests) cannot send device memory: rows are staged through pinned host
        # buffers -- same schedule, same buffers on the device side.
        self.host_staged = self.device.type == "cuda" and dist.get_backend() != "nccl"
        if self.host_staged:
            pin = lambda *shape: torch.empty(shape, dtype=torch.float32).pin_memory()
            self._h_send = [pin(layout.total) for _ in range(depth)] if rank < n_members else None
            self._h_recv: List[Optional[torch.Tensor]] = [None] * depth

    def post(self, image_index: int, member_outputs: Optional[HeadOutputs]) -> None:
        dst, slot = merge_rank(image_index, self.world), image_index % self.depth
        ops = []
        if self.rank < self.M:
            for w in self._send[slot]:
                w.wait()                                   # the send that last used this slot (image_index - depth)
            self._send[slot] = []
            self.layout.pack(member_outputs, out=self.packed[slot])
        if self.rank == dst:
            if self.stacked[slot] is None:
                self.stacked[slot] = torch.empty((self.M, self.layout.total), dtype=torch.float32, device=self.device)
                if self.host_staged:
                    self._h_recv[slot] = torch.empty((self.M, self.layout.total), dtype=torch.float32).pin_memory()
            for s in range(self.M):
                if s == self.rank:
                    self.stacked[slot][s].copy_(self.packed[slot])
                else:
                    ops.append(dist.P2POp(dist.irecv, (self._h_recv if self.host_staged else self.stacked)[slot][s], s))
        elif self.rank < self.M:
            if self.host_staged:
                self._h_send[slot].copy_(self.packed[slot])          # synchronous: the row is on the host when isend starts
                ops.append(dist.P2POp(dist.isend, self._h_send[slot], dst))
            else:
                ops.append(dist.P2POp(dist.isend, self.packed[slot], dst))
        works = dist.batch_isend_irecv(ops) if ops else []
        if self.rank == dst:
            self._recv[image_index] = works
        elif self.rank < self.M:
            self._send[slot] = works

    def collect(self, image_index: int) -> torch.Tensor:
        """(M, packed) buffer of image `image_index` on its merge rank (valid until round image_index + depth is posted)."""
        for w in self._recv.pop(image_index):
            w.wait()
        slot = image_index % self.depth
        if self.host_staged:
            for s in range(self.M):
                if s != self.rank:
                    self.stacked[slot][s].copy_(self._h_recv[slot][s])      # blocking: the host row may be overwritten by a later round
        return self.stacked[slot]

    def drain(self) -> None:
        for works in self._send:
            for w in works:
                w.wait()
        self._send = [[] for _ in range(self.depth)]

    def run(self, num_images: int, forward, merge) -> None:
        """The pipelined loop: `forward(i)` -> this rank's member HeadOutputs of image i (member ranks only; called for
        every image), `merge(i, stacked)` -> K1..K7 of image i on its merge rank.  Image i is merged after image i+1's
        forward has been enqueued, so its rows travel underneath that forward."""
        pending = None
        for i in range(num_images):
            ho = forward(i) if self.rank < self.M else None
            self.post(i, ho)
            if pending is not None:
                merge(pending, self.collect(pending))
                pending = None
            if merge_rank(i, self.world) == self.rank:
                pending = i
        if pending is not None:
            merge(pending, self.collect(pending))
        self.drain()
