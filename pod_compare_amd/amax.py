"""Operand abs-max words of the split convolutions (include/pod_mi355x.h: "operand abs-max words").

The round-5 split kernels (pod_wino_conv3x3_split, pod_conv1x1_split, pod_stem7x7_split) form every fp32 product from two f16 terms of
the power-of-two-SCALED operands; the activation scale of a launch comes from a device word holding (an upper bound of) the abs-max of
its input.  Producers publish it in their store pass (`out_amax`), anything else gets it from pod_absmax.  The word travels with the
tensor OBJECT as an attribute -- never with a data pointer, which the allocator re-uses -- together with the tensor's version counter:
an in-place torch op on the tensor afterwards invalidates it (the pod_* kernels write through raw pointers and do not bump versions).
Words come from small zeroed pools, one per stream (the zero fill is ordered on the stream that later max'es into the word); a forward
captured into a HIP graph starts a fresh pool INSIDE the capture, so every replay re-zeroes its words."""
from typing import Dict, Optional

import torch

from . import hip

POOL_WORDS = 256
RECORD = 512          # include/pod_mi355x.h: POD_AMAX_FLOATS (16 slots, one 128-byte line each)
_POOLS: Dict[int, list] = {}


def reset() -> None:
    """Forget the current pools (the next word starts a new, zeroed one).  Called at the start and end of a graph capture."""
    _POOLS.clear()


def word(device) -> torch.Tensor:
    """A zeroed abs-max record (a RECORD-float view of the current stream's pool)."""
    key = torch.cuda.current_stream(device).cuda_stream
    ent = _POOLS.get(key)
    if ent is None or ent[1] >= POOL_WORDS:
        ent = _POOLS[key] = [torch.zeros(POOL_WORDS * RECORD, dtype=torch.float32, device=device), 0]
    w = ent[0][ent[1] * RECORD:(ent[1] + 1) * RECORD]
    ent[1] += 1
    return w


def attach(t: torch.Tensor, w: Optional[torch.Tensor]) -> torch.Tensor:
    """Records that `w` bounds |t| as it stands now."""
    if w is not None:
        t._pod_amax = (w, t._version)
    return t


def produced(t: torch.Tensor) -> torch.Tensor:
    """The record a launch about to write t max'es into (its out_amax), attached to t: a fresh zeroed one -- or, when t is one slice of a
    buffer several launches fill (FPN's five levels in one buffer), the record those launches share (`_pod_amax_shared`, itself taken fresh
    from the pool for every forward).  No record outlives an image: the scale of a split is a function of the image alone (round 6; round
    5's sparse tower kept a never-zeroed record per buffer, which made an image's roundings depend on the images before it)."""
    w = getattr(t, "_pod_amax_shared", None)
    if w is None:
        w = word(t.device)
    attach(t, w)
    return w


def forget(t: torch.Tensor) -> None:
    if hasattr(t, "_pod_amax"):
        del t._pod_amax


def of(t: torch.Tensor) -> torch.Tensor:
    """The abs-max word of t: the producer's if t still is what the producer wrote, else computed now (pod_absmax)."""
    rec = getattr(t, "_pod_amax", None)
    if rec is not None and rec[1] == t._version:
        return rec[0]
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    w = word(t.device)
    hip.check(hip.load().pod_absmax(t.data_ptr(), t.numel(), w.data_ptr(), hip.current_stream()), "pod_absmax")
    attach(t, w)
    return w


def joined(dst: torch.Tensor, *srcs: torch.Tensor) -> torch.Tensor:
    """dst holds values of the srcs (a concatenation, a copy, a broadcast): its bound is the largest of theirs (their words max'ed into
    a new one: a one-element pod_absmax each)."""
    w = word(dst.device)
    for s in srcs:
        hip.check(hip.load().pod_absmax(of(s).data_ptr(), RECORD, w.data_ptr(), hip.current_stream()), "pod_absmax")
    return attach(dst, w)
