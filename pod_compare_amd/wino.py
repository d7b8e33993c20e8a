"""Host side of pod_wino_conv3x3 (csrc/k11_wino_conv.hip): the head subnets' 3x3 convolutions (probabilistic_retinanet.py:403-427)
over all FPN levels and all MC runs in one launch.  Channels-last activations, every (level, run) image in one
[pixel][C] buffer; `block_table` lists the 16x16-pixel output blocks of all images.  GPU only: there is no CPU path."""
from typing import List, Optional, Sequence, Tuple

import torch

from . import hip


def level_pixel_offsets(levels: Sequence[Tuple[int, int]], copies: int) -> List[int]:
    """First pixel of each level's (copies, H, W, C) slab in the shared buffer (level-major), plus the total."""
    offs, total = [], 0
    for h, w in levels:
        offs.append(total)
        total += copies * h * w
    return offs + [total]


_TABLES = {}


def block_table(levels: Sequence[Tuple[int, int]], copies: int, device) -> torch.Tensor:
    """int32 (n_blocks, 4) {first pixel of the image, H, W, block_row << 16 | block_col}: one record per 16x16 output block.
    Blocks of one image are adjacent (their input halos overlap: L2 reuse); big levels first."""
    key = (tuple(levels), copies, str(device))
    t = _TABLES.get(key)
    if t is None:
        offs = level_pixel_offsets(levels, copies)
        rows = []
        for (h, w), off in zip(levels, offs):
            by, bx = (h + 15) // 16, (w + 15) // 16
            n = torch.arange(copies, dtype=torch.int64).view(-1, 1, 1)
            y = torch.arange(by, dtype=torch.int64).view(1, -1, 1)
            x = torch.arange(bx, dtype=torch.int64).view(1, 1, -1)
            rec = torch.stack(torch.broadcast_tensors(off + n * h * w, torch.tensor(h), torch.tensor(w), (y << 16) | x), dim=-1)
            rows.append(rec.reshape(-1, 4))
        t = torch.cat(rows).to(torch.int32).to(device).contiguous()
        assert offs[-1] < 2 ** 31
        _TABLES[key] = t
    return t


class WinoConv:
    """One conv3x3(C -> K, stride 1, pad 1) with its filter transformed once.  K is padded to a multiple of 64 with zero
    filters (outputs written for the padded channels are zero + nothing: bias is padded with zeros too)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
        self.K, self.C = int(weight.shape[0]), int(weight.shape[1])
        self.Kpad = (self.K + 63) // 64 * 64
        if self.C % 8 or self.Kpad not in (64, 128, 256, 512):
            raise ValueError("pod_wino_conv3x3: C %% 8 == 0 and K <= 512 in steps of 64 required, got C=%d K=%d" % (self.C, self.K))
        lib = hip.load()
        self.U = torch.empty(16 * self.Kpad * self.C, dtype=torch.float32, device=weight.device)
        hip.check(lib.pod_wino_filter_transform(weight.detach().contiguous().data_ptr(), self.U.data_ptr(), self.K, self.C, hip.current_stream()),
                  "pod_wino_filter_transform")
        self.bias = None
        if bias is not None:
            self.bias = torch.zeros(self.Kpad, dtype=torch.float32, device=weight.device)
            self.bias[:self.K].copy_(bias.detach())
        self.version = (weight._version, None if bias is None else bias._version, weight.data_ptr())

    def __call__(self, src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, relu: bool = False, dropout_p: float = 0.0,
                 seed: int = 0, offset: int = 0) -> torch.Tensor:
        """src: (pixels, C), dst: (pixels, Kpad), both contiguous fp32 on the GPU."""
        assert src.is_contiguous() and dst.is_contiguous() and src.shape[-1] == self.C and dst.shape[-1] == self.Kpad
        assert src.shape[0] == dst.shape[0] and src.dtype == dst.dtype == torch.float32
        hip.check(hip.load().pod_wino_conv3x3(src.data_ptr(), dst.data_ptr(), self.U.data_ptr(), hip.ptr(self.bias), table.data_ptr(),
                                              table.shape[0], self.C, self.Kpad, 1 if relu else 0, float(dropout_p), seed, offset,
                                              hip.current_stream()), "pod_wino_conv3x3")
        return dst
