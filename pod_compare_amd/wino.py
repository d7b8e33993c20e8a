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


def block_table(levels: Sequence[Tuple[int, int]], copies: int, device, in_copies: Optional[int] = None, in_first: int = 0,
                out_copies: Optional[int] = None) -> torch.Tensor:
    """int32 (n_blocks, 4) records of pod_wino_conv3x3: {first pixel of image 0 in the input buffer, in the output buffer,
    H << 16 | W, n_images << 24 | block_row << 12 | block_col}, one per 16x16-pixel block of a CANVAS: the `copies` images of a level
    stand side by side, image n at canvas columns n*Wv .. n*Wv+W-1 with Wv = W + 1 rounded up to a multiple of 4 (the spare columns are
    the zero padding between neighbours), so a partial block at the right edge is paid once per level, not once per image -- where that
    saves blocks; otherwise every image is its own canvas.  The input
    buffer holds `in_copies` images per level (level-major) of which images in_first .. in_first + copies - 1 are read; the output
    buffer holds `out_copies` per level and images 0 .. copies - 1 are written."""
    in_copies = copies if in_copies is None else in_copies
    out_copies = copies if out_copies is None else out_copies
    assert in_first + copies <= in_copies and copies <= out_copies
    key = (tuple(levels), copies, in_copies, in_first, out_copies, str(device))
    t = _TABLES.get(key)
    if t is None:
        ioffs, ooffs = level_pixel_offsets(levels, in_copies), level_pixel_offsets(levels, out_copies)
        rows = []
        for (h, w), ioff, ooff in zip(levels, ioffs, ooffs):
            assert h < 65536 and w < 65536
            wv = (w + 4) // 4 * 4
            # side by side only where that needs fewer blocks (a width that is a multiple of 16 is better off one image per canvas)
            group = 127 if ((copies - 1) * wv + w + 15) // 16 < copies * ((w + 15) // 16) else 1
            group = max(1, min(group, (2 ** 31 - 1) // (h * w * 512 * 4), 65535 // wv))   # 32-bit byte offsets (C <= 512), 12-bit block columns
            done = 0
            while done < copies:                            # canvases of at most 127 images (the count sits in the top byte of an int32)
                n = min(group, copies - done)
                by, bx = (h + 15) // 16, ((n - 1) * wv + w + 15) // 16
                assert by < 4096 and bx < 4096
                y = torch.arange(by, dtype=torch.int64).view(-1, 1)
                x = torch.arange(bx, dtype=torch.int64).view(1, -1)
                rec = torch.stack(torch.broadcast_tensors(torch.tensor(ioff + (in_first + done) * h * w), torch.tensor(ooff + done * h * w),
                                                          torch.tensor((h << 16) | w), (n << 24) | (y << 12) | x), dim=-1)
                rows.append(rec.reshape(-1, 4))
                done += n
        assert max(ioffs[-1], ooffs[-1]) < 2 ** 31
        t = torch.cat(rows).to(torch.int32).to(device).contiguous()
        t.pod_pixels = copies * sum(h * w for h, w in levels)          # output pixels of a launch with this table
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
        self.U = torch.empty(24 * self.Kpad * self.C, dtype=torch.float32, device=weight.device)
        hip.check(lib.pod_wino_filter_transform(weight.detach().contiguous().data_ptr(), self.U.data_ptr(), self.K, self.C, hip.current_stream()),
                  "pod_wino_filter_transform")
        self.bias = None
        if bias is not None:
            self.bias = torch.zeros(self.Kpad, dtype=torch.float32, device=weight.device)
            self.bias[:self.K].copy_(bias.detach())
        self.version = (weight._version, None if bias is None else bias._version, weight.data_ptr())

    def __call__(self, src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, relu: bool = False, dropout_p: float = 0.0,
                 seed: int = 0, offset: int = 0, planes: bool = False) -> torch.Tensor:
        """src: (pixels, C) channels-last.  dst: (pixels, Kpad) channels-last, or with planes=True any contiguous buffer of
        NCHW images with K (real) planes each, level-major like the table's output side."""
        assert src.is_contiguous() and dst.is_contiguous() and src.shape[-1] == self.C and src.dtype == dst.dtype == torch.float32
        assert planes or dst.shape[-1] == self.Kpad
        hip.check(hip.load().pod_wino_conv3x3(src.data_ptr(), dst.data_ptr(), self.U.data_ptr(), hip.ptr(self.bias), table.data_ptr(),
                                              table.shape[0], self.C, self.Kpad, self.K if planes else 0, 1 if relu else 0, float(dropout_p),
                                              seed, offset, hip.current_stream()), "pod_wino_conv3x3")
        return dst
