"""Host side of pod_wino_conv3x3 (csrc/k11_wino_conv.hip): the head subnets' 3x3 convolutions (probabilistic_retinanet.py:403-427)
over all FPN levels and all MC runs in one launch.  Channels-last activations, every (level, run) image in one
[pixel][C] buffer; `block_table` lists the 16x16-pixel output blocks of all images.  GPU only: there is no CPU path."""
import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import amax, hip


def level_pixel_offsets(levels: Sequence[Tuple[int, int]], copies: int) -> List[int]:
    """First pixel of each level's (copies, H, W, C) slab in the shared buffer (level-major), plus the total."""
    offs, total = [], 0
    for h, w in levels:
        offs.append(total)
        total += copies * h * w
    return offs + [total]


_TABLES = {}


def canvas_layout(h: int, w: int, n: int) -> Tuple[int, int, List[Tuple[int, int]]]:
    """Grid (rows, cols) of a canvas holding n images of h x w -- image i at grid cell (i // cols, i % cols), top-left canvas pixel
    (row * (h + 1), col * (w + 1)): ONE zero row / column between neighbours, the convolution's padding for both -- that needs the
    fewest 16x16 blocks, and the (block_row, block_col) list of the blocks that hold at least one image pixel."""
    best = None
    for rows in range(1, n + 1):
        cols = -(-n // rows)
        if (rows - 1) * cols >= n:                              # an empty grid row
            continue
        by, bx = (rows * (h + 1) - 1 + 15) // 16, (cols * (w + 1) - 1 + 15) // 16
        blocks = []
        for r in range(by):
            m0, m1 = (16 * r) // (h + 1), min((16 * r + 15) // (h + 1), rows - 1)
            if m0 == m1 and 16 * r - m0 * (h + 1) >= h:        # the block row lies on a separator row only
                continue
            for c in range(bx):
                n0, n1 = (16 * c) // (w + 1), min((16 * c + 15) // (w + 1), cols - 1)
                if n0 == n1 and 16 * c - n0 * (w + 1) >= w:
                    continue
                if m0 * cols + n0 < n:                           # the first cell the block touches holds an image (cells fill row-major)
                    blocks.append((r, c))
                elif any(m * cols + k < n for m in range(m0, m1 + 1) for k in range(n0, n1 + 1)):
                    blocks.append((r, c))
        if best is None or len(blocks) < len(best[2]):
            best = (rows, cols, blocks)
    return best


MAX_CANVAS_BYTES = 2 ** 31 - 256     # the kernels address a canvas (the images of one table record) with 32-bit byte offsets; the
                                      # out-of-image sentinel 0x7FFFFF00 must lie past it


def block_table(levels: Sequence[Tuple[int, int]], copies: int, device, in_copies: Optional[int] = None, in_first: int = 0,
                out_copies: Optional[int] = None, channels: int = 512) -> torch.Tensor:
    """int32 (n_blocks, 4) records of pod_wino_conv3x3 (include/pod_mi355x.h): {first pixel of image 0 in the input buffer, in the
    output buffer, grid_cols << 24 | H << 12 | W, n_images << 24 | block_row << 12 | block_col}, one per 16x16-pixel block of a
    CANVAS on which the `copies` images of a level stand in a grid, one zero row / column apart (`canvas_layout`): partial blocks
    at the right and bottom edges are paid once per level, not once per image (a 6 x 11 map costs 1/7 of a block instead of one).
    The input buffer holds `in_copies` images per level (level-major) of which images in_first .. in_first + copies - 1 are read;
    the output buffer holds `out_copies` per level and images 0 .. copies - 1 are written.  `channels`: the larger of the input's and
    the (padded) output's channel count -- it bounds how many images one record may hold (32-bit byte offsets inside a canvas)."""
    in_copies = copies if in_copies is None else in_copies
    out_copies = copies if out_copies is None else out_copies
    assert in_first + copies <= in_copies and copies <= out_copies
    channels = max(int(channels), 8)
    key = (tuple(levels), copies, in_copies, in_first, out_copies, str(device), channels)
    t = _TABLES.get(key)
    if t is None:
        ioffs, ooffs = level_pixel_offsets(levels, in_copies), level_pixel_offsets(levels, out_copies)
        rows, rec_levels = [], []
        for li, ((h, w), ioff, ooff) in enumerate(zip(levels, ioffs, ooffs)):
            assert 0 < h < 4096 and 0 < w < 4096
            if h * w * channels * 4 > MAX_CANVAS_BYTES:
                raise ValueError("pod_wino_conv3x3: one %d x %d image of %d channels exceeds the kernel's 32-bit canvas offsets" % (h, w, channels))
            group = max(1, min(127, MAX_CANVAS_BYTES // (h * w * channels * 4)))   # 32-bit byte offsets inside a canvas; the image count sits in 7 bits
            done = 0
            while done < copies:
                n = min(group, copies - done)
                grows, gcols, blocks = canvas_layout(h, w, n)
                while max(b[0] for b in blocks) >= 4096 or max(b[1] for b in blocks) >= 4096 or gcols > 255:   # 12-bit block indices
                    n = max(1, n // 2)
                    grows, gcols, blocks = canvas_layout(h, w, n)
                rec = torch.tensor([[ioff + (in_first + done) * h * w, ooff + done * h * w, (gcols << 24) | (h << 12) | w, (n << 24) | (r << 12) | c]
                                    for r, c in blocks], dtype=torch.int64)
                rows.append(rec)
                rec_levels.extend([li] * len(blocks))
                done += n
        assert max(ioffs[-1], ooffs[-1]) < 2 ** 31
        t = torch.cat(rows).to(torch.int32).to(device).contiguous()
        t.pod_pixels = copies * sum(h * w for h, w in levels)          # output pixels of a launch with this table
        t.pod_levels = len(levels)
        t.pod_channels = channels
        t.pod_rec_level = torch.tensor(rec_levels, dtype=torch.int32).to(device)       # FPN level of every record (pod_sparse_live_blocks)
        if t.is_cuda and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(t.device).synchronize()           # made once, then read from any stream
        _TABLES[key] = t
    return t


# Which kernel serves a 3x3 convolution.  Default: pod_wino_conv3x3_split (csrc/k12_wino_conv_split.hip) where the channel count allows it
# (C % 16 == 0) -- the same fp32 Winograd with every product formed on the 16-bit matrix cores from split operands (rounds 3-4: exact 3-way
# bf16 splits, 6 partial products; round 5: 2-way f16 splits of the power-of-two-scaled operands, 3 partial products, fp32 accumulate).
# Its contract is tested, not assumed (tests/test_wino_conv_gpu.py): x s == x0 + x1 to 2^-23 |x s|, and on every benchmark shape its error
# against an fp64 convolution is no larger than the fp32-MFMA kernel's.  POD_WINO_SPLIT=0: pod_wino_conv3x3 (fp32 matrix instructions) everywhere.
SPLIT_BF16 = os.environ.get("POD_WINO_SPLIT", "1") != "0"      # (the name is rounds 3-4's; the switch selects the split kernel, whatever its terms)
# Workgroup form of the split kernel (include/pod_mi355x.h: POD_WINO_FORM_4 / _8; bit-identical results): 0 = the library's default
FORM = int(os.environ.get("POD_WINO_FORM", "0"))


def _launch_split(sets, table: torch.Tensor, firsts, C: int, Kpad: int, relu: bool, dropout_p: float, seed: int, epoch, n_splits: int = 0,
                  split_stride: int = 0, what: str = "pod_wino_conv3x3_split", live: Optional[torch.Tensor] = None) -> None:
    """sets: dicts {conv, src, dst, offset, replicas, planes, out_amax: bool}.  Fills a PodWinoConv and launches it; the input abs-max words
    come from pod_compare_amd.amax (the producer's, or pod_absmax), the outputs' are published by the store pass."""
    d = hip.PodWinoConv()
    d.blocks, d.n_blocks, d.n_sets = table.data_ptr(), int(table.shape[0]), len(sets)
    d.C, d.K, d.relu, d.p, d.seed, d.epoch = C, Kpad, 1 if relu else 0, float(dropout_p), int(seed), hip.ptr(epoch)
    d.n_splits, d.split_stride, d.live_blocks, d.form = int(n_splits), int(split_stride), hip.ptr(live), FORM
    for i, s in enumerate(sets):
        q, conv = d.sets[i], s["conv"]
        planes = bool(s.get("planes", False))
        q.in_, q.out, q.Us, q.bias = s["src"].data_ptr(), s["dst"].data_ptr(), conv.U.data_ptr(), hip.ptr(conv.bias) if s.get("bias", True) else None
        q.in_amax = amax.of(s["src"]).data_ptr()
        q.out_amax = amax.produced(s["dst"]).data_ptr() if s.get("out_amax", not planes and n_splits <= 1) else None
        q.offset, q.first_block, q.replicas, q.k_planes = int(s.get("offset", 0)), int(firsts[i]), int(s.get("replicas", 0)), (conv.K if planes else 0)
    hip.check(hip.load().pod_wino_conv3x3_split(ctypes.byref(d), hip.current_stream()), what)


def grouped_launch(sets, relu: bool = False, dropout_p: float = 0.0, seed: int = 0, epoch: Optional[torch.Tensor] = None, live=None) -> None:
    """Up to four convolutions of one shape in ONE grid (pod_wino_conv3x3_split, n_sets > 1).  sets: dicts {conv: WinoConv (split kernel),
    src, dst, table, offset = 0, replicas = 0 (r >= 1: WinoConv.replicas' store pass), planes = False}; every set keeps its own buffers, table, filter, bias, Philox offset.
    Bit for bit the separate launches conv(src, dst, table, ...) / conv.replicas(...)."""
    assert 1 <= len(sets) <= 4
    c0 = sets[0]["conv"]
    firsts, tabs = [], []
    n = 0
    for s in sets:
        conv, src, dst, table = s["conv"], s["src"], s["dst"], s["table"]
        planes = bool(s.get("planes", False))
        assert conv.split and conv.C == c0.C and conv.Kpad == c0.Kpad and src.is_contiguous() and dst.is_contiguous() and src.shape[-1] == conv.C
        assert planes or dst.shape[-1] == conv.Kpad
        assert getattr(table, "pod_channels", 512) >= max(conv.C, conv.K if planes else conv.Kpad)
        firsts.append(n)
        n += int(table.shape[0])
        tabs.append(table)
    key = tuple((t.data_ptr(), t._version, int(t.shape[0])) for t in tabs)
    cat = _GROUPED_TABLES.get(key)
    if cat is None:
        if len(_GROUPED_TABLES) >= 64:
            _GROUPED_TABLES.pop(next(iter(_GROUPED_TABLES)))
        cat = _GROUPED_TABLES[key] = (torch.cat(tabs).contiguous(), tabs)          # (keeps the parts alive: their addresses are the key)
        cat[0].pod_rec_level = torch.cat([t.pod_rec_level for t in tabs]).contiguous()
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(tabs[0].device).synchronize()               # made once, then read from any stream
    # live: a callable (concatenated table) -> device list of its live records (sparse launch), or None
    _launch_split(sets, cat[0], firsts, c0.C, c0.Kpad, relu, dropout_p, seed, epoch, live=None if live is None else live(cat[0]))


_GROUPED_TABLES = {}


class WinoConv:
    """One conv3x3(C -> K, stride 1, pad 1) with its filter transformed once.  K is padded to a multiple of 64 with zero
    filters (outputs written for the padded channels are zero + nothing: bias is padded with zeros too)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], split: Optional[bool] = None):
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
        self.K, self.C = int(weight.shape[0]), int(weight.shape[1])
        self.Kpad = (self.K + 63) // 64 * 64
        if self.C % 8 or self.Kpad not in (64, 128, 256, 512):
            raise ValueError("pod_wino_conv3x3: C %% 8 == 0 and K <= 512 in steps of 64 required, got C=%d K=%d" % (self.C, self.K))
        lib = hip.load()
        self.split = (SPLIT_BF16 if split is None else bool(split)) and self.C % 16 == 0
        if self.split:      # two f16 terms per (scaled) transformed filter value, in the split kernel's load order, + the abs-max trailer
            self.U = torch.empty(lib.pod_wino_filter_split_bytes(self.K, self.C) // 2, dtype=torch.int16, device=weight.device)
            hip.check(lib.pod_wino_filter_transform_split(weight.detach().contiguous().data_ptr(), self.U.data_ptr(), self.K, self.C,
                                                          hip.current_stream()), "pod_wino_filter_transform_split")
        else:
            self.U = torch.empty(24 * self.Kpad * self.C, dtype=torch.float32, device=weight.device)
            hip.check(lib.pod_wino_filter_transform(weight.detach().contiguous().data_ptr(), self.U.data_ptr(), self.K, self.C, hip.current_stream()),
                      "pod_wino_filter_transform")
        self.bias = None
        if bias is not None:
            self.bias = torch.zeros(self.Kpad, dtype=torch.float32, device=weight.device)
            self.bias[:self.K].copy_(bias.detach())
        self.version = (weight._version, None if bias is None else bias._version, weight.data_ptr())

    def __call__(self, src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, relu: bool = False, dropout_p: float = 0.0,
                 seed: int = 0, offset: int = 0, planes: bool = False, epoch: Optional[torch.Tensor] = None, live: Optional[torch.Tensor] = None) -> torch.Tensor:
        """src: (pixels, C) channels-last.  dst: (pixels, Kpad) channels-last, or with planes=True any contiguous buffer of
        NCHW images with K (real) planes each, level-major like the table's output side.  epoch: a device int64 word folded into
        the dropout masks' Philox key (launches replayed from a HIP graph, include/pod_mi355x.h)."""
        assert src.is_contiguous() and dst.is_contiguous() and src.shape[-1] == self.C and src.dtype == dst.dtype == torch.float32
        assert planes or dst.shape[-1] == self.Kpad
        assert getattr(table, "pod_channels", 512) >= max(self.C, self.K if planes else self.Kpad), "block_table(channels=...) below this conv's channel count"
        if self.split:
            _launch_split([{"conv": self, "src": src, "dst": dst, "offset": offset, "planes": planes}], table, [0], self.C, self.Kpad, relu, dropout_p, seed, epoch,
                          live=live)
            return dst
        assert live is None, "sparse launches need the split kernel"
        hip.check(hip.load().pod_wino_conv3x3(src.data_ptr(), dst.data_ptr(), self.U.data_ptr(), hip.ptr(self.bias), table.data_ptr(),
                                              table.shape[0], self.C, self.Kpad, self.K if planes else 0, 1 if relu else 0, float(dropout_p),
                                              seed, offset, hip.ptr(epoch), hip.current_stream()), "pod_wino_conv3x3")
        amax.forget(dst)
        return dst

    def replicas(self, src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, replicas: int, relu: bool = False, dropout_p: float = 0.0,
                 seed: int = 0, offset: int = 0, epoch: Optional[torch.Tensor] = None, live: Optional[torch.Tensor] = None) -> torch.Tensor:
        """conv + bias (+ ReLU) of ONE image per level, stored `replicas` times with a dropout mask each (pod_wino_conv3x3_split_replicas):
        table = block_table(levels, 1, out_copies=replicas); dst: the `replicas` images per level, channels-last.  The masks are those
        of pod_expand_dropout called per level with offset + (first float of the level in dst) / 8 (`expand_offset`)."""
        assert self.split and self.K == self.Kpad and 1 <= replicas <= 127
        assert src.is_contiguous() and dst.is_contiguous() and src.shape[-1] == self.C and dst.shape[-1] == self.Kpad and src.dtype == dst.dtype == torch.float32
        _launch_split([{"conv": self, "src": src, "dst": dst, "offset": offset, "replicas": int(replicas)}], table, [0], self.C, self.Kpad, relu, dropout_p, seed, epoch,
                      live=live)
        return dst

    def splits_for(self, n_blocks: int, cus: int = 256) -> int:
        """How many ways to cut the input channels of a launch of n_blocks output blocks (pod_wino_conv3x3_split_partial): small maps
        give this tiling too few workgroups for the chip (res5 of a 768 x 1344 frame: 6 blocks x 8 filter slices = 48), each walking all
        C / 16 chunks; cutting C makes n_splits x as many workgroups of 1 / n_splits the chunks.  1 = no split."""
        if not self.split:
            return 1
        wgs, nchunk, s = n_blocks * (self.Kpad // 64), self.C // 16, 1
        while s < 4 and wgs * s * 2 <= cus and nchunk % (s * 2) == 0 and (nchunk // (s * 2)) % 2 == 0 and nchunk // (s * 2) >= 4:
            s *= 2
        return s

    def planes_of_one_image(self, src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, relu: bool = False, n_splits: Optional[int] = None) -> torch.Tensor:
        """conv + bias (+ ReLU) of ONE image (src (H*W, C) channels-last) as NCHW planes (dst: K x H*W floats), cutting the input channels
        over workgroup sets when the map is small (`splits_for`); partial sums are added in a fixed order (pod_wino_reduce)."""
        s = self.splits_for(int(table.shape[0])) if n_splits is None else int(n_splits)
        if s <= 1:
            return self(src, dst, table, relu=relu, planes=True)
        hw = int(src.shape[0])
        partials = torch.empty((s, hw, self.Kpad), dtype=torch.float32, device=src.device)
        self._partial(src, partials, table, s, hw)
        hip.check(hip.load().pod_wino_reduce(partials.data_ptr(), s, hw * self.Kpad, hip.ptr(self.bias), dst.data_ptr(), hw, self.Kpad, self.K, 1 if relu else 0,
                                             None, hip.current_stream()), "pod_wino_reduce")
        amax.forget(dst)
        return dst

    def _partial(self, src, partials, table, s, hw):
        _launch_split([{"conv": self, "src": src, "dst": partials, "bias": False, "out_amax": False}], table, [0], self.C, self.Kpad, False, 0.0, 0, None,
                      n_splits=s, split_stride=hw * self.Kpad)

    def channels_last_of_one_image(self, src: torch.Tensor, table: torch.Tensor, relu: bool = False, n_splits: Optional[int] = None,
                                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """conv + bias (+ ReLU) of ONE image, channels-last in and out ((H*W, C) -> (H*W, K); K % 64 == 0): the form a channels-last
        backbone chains (conv1x1.Conv1x1 before and after); small maps are cut over their input channels like `planes_of_one_image`."""
        assert self.K == self.Kpad, "channels-last output needs K % 64 == 0"
        hw = int(src.shape[0])
        dst = torch.empty((hw, self.Kpad), dtype=torch.float32, device=src.device) if out is None else out
        assert dst.is_contiguous() and tuple(dst.shape) == (hw, self.Kpad) and dst.dtype == torch.float32
        s = self.splits_for(int(table.shape[0])) if n_splits is None else int(n_splits)
        if s <= 1:
            return self(src, dst, table, relu=relu)
        partials = torch.empty((s, hw, self.Kpad), dtype=torch.float32, device=src.device)
        self._partial(src, partials, table, s, hw)
        hip.check(hip.load().pod_reduce_partials(partials.data_ptr(), s, hw * self.Kpad, hip.ptr(self.bias), None, dst.data_ptr(), hw * self.Kpad, self.Kpad,
                                                 1 if relu else 0, amax.produced(dst).data_ptr(), hip.current_stream()), "pod_reduce_partials")
        return dst
