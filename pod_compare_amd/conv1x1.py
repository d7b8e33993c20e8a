"""Host side of pod_conv1x1_split (csrc/k13_conv1x1_split.hip): the backbone's / FPN's 1x1 convolutions (detectron2 BottleneckBlock conv1 /
conv3 / shortcut, FPN lateral convs; probabilistic_retinanet.py:96-100 runs them as `self.backbone(images.tensor)`) as a channels-last GEMM
with every fp32 product formed on the 16-bit matrix cores from split operands (round 5: 2-way f16 splits of the power-of-two-scaled
operands, 3 partial products; pod_compare_amd/amax.py carries the operand abs-max words).  GPU only: there is no CPU path."""
from typing import Optional

import torch

from . import amax, hip


class Conv1x1:
    """One conv1x1(Cin -> Cout, stride 1 or 2) with its weight split once.  Activations are channels-last (pixels, C) fp32."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1):
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1)
        self.K, self.C, self.stride = int(weight.shape[0]), int(weight.shape[1]), int(stride)
        if self.C % 16 or self.K % 64 or self.stride not in (1, 2):
            raise ValueError("pod_conv1x1_split: Cin %% 16 == 0, Cout %% 64 == 0, stride 1 or 2 required, got Cin=%d Cout=%d stride=%d" % (self.C, self.K, self.stride))
        self.Ws = torch.empty(2 * self.K * self.C + 8, dtype=torch.int16, device=weight.device)      # two f16 terms per value + the abs-max trailer
        hip.check(hip.load().pod_conv1x1_filter_split(weight.detach().reshape(self.K, self.C).contiguous().data_ptr(), self.Ws.data_ptr(), self.K, self.C,
                                                      hip.current_stream()), "pod_conv1x1_filter_split")
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous().clone()

    @staticmethod
    def eligible(conv: Optional[torch.nn.Conv2d]) -> bool:
        return (conv is not None and tuple(conv.kernel_size) == (1, 1) and tuple(conv.padding) == (0, 0) and conv.groups == 1
                and tuple(conv.dilation) == (1, 1) and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
                and conv.in_channels % 16 == 0 and conv.out_channels % 64 == 0)

    def out_hw(self, h: int, w: int):
        return ((h - 1) // self.stride + 1, (w - 1) // self.stride + 1)

    def splits_for(self, p_out: int, simds: int = 1024) -> int:
        """Cut the input channels over workgroup sets when the map gives the 64-pixel x 64-channel tiling (one wavefront each) too few
        tiles for the chip.  Read off tools/conv1x1_splits.py: alone on its SIMD a wavefront takes ~0.68 us per k-step (16 channels), a
        split costs the reduce launch (~3 us) plus writing s and reading s + 1 copies of the output at ~4 TB/s, and past one wavefront
        per SIMD nothing is gained."""
        tiles = ((p_out + 63) // 64) * (self.K // 64)
        nks = self.C // 16
        out_mb = p_out * self.K * 4 / 1e6
        step = 0.38                                              # us per k-step of a lone wavefront with the 2-way f16 split (round 5)
        best, best_t = 1, (nks // self.auto_waves(tiles, nks)) * step
        for s in (2, 4, 8, 16):
            if nks % s or nks // s < 4 or tiles * s > simds:
                continue
            t = (nks // s // self.auto_waves(tiles * s, nks // s)) * step + 3.0 + (s + 1) * out_mb / 4.0
            if t < best_t:
                best, best_t = s, t
        return best

    @staticmethod
    def auto_waves(tiles: int, per_split: int) -> int:
        """The library's choice of wavefronts per workgroup (pod_conv1x1_split, waves = 0): split-K inside the workgroup, accumulators added in
        LDS -- as many as keep the launch within one wavefront per SIMD and leave every wavefront at least 8 k-steps, in whole pairs."""
        waves = 1
        while waves < 4 and per_split % (waves * 4) == 0 and per_split // (waves * 2) >= 8 and tiles * waves * 2 <= 1024:
            waves *= 2
        return waves

    def __call__(self, x: torch.Tensor, h: int, w: int, relu: bool = False, residual: Optional[torch.Tensor] = None,
                 n_splits: Optional[int] = None, waves: int = 0, out: Optional[torch.Tensor] = None, residual_up2: bool = False) -> torch.Tensor:
        """x: (h * w, Cin) channels-last of ONE image -> (h_out * w_out, Cout) channels-last = act(conv(x) + bias [+ residual]).
        residual_up2: `residual` is the half-resolution map ((h_out + 1) // 2 x (w_out + 1) // 2 pixels), added nearest-neighbour upsampled
        (FPN's top-down sum) without being materialised."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (h * w, self.C)
        ho, wo = self.out_hw(h, w)
        y = torch.empty((ho * wo, self.K), dtype=torch.float32, device=x.device) if out is None else out
        assert y.is_contiguous() and tuple(y.shape) == (ho * wo, self.K) and y.dtype == torch.float32
        s = self.splits_for(ho * wo) if n_splits is None else int(n_splits)
        if residual is not None and residual_up2:
            hr, wr = (ho + 1) // 2, (wo + 1) // 2
            assert residual.is_contiguous() and tuple(residual.shape) == (hr * wr, self.K) and residual.dtype == torch.float32
            if s > 1:                                            # (the second launch of a split adds full-resolution residuals only)
                residual = residual.view(hr, wr, self.K).repeat_interleave(2, 0).repeat_interleave(2, 1)[:ho, :wo].reshape(ho * wo, self.K).contiguous()
                residual_up2 = False
        else:
            residual_up2 = False
        if residual is not None and not residual_up2:
            assert residual.is_contiguous() and tuple(residual.shape) == tuple(y.shape) and residual.dtype == torch.float32
        partials = torch.empty((s, ho * wo, self.K), dtype=torch.float32, device=x.device) if s > 1 else None
        hip.check(hip.load().pod_conv1x1_split(x.data_ptr(), y.data_ptr(), self.Ws.data_ptr(), hip.ptr(self.bias), hip.ptr(residual), ho, wo, h, w, self.stride,
                                               self.C, self.K, (1 if relu else 0) | (2 if residual_up2 else 0), s, hip.ptr(partials), int(waves), amax.of(x).data_ptr(), amax.produced(y).data_ptr(),
                                               hip.current_stream()), "pod_conv1x1_split")
        return y


class Stem7x7:
    """The ResNet stem conv (7x7, stride 2, padding 3, 3 -> 64 channels; FrozenBN folded into weight + bias) with its weight split once:
    (1, 3, H, W) fp32 planes -> ((H-1)//2+1) x ((W-1)//2+1) pixels x 64 channels, channels-last (pod_stem7x7_split)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        assert weight.is_cuda and weight.dtype == torch.float32 and tuple(weight.shape) == (64, 3, 7, 7)
        self.Ws = torch.empty(2 * 64 * 192 + 8, dtype=torch.int16, device=weight.device)       # two f16 terms per value + the abs-max trailer
        self._bounds = {}
        hip.check(hip.load().pod_stem7x7_filter_split(weight.detach().contiguous().data_ptr(), self.Ws.data_ptr(), hip.current_stream()), "pod_stem7x7_filter_split")
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous().clone()

    @staticmethod
    def eligible(conv: Optional[torch.nn.Conv2d]) -> bool:
        return (conv is not None and tuple(conv.weight.shape) == (64, 3, 7, 7) and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (3, 3)
                and conv.groups == 1 and tuple(conv.dilation) == (1, 1))

    def __call__(self, x: torch.Tensor, relu: bool = True, mean: Optional[torch.Tensor] = None, std: Optional[torch.Tensor] = None,
                 padded_hw: Optional[tuple] = None):
        """x: (1, 3, H, W) or (3, H, W), contiguous, fp32 or uint8 -> (y (Ho * Wo, 64), Ho, Wo).  mean / std (3 floats each): the frame is
        normalised on load; padded_hw: the extent the frame is zero-padded to (the output's size follows it)."""
        assert x.is_cuda and x.dtype in (torch.float32, torch.uint8) and x.is_contiguous() and x.numel() == 3 * x.shape[-2] * x.shape[-1]
        hi, wi = int(x.shape[-2]), int(x.shape[-1])
        h, w = (hi, wi) if padded_hw is None else (int(padded_hw[0]), int(padded_hw[1]))
        assert (mean is None) == (std is None) and h >= hi and w >= wi
        if mean is not None:
            assert mean.is_cuda and std.is_cuda and mean.dtype == std.dtype == torch.float32 and mean.numel() == std.numel() == 3 and mean.is_contiguous() and std.is_contiguous()
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((ho * wo, 64), dtype=torch.float32, device=x.device)
        bound = self._input_bound(x, mean, std)
        hip.check(hip.load().pod_stem7x7_split(x.data_ptr(), 1 if x.dtype == torch.uint8 else 0, hi, wi, hip.ptr(mean), hip.ptr(std), y.data_ptr(), self.Ws.data_ptr(),
                                               hip.ptr(self.bias), h, w, 1 if relu else 0, bound.data_ptr(), amax.produced(y).data_ptr(),
                                               hip.current_stream()), "pod_stem7x7_split")
        return y, ho, wo

    def _input_bound(self, x: torch.Tensor, mean: Optional[torch.Tensor], std: Optional[torch.Tensor]) -> torch.Tensor:
        """A device record bounding max |normalised input| (the kernel's `in_amax`): |(x - mean) / std| <= (max |x| + max |mean|) / min |std|, with
        max |x| = 255 for a uint8 frame (a constant word per (mean, std)) and the frame's own abs-max word otherwise."""
        if mean is None:
            return amax.of(x)
        key = (mean.data_ptr(), mean._version, std.data_ptr(), std._version)
        c = self._bounds.get(key)
        if c is None:
            inv = 1.0 / float(std.abs().min())
            c = self._bounds[key] = (inv, torch.full((1,), float(mean.abs().max()) * inv, dtype=torch.float32, device=x.device),
                                     torch.full((amax.RECORD,), (255.0 + float(mean.abs().max())) * inv, dtype=torch.float32, device=x.device))
            torch.cuda.current_stream(x.device).synchronize()              # made once, then read from any stream
        if x.dtype == torch.uint8:
            return c[2]
        return torch.add(c[1], amax.of(x), alpha=c[0])


class Conv3x3S2:
    """conv3x3(stride 2, padding 1) on a channels-last map as pod_im2col3x3s2_cl + pod_conv1x1_split: FPN's LastLevelP6P7 (p6 on res5, p7 on
    relu(p6): 252 and 66 output pixels at the benchmark frame -- the long K of the patch matrix, 9 Cin, is cut over workgroup sets and
    wavefronts by the 1x1 kernel's own policy).  Bit-reproducible run to run (fixed-order partial sums), unlike the MIOpen kernels it replaces."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
        self.K, self.C = int(weight.shape[0]), int(weight.shape[1])
        w9 = weight.detach().permute(0, 2, 3, 1).reshape(self.K, 9 * self.C, 1, 1).contiguous()       # (Cout, ty, tx, Cin): the patch matrix's column order
        self.gemm = Conv1x1(w9, bias, 1)

    @staticmethod
    def eligible(conv: Optional[torch.nn.Conv2d]) -> bool:
        return (conv is not None and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (1, 1)
                and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and conv.in_channels % 16 == 0 and conv.out_channels % 64 == 0)

    @staticmethod
    def out_hw(h: int, w: int):
        return ((h - 1) // 2 + 1, (w - 1) // 2 + 1)

    def __call__(self, x: torch.Tensor, h: int, w: int, relu_input: bool = False, relu: bool = False, out: Optional[torch.Tensor] = None):
        """x: (h * w, Cin) channels-last of ONE image -> (y (ho * wo, Cout) channels-last = act(conv(act_in(x)) + bias), ho, wo)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (h * w, self.C)
        ho, wo = self.out_hw(h, w)
        cols = torch.empty((ho * wo, 9 * self.C), dtype=torch.float32, device=x.device)
        bound = amax.of(x)                                   # max |x| bounds the patch matrix too (its entries are x's, relu'd or not, and zeros)
        hip.check(hip.load().pod_im2col3x3s2_cl(x.data_ptr(), cols.data_ptr(), h, w, self.C, 1 if relu_input else 0, hip.current_stream()), "pod_im2col3x3s2_cl")
        amax.attach(cols, bound)
        return self.gemm(cols, ho, wo, relu=relu, out=out), ho, wo


def maxpool3x3s2_cl(x: torch.Tensor, h: int, w: int):
    """max_pool2d(kernel 3, stride 2, padding 1) of a channels-last map (h * w, C) -> (y (hp * wp, C), hp, wp)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[0] == h * w and x.shape[1] % 4 == 0
    hp, wp = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((hp * wp, x.shape[1]), dtype=torch.float32, device=x.device)
    hip.check(hip.load().pod_maxpool3x3s2_cl(x.data_ptr(), y.data_ptr(), h, w, int(x.shape[1]), hip.current_stream()), "pod_maxpool3x3s2_cl")
    rec = getattr(x, "_pod_amax", None)
    if rec is not None and rec[1] == x._version:
        amax.attach(y, rec[0])                     # every output IS one of the inputs: the bound carries over
    return y, hp, wp
