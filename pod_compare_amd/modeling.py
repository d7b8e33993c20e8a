"""ResNet-50-FPN + probabilistic RetinaNet head on PyTorch-ROCm (row a1 of SURVEY 8a).

The reference model (probabilistic_modeling/probabilistic_retinanet.py) subclasses detectron2's
RetinaNet, which is not installed; this is a from-scratch statement of the same architecture in
plain torch (convolutions run in MIOpen -- they are not part of the hand-written hot path):

  backbone  detectron2 `build_retinanet_resnet_fpn_backbone`: ResNet-50 (FrozenBN, stride in the
            1x1 conv of each bottleneck), FPN on res3..res5 (256 ch) + LastLevelP6P7 from res5
            (Base-RetinaNet.yaml:3-10).
  head      ProbabilisticRetinaNetHead, PR:370-537: two 4 x (conv3x3 256->256, ReLU, Dropout p)
            subnets; cls_score 256->A*K (bias -log(99), PR:454-455), bbox_pred 256->A*4,
            cls_var 256->A*K (bias -10, PR:458-470), bbox_cov 256->A*D (std 1e-4, PR:473-484).

MI355X-first differences from the reference's forward (PR:95-108, PR:486-537), numerically neutral:
  * MC-dropout runs are batched along the batch dimension instead of replicating python lists of
    feature maps N times (PR:104-106); outputs stay NCHW `(N, A*C, H, W)` -- exactly the layout
    kernel K1 streams -- so `permute_to_N_HWA_K` (PR:343-349) never runs;
  * the first conv+ReLU of each subnet sees the same input in every run (dropout only follows it),
    so it is evaluated once and broadcast;
  * in MC mode the mean and variance branches still use independent dropout draws (SURVEY Q2:
    the reference evaluates each subnet twice, PR:518-523); without dropout the two evaluations are
    identical and the trunk is shared (SURVEY f-4);
  * dropout is enabled by a flag instead of `model.train()` (PI:53-56).
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import anchors as _anchors
from .synthetic import HeadOutputs

PIXEL_MEAN_BGR = (103.530, 116.280, 123.675)   # detectron2 defaults used by the BDD configs
PIXEL_STD = (1.0, 1.0, 1.0)


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm with fixed statistics and affine terms (detectron2 FrozenBatchNorm2d)."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def fold_frozen_bn(module: nn.Module) -> int:
    """Folds every `Sequential(Conv2d(bias=False), FrozenBatchNorm2d)` pair into one biased conv (w * scale, shift):
    the same affine map without ~100 tiny element-wise launches per image in the batch-1 backbone.  Returns the number
    of pairs folded.  Inference only (the statistics are frozen by definition)."""
    n = 0
    for child in module.children():
        if isinstance(child, nn.Sequential) and len(child) == 2 and isinstance(child[0], nn.Conv2d) and \
                isinstance(child[1], FrozenBatchNorm2d) and child[0].bias is None:
            conv, bn = child[0], child[1]
            scale = bn.weight * (bn.running_var + bn.eps).rsqrt()
            shift = bn.bias - bn.running_mean * scale
            with torch.no_grad():
                conv.weight.mul_(scale.reshape(-1, 1, 1, 1))
                conv.bias = nn.Parameter(shift.clone(), requires_grad=False)
            child[1] = nn.Identity()
            n += 1
        else:
            n += fold_frozen_bn(child)
    if n:                                                       # (a model serving HIP graphs: they were captured against the unfolded filters)
        bump_param_generation()
    return n


# Anything that replaces or moves parameter STORAGE without bumping a tensor's version counter -- nn.Module._apply on the model or on ANY of
# its submodules (model.head.to(...), model.bottom_up.float()), load_state_dict, fold_frozen_bn on a part of the model -- moves this
# process-wide counter on; every model's graph fingerprint contains it, so graphs captured against the old storage are dropped (ADVICE r5:
# a counter on the root module alone missed the submodule cases).
_PARAM_GENERATION = [0]


def bump_param_generation() -> None:
    _PARAM_GENERATION[0] += 1


class _TracksStorage(nn.Module):
    """nn.Module whose `_apply` (to / cuda / float / ...) is seen by the graph fingerprints."""

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        bump_param_generation()
        return out


def _plain_conv(m) -> Optional[nn.Conv2d]:
    """The biased conv behind `m` if its element-wise tail can be fused: a Conv2d, or a folded (conv, Identity) pair."""
    if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.Identity):
        m = m[0]
    return m if isinstance(m, nn.Conv2d) else None


FUSE_CONV_TAIL = True     # GPU only: conv without bias + ONE pod_bias_act pass (bias, residual, ReLU, dropout)


def conv_bias_act(m, x: torch.Tensor, relu: bool = False, residual: Optional[torch.Tensor] = None, residual_module=None,
                  residual_input: Optional[torch.Tensor] = None, dropout_p: float = 0.0, seed: int = 0, offset: int = 0,
                  out_nchw: bool = False, residual_raw: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(m(x) [+ residual]) with the element-wise tail in one HIP pass.

    torch's conv on ROCm is MIOpen's kernel plus a separate bias `add_`; followed by clamp (and the bottleneck's add, and
    dropout) that is 2-4 read+write passes over the activation.  Here the conv runs without its bias and pod_bias_act does
    `dropout(relu((y + b[c]) + (r + rb[c])))` in place.  `residual_module(residual_input)` (a shortcut conv) is evaluated without
    its bias too and folded into the same pass.  CPU tensors / unfolded BN / non-fp32 take the plain torch ops."""
    conv = _plain_conv(m)
    rconv = _plain_conv(residual_module) if residual_module is not None else None
    fuse = (FUSE_CONV_TAIL and conv is not None and x.is_cuda and x.dtype == torch.float32 and
            (residual_module is None or rconv is not None))
    if not fuse:
        y = m(x)
        if residual_module is not None:
            residual = residual_module(residual_input)
        if residual is not None:
            y = y + residual
        if relu:
            y = F.relu_(y)
        return F.dropout(y, dropout_p, training=True) if dropout_p > 0.0 else y
    from . import hip
    lib = hip.load()
    nhwc = x.dim() == 4 and x.shape[1] > 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    y = F.conv2d(x, _weight_for(conv, nhwc), None, conv.stride, conv.padding, conv.dilation, conv.groups)
    res_bias = None
    if rconv is not None:
        residual = residual_raw if residual_raw is not None else shortcut_raw(rconv, residual_input)
        res_bias = rconv.bias
    C, HW = y.shape[1], y.shape[2] * y.shape[3]
    if nhwc and residual is None and y.is_contiguous(memory_format=torch.channels_last):
        if out_nchw and C % 4 == 0 and HW % 4 == 0:
            # leave the channels-last trunk on the element-wise pass that exists anyway: NHWC in, NCHW planes out
            out = torch.empty(y.shape, dtype=y.dtype, device=y.device)
            hip.check(lib.pod_bias_act_to_nchw(y.data_ptr(), out.data_ptr(), hip.ptr(conv.bias), y.shape[0], C, HW, 1 if relu else 0,
                                               float(dropout_p), seed, offset, hip.current_stream()), "pod_bias_act_to_nchw")
            return out
        HW = 1                                  # NHWC: channel = flat index mod C (include/pod_mi355x.h)
    else:
        if not y.is_contiguous():
            y = y.contiguous()
        if residual is not None:
            assert residual.shape == y.shape
            residual = residual.contiguous()
    hip.check(lib.pod_bias_act(y.data_ptr(), hip.ptr(conv.bias), hip.ptr(residual), hip.ptr(res_bias), y.numel(), C, HW,
                               1 if relu else 0, float(dropout_p), seed, offset, hip.current_stream()), "pod_bias_act")
    return y


def shortcut_raw(rconv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A shortcut conv without its bias (the bias rides on the pass that adds the residual)."""
    return F.conv2d(x, rconv.weight, None, rconv.stride, rconv.padding, rconv.dilation, rconv.groups)


def _weight_for(conv: nn.Conv2d, channels_last: bool) -> torch.Tensor:
    """conv.weight, or a cached channels-last copy of it (refreshed when the parameter is modified or moved)."""
    w = conv.weight
    if not channels_last:
        return w
    key = (w.data_ptr(), w._version)
    cached = getattr(conv, "_pod_w_nhwc", None)
    if cached is None or cached[0] != key:
        cached = (key, w.detach().contiguous(memory_format=torch.channels_last))
        if w.is_cuda:
            torch.cuda.current_stream(w.device).synchronize()   # made once, then read from any stream
        conv._pod_w_nhwc = cached
    return cached[1]


WINO_HEAD = True              # GPU only: the head subnets' 3x3 convs on pod_wino_conv3x3 (all levels, all runs per launch)
WINO_BACKBONE = __import__("os").environ.get("POD_WINO_BACKBONE", "1") != "0"   # GPU only: the bottlenecks' and the FPN's 3x3 / stride-1 convs on pod_wino_conv3x3 too (batch 1: few
                              # workgroups per launch, but a third of MIOpen's CU-time per FLOP -- the other streams' images fill the idle CUs)
WINO_BACKBONE_MIN_CELLS = int(__import__("os").environ.get("POD_WINO_MIN_CELLS", "0"))   # (experiment knob: smaller maps stay on MIOpen)
NHWC_TRUNK_MIN_CELLS = 8192   # head trunks of maps at least this large run channels-last (p3 of a 768x1344 input: 16128)


def wino_of(conv: nn.Conv2d):
    """The conv's Winograd-transformed filter (pod_wino_filter_transform), refreshed when the parameters change."""
    from . import wino
    from .wino import WinoConv
    key = (conv.weight.data_ptr(), conv.weight._version, None if conv.bias is None else (conv.bias.data_ptr(), conv.bias._version), wino.SPLIT_BF16)
    cached = getattr(conv, "_pod_wino", None)
    if cached is None or cached[0] != key:
        cached = (key, WinoConv(conv.weight, conv.bias))
        torch.cuda.current_stream(conv.weight.device).synchronize()      # made once, then read from any stream
        conv._pod_wino = cached
    return cached[1]


def _wino_limits() -> int:
    from .wino import MAX_CANVAS_BYTES
    return MAX_CANVAS_BYTES


def wino_eligible(conv: Optional[nn.Conv2d], x: torch.Tensor) -> bool:
    """pod_wino_conv3x3 can stand in for `conv` on x: 3x3 / stride 1 / pad 1, fp32 on the GPU, one image, channel counts it tiles."""
    return (WINO_BACKBONE and FUSE_CONV_TAIL and conv is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1
            and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1) and conv.groups == 1
            and tuple(conv.dilation) == (1, 1) and conv.in_channels % 8 == 0 and conv.out_channels in (64, 128, 256, 512)
            and x.shape[2] < 4096 and x.shape[3] < 4096 and x.shape[2] * x.shape[3] >= WINO_BACKBONE_MIN_CELLS
            and x.shape[2] * x.shape[3] * max(conv.in_channels, conv.out_channels) * 4 <= _wino_limits())      # 32-bit offsets inside a canvas


def wino_conv_nchw(conv: nn.Conv2d, x: torch.Tensor, relu: bool, pre_bias: Optional[torch.Tensor] = None, pre_relu: bool = False) -> torch.Tensor:
    """act(conv(pre_act(x + pre_bias))) for an NCHW x (1, C, H, W): the element-wise tail of the PRODUCER of x (bias + ReLU of the conv
    before) rides on the pass that lays x out channels-last (pod_bias_act_to_nhwc), the 3x3 conv runs on pod_wino_conv3x3 with its own
    bias + ReLU in the store, and the result comes back as NCHW planes."""
    from . import hip
    from .wino import block_table
    _, C, H, W = x.shape
    a = torch.empty((H * W, C), dtype=x.dtype, device=x.device)
    hip.check(hip.load().pod_bias_act_to_nhwc(x.contiguous().data_ptr(), a.data_ptr(), hip.ptr(pre_bias), 1, C, H * W, 1 if pre_relu else 0,
                                              hip.current_stream()), "pod_bias_act_to_nhwc")
    out = torch.empty((1, conv.out_channels, H, W), dtype=x.dtype, device=x.device)
    wino_of(conv).planes_of_one_image(a, out.view(-1), block_table([(H, W)], 1, x.device), relu=relu)     # (small maps: split over C)
    return out


_SIDE_STREAMS: Dict[Tuple[int, int], List["torch.cuda.Stream"]] = {}


BRANCHES = set(x for x in __import__("os").environ.get("POD_GRAPH_BRANCHES", "").split(",") if x)    # which forks are taken: head, pred, fpn, shortcut


def branches(fns, kind=""):
    """Runs independent pieces of a forward: one after the other on the current stream -- or, for the kinds named in
    POD_GRAPH_BRANCHES and while the forward is being CAPTURED into a HIP graph, each on its own stream, forked from and joined back
    into the capturing stream, so that the graph holds them as parallel branches (a single-run head layer is 372 workgroups for 256
    CUs, a predictor 93, an FPN output conv of p5 24: side by side they would fill the chip).  MEASURED (round 4, cfg2, images/s): no
    branches 398; cls / bbox trunks 393; the four predictors 241; FPN output convs 277; bottleneck shortcuts 351 -- and 1.0 - 1.6 ms
    of host time per replay instead of 0.13: ROCm's graph executor pays more for every fork / join than the idle CUs were worth.
    Off by default; kept as the switch that reproduces the measurement."""
    if len(fns) < 2 or kind not in BRANCHES or not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return [f() for f in fns]
    cur = torch.cuda.current_stream()
    key = (cur.device.index or 0, cur.cuda_stream)
    side = _SIDE_STREAMS.setdefault(key, [])
    while len(side) < len(fns) - 1:
        side.append(torch.cuda.Stream(device=cur.device))
    outs = [None] * len(fns)
    for st in side[:len(fns) - 1]:
        st.wait_stream(cur)
    for i, f in enumerate(fns[1:]):
        with torch.cuda.stream(side[i]):
            outs[i + 1] = f()
    outs[0] = fns[0]()
    for st in side[:len(fns) - 1]:
        cur.wait_stream(st)
    return outs


# ---- channels-last backbone (round 4) ---------------------------------------------------------------------------------------
# With the 3x3 convolutions on pod_wino_conv3x3[_split] (channels-last in and out) the 1x1 convolutions were what kept the backbone in
# NCHW: MIOpen GEMM + one element-wise pass each for bias / residual / ReLU, plus a layout pass in front of every 3x3.  On
# pod_conv1x1_split (csrc/k13_conv1x1_split.hip: channels-last GEMM, bias + residual + ReLU in the store, same exact-split products as
# the 3x3 kernel) a bottleneck is three launches on (pixels, C) buffers and the whole trunk stays channels-last from the max-pool on.
GROUPED_HEAD = __import__("os").environ.get("POD_GROUPED_HEAD", "1") != "0"         # layer l of the cls and the bbox subnet, and the four predictors, in one launch each
FUSED_REPLICAS = __import__("os").environ.get("POD_FUSED_REPLICAS", "1") != "0"     # the first conv of an MC-dropout subnet stores its masked replicas itself
CL_BACKBONE = __import__("os").environ.get("POD_CL_BACKBONE", "1") != "0"
FUSED_PREPROCESS = __import__("os").environ.get("POD_FUSED_PREPROCESS", "1") != "0"   # ... which then also normalises and pads the frame on load
HIP_P6P7 = __import__("os").environ.get("POD_HIP_P6P7", "1") != "0"         # FPN's p6 / p7 (3x3 / stride 2) as pod_im2col3x3s2_cl + pod_conv1x1_split instead of MIOpen
FUSED_TOPDOWN = __import__("os").environ.get("POD_FUSED_TOPDOWN", "1") != "0"   # FPN's top-down map read at half resolution by the lateral conv's store (POD_C1_RESIDUAL_UP2) instead of F.interpolate
HIP_STEM = __import__("os").environ.get("POD_HIP_STEM", "1") != "0"         # the 7x7 stem + max-pool of the channels-last trunk on pod_stem7x7_split / pod_maxpool3x3s2_cl


def c1_of(conv: nn.Conv2d):
    """The conv's pod_conv1x1_split form (weight split once), refreshed when the parameters change."""
    from .conv1x1 import Conv1x1
    key = (conv.weight.data_ptr(), conv.weight._version, None if conv.bias is None else (conv.bias.data_ptr(), conv.bias._version))
    cached = getattr(conv, "_pod_c1", None)
    if cached is None or cached[0] != key:
        cached = (key, Conv1x1(conv.weight, conv.bias, conv.stride[0]))
        torch.cuda.current_stream(conv.weight.device).synchronize()      # made once, then read from any stream
        conv._pod_c1 = cached
    return cached[1]


def stem_of(conv: nn.Conv2d):
    """The stem conv's pod_stem7x7_split form (weight split once), refreshed when the parameters change."""
    from .conv1x1 import Stem7x7
    key = (conv.weight.data_ptr(), conv.weight._version, None if conv.bias is None else (conv.bias.data_ptr(), conv.bias._version))
    cached = getattr(conv, "_pod_stem", None)
    if cached is None or cached[0] != key:
        cached = (key, Stem7x7(conv.weight, conv.bias))
        torch.cuda.current_stream(conv.weight.device).synchronize()      # made once, then read from any stream
        conv._pod_stem = cached
    return cached[1]


def s2_of(conv: nn.Conv2d):
    """The conv's im2col + pod_conv1x1_split form (3x3 / stride 2: FPN's p6 / p7; weight re-laid and split once), refreshed when the parameters change."""
    from .conv1x1 import Conv3x3S2
    key = (conv.weight.data_ptr(), conv.weight._version, None if conv.bias is None else (conv.bias.data_ptr(), conv.bias._version))
    cached = getattr(conv, "_pod_s2", None)
    if cached is None or cached[0] != key:
        cached = (key, Conv3x3S2(conv.weight, conv.bias))
        torch.cuda.current_stream(conv.weight.device).synchronize()      # made once, then read from any stream
        conv._pod_s2 = cached
    return cached[1]


def _c1_ok(conv: Optional[nn.Conv2d]) -> bool:
    from .conv1x1 import Conv1x1
    return conv is not None and conv.bias is not None and Conv1x1.eligible(conv)


def _w3_ok(conv: Optional[nn.Conv2d]) -> bool:
    return (conv is not None and conv.bias is not None and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (1, 1) and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and conv.in_channels % 16 == 0
            and conv.out_channels in (64, 128, 256, 512))


def wino_cl(conv: nn.Conv2d, x: torch.Tensor, h: int, w: int, relu: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(conv3x3(x) + bias) of one channels-last image (h * w, C) -> (h * w, K)."""
    from .wino import block_table
    return wino_of(conv).channels_last_of_one_image(x, block_table([(h, w)], 1, x.device, channels=max(conv.in_channels, conv.out_channels)), relu=relu, out=out)


def levels_channels_last(features: List[torch.Tensor]) -> torch.Tensor:
    """The FPN levels as ONE channels-last buffer, level after level (what the head's grouped launches read): the buffer FPN.forward_cl wrote
    them into when they still are its consecutive slices (no copy), else their concatenation."""
    buf = getattr(features[0], "_pod_cl_levels", None)
    if buf is not None and buf.shape[0] == sum(int(f.shape[2]) * int(f.shape[3]) for f in features) and buf._version == features[0]._pod_cl_version:
        at, ok = buf.data_ptr(), True
        for f in features:
            ok = ok and f.shape[0] == 1 and f.shape[1] == buf.shape[1] and f.data_ptr() == at and f.stride(1) == 1
            at += int(f.shape[2]) * int(f.shape[3]) * int(buf.shape[1]) * 4
        if ok:
            return buf
    return torch.cat([f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]) for f in features])


def cl_as_nchw(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """(h * w, C) channels-last buffer as a (1, C, h, w) tensor (channels_last strides: a view)."""
    return x.view(1, h, w, x.shape[1]).permute(0, 3, 1, 2)


def nchw_as_cl(x: torch.Tensor) -> torch.Tensor:
    """(1, C, h, w) tensor of either memory format as an (h * w, C) channels-last buffer (a view when it already is channels-last)."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _conv_bn(cin, cout, k, stride=1, padding=0):
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
    nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")   # c2_msra_fill
    return nn.Sequential(conv, FrozenBatchNorm2d(cout))


class Bottleneck(_TracksStorage):
    def __init__(self, cin, cout, mid, stride):
        super().__init__()
        self.shortcut = _conv_bn(cin, cout, 1, stride) if cin != cout else None
        self.conv1 = _conv_bn(cin, mid, 1, stride)          # STRIDE_IN_1X1 = True (MSRA R-50)
        self.conv2 = _conv_bn(mid, mid, 3, 1, 1)
        self.conv3 = _conv_bn(mid, cout, 1)

    def forward(self, x):
        c1, c2 = _plain_conv(self.conv1), _plain_conv(self.conv2)

        def main():
            if c1 is not None and c1.bias is not None and c2 is not None and c2.bias is not None and wino_eligible(c2, x):
                y = F.conv2d(x, c1.weight, None, c1.stride, c1.padding)                   # conv1 without its bias (MIOpen, NCHW)
                return wino_conv_nchw(c2, y, relu=True, pre_bias=c1.bias, pre_relu=True)  # its bias + ReLU, then conv2 + bias + ReLU
            return conv_bias_act(self.conv2, conv_bias_act(self.conv1, x, relu=True), relu=True)

        if self.shortcut is None:
            return conv_bias_act(self.conv3, main(), relu=True, residual=x)
        rconv = _plain_conv(self.shortcut)
        if rconv is None or rconv.bias is None or not (FUSE_CONV_TAIL and x.is_cuda and x.dtype == torch.float32):
            return conv_bias_act(self.conv3, main(), relu=True, residual_module=self.shortcut, residual_input=x)
        out, raw = branches([main, lambda: shortcut_raw(rconv, x)], "shortcut")           # (parallel graph branches when captured)
        return conv_bias_act(self.conv3, out, relu=True, residual_module=self.shortcut, residual_input=x, residual_raw=raw)


    def cl_eligible(self) -> bool:
        convs = (_plain_conv(self.conv1), _plain_conv(self.conv2), _plain_conv(self.conv3))
        sc = None if self.shortcut is None else _plain_conv(self.shortcut)
        return _c1_ok(convs[0]) and _w3_ok(convs[1]) and _c1_ok(convs[2]) and (self.shortcut is None or _c1_ok(sc))

    def forward_cl(self, x: torch.Tensor, h: int, w: int):
        """The block on a channels-last buffer x (h * w, Cin): conv1 (1x1, stride) -> conv2 (3x3) -> conv3 (1x1) + shortcut + ReLU, three
        (four) launches, every bias / ReLU / residual in the producing kernel's store.  Returns (y, h_out, w_out)."""
        c1, c2, c3 = c1_of(_plain_conv(self.conv1)), _plain_conv(self.conv2), c1_of(_plain_conv(self.conv3))
        h1, w1 = c1.out_hw(h, w)
        a = c1(x, h, w, relu=True)
        b = wino_cl(c2, a, h1, w1, relu=True)
        r = x if self.shortcut is None else c1_of(_plain_conv(self.shortcut))(x, h, w, relu=False)
        return c3(b, h1, w1, relu=True, residual=r), h1, w1


class ResNet50(_TracksStorage):
    """res2..res5 of ResNet-50; returns res3, res4, res5 (strides 8, 16, 32)."""

    def __init__(self):
        super().__init__()
        self.stem = _conv_bn(3, 64, 7, 2, 3)
        cfg = ((3, 64, 256, 1), (4, 128, 512, 2), (6, 256, 1024, 2), (3, 512, 2048, 2))
        stages, cin = [], 64
        for blocks, mid, cout, stride in cfg:
            layers = []
            for b in range(blocks):
                layers.append(Bottleneck(cin, cout, mid, stride if b == 0 else 1))
                cin = cout
            stages.append(nn.Sequential(*layers))
        self.res2, self.res3, self.res4, self.res5 = stages

    def forward(self, x):
        x = conv_bias_act(self.stem, x, relu=True)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        x = self.res2(x)
        c3 = self.res3(x)
        c4 = self.res4(c3)
        c5 = self.res5(c4)
        return c3, c4, c5

    def cl_eligible(self) -> bool:
        return all(b.cl_eligible() for stage in (self.res2, self.res3, self.res4, self.res5) for b in stage)

    def hip_stem_ok(self) -> bool:
        from .conv1x1 import Stem7x7
        stem = _plain_conv(self.stem)
        return HIP_STEM and stem is not None and stem.bias is not None and Stem7x7.eligible(stem)

    def forward_cl(self, x, frame=None):
        """Channels-last from the stem on: returns [(c3, h, w), (c4, h, w), (c5, h, w)] with c* (h * w, C) buffers.  frame = (image (3, h, w)
        uint8 / fp32 on the device, pixel mean, pixel std, padded (h, w)) instead of x: the stem kernel normalises and pads on load."""
        stem = _plain_conv(self.stem)
        from .conv1x1 import maxpool3x3s2_cl
        if frame is not None:
            img, mean, std, padded = frame
            t, h, w = stem_of(stem)(img, relu=True, mean=mean, std=std, padded_hw=padded)
            t, h, w = maxpool3x3s2_cl(t, h, w)
        elif self.hip_stem_ok() and x.shape[0] == 1 and x.is_contiguous():
            t, h, w = stem_of(stem)(x, relu=True)                        # 7x7 / stride 2 on pod_stem7x7_split: channels-last out
            t, h, w = maxpool3x3s2_cl(t, h, w)
        else:
            x = conv_bias_act(self.stem, x, relu=True)                   # MIOpen, NCHW
            x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
            h, w = int(x.shape[2]), int(x.shape[3])
            t = nchw_as_cl(x)
        outs = []
        for stage in (self.res2, self.res3, self.res4, self.res5):
            for block in stage:
                t, h, w = block.forward_cl(t, h, w)
            outs.append((t, h, w))
        return outs[1:]


class FPN(_TracksStorage):
    """detectron2 FPN (sum fusion, no norm) on res3..res5 + LastLevelP6P7(in_feature='res5')."""

    def __init__(self, in_channels=(512, 1024, 2048), out_channels=256):
        super().__init__()
        self.lateral = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels)
        self.output = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels)
        self.p6 = nn.Conv2d(in_channels[-1], out_channels, 3, stride=2, padding=1)
        self.p7 = nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=1)
        for m in list(self.lateral) + list(self.output) + [self.p6, self.p7]:
            nn.init.kaiming_uniform_(m.weight, a=1)      # c2_xavier_fill
            nn.init.constant_(m.bias, 0)

    def forward(self, feats):
        c3, c4, c5 = feats
        l5 = self.lateral[2](c5)
        l4 = self.lateral[1](c4) + F.interpolate(l5, size=c4.shape[-2:], mode="nearest")
        l3 = self.lateral[0](c3) + F.interpolate(l4, size=c3.shape[-2:], mode="nearest")
        out = lambda m, l: (lambda: wino_conv_nchw(m, l, relu=False) if wino_eligible(m, l) else m(l))

        def top():
            p6 = self.p6(c5)
            return p6, self.p7(F.relu(p6))
        p3, p4, p5, (p6, p7) = branches([out(self.output[0], l3), out(self.output[1], l4), out(self.output[2], l5), top], "fpn")
        return [p3, p4, p5, p6, p7]

    def cl_eligible(self) -> bool:
        return all(_c1_ok(m) for m in self.lateral) and all(_w3_ok(m) for m in self.output)

    def forward_cl(self, feats):
        """feats: [(c3, h, w), (c4, h, w), (c5, h, w)] channels-last buffers.  Lateral 1x1 convs on pod_conv1x1_split, top-down sums on
        channels-last views, output 3x3 convs on pod_wino_conv3x3[_split]; p6 / p7 (stride-2 3x3) as a patch matrix on pod_conv1x1_split (round 5; POD_HIP_P6P7=0: MIOpen on channels_last views).
        Returns (1, 256, h, w) tensors with channels_last strides: the head lays them out channels-last anyway."""
        (c3, h3, w3), (c4, h4, w4), (c5, h5, w5) = feats
        l5 = c1_of(self.lateral[2])(c5, h5, w5)
        def lateral(conv, c, h, w, top, ht, wt):               # lateral + nearest-upsampled top-down map in one store
            if FUSED_TOPDOWN and (ht, wt) == ((h + 1) // 2, (w + 1) // 2):      # a factor of two (every ResNet stage): read at (y >> 1, x >> 1), never materialised
                return c1_of(conv)(c, h, w, residual=top, residual_up2=True)
            return c1_of(conv)(c, h, w, residual=nchw_as_cl(F.interpolate(cl_as_nchw(top, ht, wt), size=(h, w), mode="nearest")))
        l4 = lateral(self.lateral[1], c4, h4, w4, l5, h5, w5)
        l3 = lateral(self.lateral[0], c3, h3, w3, l4, h4, w4)
        from . import amax
        from .conv1x1 import Conv3x3S2
        if HIP_P6P7 and Conv3x3S2.eligible(self.p6) and Conv3x3S2.eligible(self.p7) and self.output[0].out_channels == self.p6.out_channels == self.p7.out_channels:
            # the five levels go straight into ONE buffer, level after level -- the layout the head's grouped launches read (no concatenation
            # in front of the head) -- and max their abs-max into one record (the head's `in_amax`)
            (h6, w6) = Conv3x3S2.out_hw(h5, w5)
            (h7, w7) = Conv3x3S2.out_hw(h6, w6)
            hw = [(h3, w3), (h4, w4), (h5, w5), (h6, w6), (h7, w7)]
            buf = torch.empty((sum(h * w for h, w in hw), self.p6.out_channels), dtype=torch.float32, device=c5.device)
            rec, parts, at = amax.word(buf.device), [], 0
            for h, w in hw:
                parts.append(buf[at:at + h * w])
                parts[-1]._pod_amax_shared = rec             # (amax.produced: the launch that writes this slice max'es into the shared record)
                at += h * w
            wino_cl(self.output[0], l3, h3, w3, relu=False, out=parts[0])
            wino_cl(self.output[1], l4, h4, w4, relu=False, out=parts[1])
            wino_cl(self.output[2], l5, h5, w5, relu=False, out=parts[2])
            s2_of(self.p6)(c5, h5, w5, out=parts[3])
            s2_of(self.p7)(parts[3], h6, w6, relu_input=True, out=parts[4])
            amax.attach(buf, rec)
            outs = [cl_as_nchw(t, h, w) for t, (h, w) in zip(parts, hw)]
            outs[0]._pod_cl_levels, outs[0]._pod_cl_version = buf, buf._version
            return outs
        p3 = cl_as_nchw(wino_cl(self.output[0], l3, h3, w3, relu=False), h3, w3)
        p4 = cl_as_nchw(wino_cl(self.output[1], l4, h4, w4, relu=False), h4, w4)
        p5 = cl_as_nchw(wino_cl(self.output[2], l5, h5, w5, relu=False), h5, w5)
        p6 = self.p6(cl_as_nchw(c5, h5, w5))
        p7 = self.p7(F.relu(p6))
        return [p3, p4, p5, p6, p7]


class ProbabilisticRetinaNetHead(_TracksStorage):
    """PR:365-537."""

    def __init__(self, in_channels=256, num_anchors=9, num_classes=7, num_convs=4, prior_prob=0.01,
                 dropout_rate=0.0, compute_cls_var=False, compute_bbox_cov=False, bbox_cov_dims=4):
        super().__init__()
        self.num_anchors, self.num_classes = num_anchors, num_classes
        self.dropout_rate = float(dropout_rate)
        self.fused_relu_dropout = True      # GPU only; falls back to torch ops on CPU tensors
        self.dropout_seed = 0x0D50ED
        self._drop_calls = 0                # distinct Philox counter block per call
        # parity mode (the head's analogue of the hot path's eps replay): a callable (subnet 0 = cls / 1 = bbox, layer, level,
        # copy) -> bool keep-mask (C, H, W) that replaces the Philox dropout masks, so that a head evaluation can be held to
        # the reference's `nn.Dropout` on recorded masks (tests/test_head_reference*.py); None in production
        self.dropout_replay = None
        # device word folded into the Philox key of every dropout mask of the Winograd head path (include/pod_mi355x.h: `epoch`): a
        # forward replayed from a HIP graph bumps it (the launch arguments seed / offset are constants of a captured launch)
        self.register_buffer("_epoch", torch.zeros(1, dtype=torch.int64), persistent=False)
        self._sparse_pool = {}
        self.compute_cls_var, self.compute_bbox_cov, self.bbox_cov_dims = compute_cls_var, compute_bbox_cov, bbox_cov_dims
        self.cls_subnet = nn.ModuleList(nn.Conv2d(in_channels, in_channels, 3, padding=1) for _ in range(num_convs))
        self.bbox_subnet = nn.ModuleList(nn.Conv2d(in_channels, in_channels, 3, padding=1) for _ in range(num_convs))
        self.cls_score = nn.Conv2d(in_channels, num_anchors * num_classes, 3, padding=1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 3, padding=1)
        for m in list(self.cls_subnet) + list(self.bbox_subnet) + [self.cls_score, self.bbox_pred]:
            nn.init.normal_(m.weight, mean=0, std=0.01)
            nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_score.bias, -math.log((1 - prior_prob) / prior_prob))      # PR:454-455
        self.cls_var = self.bbox_cov = None
        if compute_cls_var:                                                                    # PR:458-470
            self.cls_var = nn.Conv2d(in_channels, num_anchors * num_classes, 3, padding=1)
            nn.init.normal_(self.cls_var.weight, mean=0, std=0.01)
            nn.init.constant_(self.cls_var.bias, -10.0)
        if compute_bbox_cov:                                                                   # PR:473-484
            self.bbox_cov = nn.Conv2d(in_channels, num_anchors * bbox_cov_dims, 3, padding=1)
            nn.init.normal_(self.bbox_cov.weight, mean=0, std=0.0001)
            nn.init.constant_(self.bbox_cov.bias, 0)

    def _replayed(self, x: torch.Tensor, sid: int, layer: int, level: int) -> torch.Tensor:
        """x (copies, C, H, W), any memory format, times the replayed keep-masks / (1 - p) (torch's dropout arithmetic)."""
        keep = torch.stack([self.dropout_replay(sid, layer, level, c) for c in range(x.shape[0])]).to(x.device)
        return x * (keep.to(x.dtype) / (1.0 - self.dropout_rate))

    def _trunk(self, convs, feature, copies: int, dropout: bool, level: int = 0):
        """`copies` independent evaluations of a subnet, batched on dim 0.  The first conv+ReLU is
        identical across copies (dropout only follows it) and is computed once.

        On the GPU the trunk of a large map runs channels-last: MIOpen's fp32 implicit-GEMM kernel for these shapes is an
        NHWC kernel and otherwise transposes its input and output on every call (2 x 120 us per conv on p3, 19 copies)."""
        fused = self.fused_relu_dropout and feature.is_cuda and feature.dtype == torch.float32 and FUSE_CONV_TAIL
        nhwc = fused and dropout and feature.shape[-2] * feature.shape[-1] >= NHWC_TRUNK_MIN_CELLS
        if nhwc:
            feature = feature.contiguous(memory_format=torch.channels_last)
        x = conv_bias_act(convs[0], feature, relu=True)
        if not dropout:
            for conv in convs[1:]:
                x = conv_bias_act(conv, x, relu=True)
            return x                                  # batch 1; shared by every copy
        sid = 0 if convs is self.cls_subnet else 1
        if self.dropout_replay is not None:
            x = self._replayed(x.expand(copies, -1, -1, -1), sid, 0, level)
            for j, conv in enumerate(convs[1:], 1):
                x = self._replayed(conv_bias_act(conv, x, relu=True) if fused else F.relu(conv(x)), sid, j, level)
            return x.contiguous()
        if fused and x.numel() % 4 == 0:
            from . import hip
            src = x if (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)) else x.contiguous()
            x = torch.empty((copies,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device,
                            memory_format=torch.channels_last if (nhwc and not src.is_contiguous()) else torch.contiguous_format)
            self._drop_calls += 1
            hip.check(hip.load().pod_expand_dropout(src.data_ptr(), x.data_ptr(), src.numel(), copies, float(self.dropout_rate),
                                                    self.dropout_seed, self._drop_calls << 34, None, hip.current_stream()), "pod_expand_dropout")
        else:
            x = F.dropout(x.expand(copies, -1, -1, -1), self.dropout_rate, training=True)
        for conv in convs[1:]:
            if fused:
                self._drop_calls += 1                 # distinct Philox counter block per call
                x = conv_bias_act(conv, x, relu=True, dropout_p=self.dropout_rate, seed=self.dropout_seed,
                                  offset=self._drop_calls << 34, out_nchw=conv is convs[-1])
            else:
                x = self._relu_dropout(conv(x))
        return x

    def _wino(self, conv: nn.Conv2d):
        return wino_of(conv)

    def _trunk_all_levels(self, convs, x0: torch.Tensor, levels, copies: int, dropout: bool, live=None, bufs=None):
        """`copies` evaluations of a subnet on ALL levels: one pod_wino_conv3x3 launch per conv layer (fp32 Winograd on the
        matrix cores, bias + ReLU + dropout in its store) instead of one MIOpen call + one element-wise pass per level and
        layer.  x0: (pixels of all levels, C) channels-last, level after level.  Returns (buffer, images per level): the last
        activation as (pixels of all levels x images, C) channels-last, level after level -- `copies` images per level with
        dropout, one without (every copy would be identical).
        live (sparse bbox tower, pod_compare_amd/sparse.py): a LiveBlocks -- layer j is launched over the blocks of reach L - j only;
        bufs: `new(key, shape)` handing out the tower's re-used buffers (only live blocks are written; what dead blocks hold is never read)."""
        from . import hip
        from .sparse import DENSE_INPUT, reach_of_subnet_layer
        from .wino import block_table, level_pixel_offsets
        lib, C, L = hip.load(), x0.shape[1], len(convs)
        t1 = block_table(levels, 1, x0.device)
        off1 = level_pixel_offsets(levels, 1)
        first = self._wino(convs[0])
        # (keyword only when sparse: the convolution objects of the dense path -- a test's fp64 double among them -- need not know it)
        # (layer 0 reads the FPN features, which are dense; layer j > 0 reads what layer j - 1 computed for THIS image and nothing else)
        lv = (lambda table, layer: {}) if live is None else (
            lambda table, layer: {"live": live(table, reach_of_subnet_layer(layer, L), DENSE_INPUT if layer == 0 else -1)})
        new = (lambda key, shape: torch.empty(shape, dtype=x0.dtype, device=x0.device)) if bufs is None else bufs
        if not dropout:
            y = first(x0, new("t0", tuple(x0.shape)), t1, relu=True, **lv(t1, 0))
            for j, conv in enumerate(convs[1:], 1):
                y = self._wino(conv)(y, new("t%d" % (j & 1), tuple(y.shape)), t1, relu=True, **lv(t1, j))
            return y, 1
        offn = level_pixel_offsets(levels, copies)
        a = new("a", (offn[-1], C))
        replay = self.dropout_replay is not None
        sid = 0 if convs is self.cls_subnet else 1
        # The first activation is identical for every copy: computed once, stored `copies` times under the copies' dropout masks.  One
        # Philox offset for the whole buffer: the mask of an element is keyed by its index in `a`.
        self._drop_calls += 1
        p_first = 0.0 if replay else float(self.dropout_rate)
        if first.split and copies <= 127 and FUSED_REPLICAS:
            # ... by the conv's own store pass (pod_wino_conv3x3_split_replicas)
            tr = block_table(levels, 1, x0.device, out_copies=copies)
            first.replicas(x0, a, tr, copies, relu=True, dropout_p=p_first, seed=self.dropout_seed, offset=self._drop_calls << 34, epoch=self._epoch,
                           **lv(tr, 0))
        else:
            assert live is None, "the sparse tower needs the split kernel's replica store"
            # ... or by a pass of its own per level (the fp32-MFMA kernel; the same masks)
            y = first(x0, torch.empty_like(x0), t1, relu=True)
            for i, (h, w) in enumerate(levels):
                hip.check(lib.pod_expand_dropout(y[off1[i]:].data_ptr(), a[offn[i]:].data_ptr(), h * w * C, copies, p_first, self.dropout_seed,
                                                 (self._drop_calls << 34) + offn[i] * C // 8, self._epoch.data_ptr(), hip.current_stream()),
                          "pod_expand_dropout")

        def mask_in_place(buf, layer):              # parity mode: the recorded masks on the channels-last images of the buffer
            for i, (h, w) in enumerate(levels):
                v = buf[offn[i]:offn[i + 1]].view(copies, h, w, C)
                v.copy_(self._replayed(v.permute(0, 3, 1, 2), sid, layer, i).permute(0, 2, 3, 1))

        if replay:
            mask_in_place(a, 0)
        tn = block_table(levels, copies, x0.device)
        b = new("b", tuple(a.shape))
        for j, conv in enumerate(convs[1:], 1):
            self._drop_calls += 1
            self._wino(conv)(a, b, tn, relu=True, dropout_p=0.0 if replay else self.dropout_rate, seed=self.dropout_seed,
                             offset=self._drop_calls << 34, epoch=self._epoch, **lv(tn, j))
            if replay:
                mask_in_place(b, j)
            a, b = b, a
        return a, copies

    def _grouped_ok(self, *copies) -> bool:
        """Both subnets' launches in one grid (pod_wino_conv3x3_split_grouped): the split kernel, the replicas in the store pass."""
        return (GROUPED_HEAD and FUSED_REPLICAS and not BRANCHES and all(1 <= c <= 127 for c in copies)
                and all(self._wino(c).split for c in list(self.cls_subnet) + list(self.bbox_subnet)))

    def _trunks_grouped(self, x0: torch.Tensor, levels, copies_c: int, copies_b: int, dropout: bool):
        """`_trunk_all_levels` of the cls and the bbox subnet with layer l of both in ONE launch (the kernel runs one workgroup per CU, so
        a launch costs whole rounds of 256 workgroups: two launches of 1.5 rounds cost 4, one of 3.0 costs 3).  Same buffers, tables,
        Philox offsets and therefore the same bits as the two separate trunks.  Returns (cls buffer, images, bbox buffer, images)."""
        from .wino import block_table, grouped_launch, level_pixel_offsets
        subs, L, C, dev = (self.cls_subnet, self.bbox_subnet), len(self.cls_subnet), x0.shape[1], x0.device
        if not dropout:
            ys = [x0, x0]
            t1 = block_table(levels, 1, dev)
            for l in range(L):
                outs = [torch.empty_like(x0), torch.empty_like(x0)]
                grouped_launch([{"conv": self._wino(subs[i][l]), "src": ys[i], "dst": outs[i], "table": t1} for i in range(2)], relu=True)
                ys = outs
            return ys[0], 1, ys[1], 1
        copies = (copies_c, copies_b)
        replay = self.dropout_replay is not None
        p = 0.0 if replay else float(self.dropout_rate)
        ids = [[self._drop_calls + i * L + l + 1 for l in range(L)] for i in range(2)]       # the offsets the two separate trunks would draw
        self._drop_calls += 2 * L
        offn = [level_pixel_offsets(levels, c) for c in copies]
        a = [torch.empty((offn[i][-1], C), dtype=x0.dtype, device=dev) for i in range(2)]

        def mask_in_place(bufs, layer):             # parity mode: the recorded masks on the channels-last images of the buffers
            for i in range(2):
                for lv, (h, w) in enumerate(levels):
                    v = bufs[i][offn[i][lv]:offn[i][lv + 1]].view(copies[i], h, w, C)
                    v.copy_(self._replayed(v.permute(0, 3, 1, 2), i, layer, lv).permute(0, 2, 3, 1))

        grouped_launch([{"conv": self._wino(subs[i][0]), "src": x0, "dst": a[i], "table": block_table(levels, 1, dev, out_copies=copies[i]),
                         "offset": ids[i][0] << 34, "replicas": copies[i]} for i in range(2)],
                       relu=True, dropout_p=p, seed=self.dropout_seed, epoch=self._epoch)
        if replay:
            mask_in_place(a, 0)
        tn = [block_table(levels, c, dev) for c in copies]
        b = [torch.empty_like(t) for t in a]
        for l in range(1, L):
            grouped_launch([{"conv": self._wino(subs[i][l]), "src": a[i], "dst": b[i], "table": tn[i], "offset": ids[i][l] << 34} for i in range(2)],
                           relu=True, dropout_p=p, seed=self.dropout_seed, epoch=self._epoch)
            if replay:
                mask_in_place(b, l)
            a, b = b, a
        return a[0], copies_c, a[1], copies_b

    def _predict_grouped(self, jobs, levels, live=None, bufs=None):
        """The predictor convs in ONE launch.  jobs: (conv, buffer, images in the buffer per level, first image, image count, output images)
        as `_predict_all_levels` takes them; returns its result for each job.  live / bufs: the sparse bbox tower (`_trunk_all_levels`)."""
        from .sparse import REACH_PREDICTOR
        from .wino import block_table, grouped_launch, level_pixel_offsets
        sets, outs = [], []
        for ji, (conv, buf, buf_copies, first, count, out_copies) in enumerate(jobs):
            K = conv.out_channels
            offs = level_pixel_offsets(levels, out_copies)
            if bufs is not None:
                out = bufs("p%d" % ji, (offs[-1] * K,))
            else:
                out = (torch.zeros if out_copies > count else torch.empty)(offs[-1] * K, dtype=buf.dtype, device=buf.device)
            sets.append({"conv": self._wino(conv), "src": buf, "dst": out, "planes": True,
                         "table": block_table(levels, count, buf.device, in_copies=buf_copies, in_first=first, out_copies=out_copies)})
            outs.append([out[offs[i] * K:offs[i + 1] * K].view(out_copies, K, h, w) for i, (h, w) in enumerate(levels)])
        grouped_launch(sets, live=None if live is None else (lambda table: live(table, REACH_PREDICTOR)))
        return outs

    def _predict_all_levels(self, conv, buf: torch.Tensor, levels, buf_copies: int, first: int, count: int, out_copies: int):
        """A predictor conv (cls_score / bbox_pred / cls_var / bbox_cov, PR:430-484) on images first .. first+count-1 of every level
        of a trunk buffer, one launch; returns per level an (out_copies, K, H, W) NCHW tensor -- the planes K1 streams -- whose
        images past `count` are zero (never uninitialised memory: a consumer without the quirk merge would read them)."""
        from .wino import block_table, level_pixel_offsets
        K = conv.out_channels
        offs = level_pixel_offsets(levels, out_copies)
        out = (torch.zeros if out_copies > count else torch.empty)(offs[-1] * K, dtype=buf.dtype, device=buf.device)
        table = block_table(levels, count, buf.device, in_copies=buf_copies, in_first=first, out_copies=out_copies)
        self._wino(conv)(buf, out, table, planes=True)
        return [out[offs[i] * K:offs[i + 1] * K].view(out_copies, K, h, w) for i, (h, w) in enumerate(levels)]

    def takes_wino_path(self) -> bool:
        """Every conv of the head runs on pod_wino_conv3x3[_split] (GPU, fp32, one image): then all its dropout masks take the `_epoch` word."""
        return (WINO_HEAD and self.fused_relu_dropout and FUSE_CONV_TAIL and self.cls_subnet[0].in_channels % 8 == 0
                and self.cls_subnet[0].out_channels in (64, 128, 256, 512)
                and max(c.out_channels for c in (self.cls_score, self.bbox_pred, self.cls_var, self.bbox_cov) if c is not None) <= 512)

    def _relu_dropout(self, x: torch.Tensor) -> torch.Tensor:
        """ReLU + Dropout(p) after a subnet conv (PR:403-424).  On the GPU one fused in-place HIP pass
        (pod_relu_dropout) instead of torch's clamp + fused_dropout kernels."""
        if not (self.fused_relu_dropout and x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
            return F.dropout(F.relu(x), self.dropout_rate, training=True)
        from . import hip
        lib = hip.load()
        self._drop_calls += 1
        hip.check(lib.pod_relu_dropout(x.data_ptr(), x.numel(), float(self.dropout_rate), self.dropout_seed,
                                       self._drop_calls << 34, hip.current_stream()), "pod_relu_dropout")
        return x

    def sparse_buffers(self, key):
        """Re-used activation buffers of the sparse bbox tower, per (stream, geometry): new(name, shape)."""
        pool = self._sparse_pool.setdefault(key, {})

        def new(name, shape):
            t = pool.get(name)
            if t is None or tuple(t.shape) != tuple(shape):
                # zeroed once (a reader of the DENSE tensors finds finite numbers outside the live blocks).  The tower itself never reads
                # what an earlier image left here: a sparse launch reads the cells the layer below did not compute for this image as 0.0
                # (the need bits of sparse.LiveBlocks), and every launch gets a fresh abs-max record -- results depend on the image alone
                t = pool[name] = torch.zeros(shape, dtype=torch.float32, device=self.cls_score.weight.device)
            return t
        return new

    def forward_cls(self, features: List[torch.Tensor], num_runs: int = 1, mc_dropout: bool = False, skip_unused_last_run: bool = False) -> dict:
        """First half of the sparse evaluation (`forward`, sparse_bbox): the cls subnet + cls_score (+ cls_var) of all levels and runs.
        Returns the state `forward_bbox` continues from (the level features as one channels-last buffer among it)."""
        dropout = mc_dropout and self.dropout_rate > 0.0
        n = num_runs
        skip = 1 if (skip_unused_last_run and dropout and n > 1) else 0
        m = n - skip
        cls_copies = m * (2 if self.compute_cls_var else 1)
        box_copies = n + (m if self.compute_bbox_cov else 0)
        ok = (self.takes_wino_path() and features[0].is_cuda and features[0].dtype == torch.float32 and features[0].shape[0] == 1
              and features[0].shape[1] == self.cls_subnet[0].in_channels and self._grouped_ok(cls_copies, box_copies))
        if not ok:
            raise RuntimeError("the sparse bbox tower needs the split kernels on the GPU (POD_WINO_SPLIT=1, one fp32 image, grouped head)")
        levels = [(int(f.shape[2]), int(f.shape[3])) for f in features]
        x0 = levels_channels_last(features)                                                    # channels-last, level after level
        tc, nc = self._trunk_all_levels(self.cls_subnet, x0, levels, cls_copies, dropout)
        cj = [(self.cls_score, tc, nc, 0, m, n)] if dropout else [(self.cls_score, tc, 1, 0, 1, 1)]
        if self.compute_cls_var:
            cj += [(self.cls_var, tc, nc, m, m, n)] if dropout else [(self.cls_var, tc, 1, 0, 1, 1)]
        res = self._predict_grouped(cj, levels)
        if not dropout and n > 1:
            res = [[t.expand(n, -1, -1, -1).contiguous() for t in ts] for ts in res]
        return {"x0": x0, "levels": levels, "n": n, "m": m, "dropout": dropout, "skip": skip, "box_copies": box_copies,
                "logits": res[0], "logit_vars": res[1] if self.compute_cls_var else None}

    def forward_bbox(self, st: dict, live):
        """Second half: bbox_subnet + bbox_pred (+ bbox_cov) over the blocks `live` (sparse.LiveBlocks) lists; None = all of them."""
        x0, levels, n, m, dropout = st["x0"], st["levels"], st["n"], st["m"], st["dropout"]
        bufs = None if live is None else self.sparse_buffers((torch.cuda.current_stream(x0.device).cuda_stream, tuple(levels), n, dropout, st["skip"]))
        tb, nb = self._trunk_all_levels(self.bbox_subnet, x0, levels, st["box_copies"], dropout, live=live, bufs=bufs)
        bj = [(self.bbox_pred, tb, nb, 0, n, n)] if dropout else [(self.bbox_pred, tb, 1, 0, 1, 1)]
        if self.compute_bbox_cov:
            bj += [(self.bbox_cov, tb, nb, n, m, n)] if dropout else [(self.bbox_cov, tb, 1, 0, 1, 1)]
        res = self._predict_grouped(bj, levels, live=live, bufs=bufs)
        if not dropout and n > 1:
            res = [[t.expand(n, -1, -1, -1).contiguous() for t in ts] for ts in res]
        return res[0], (res[1] if self.compute_bbox_cov else None)

    def forward(self, features: List[torch.Tensor], num_runs: int = 1, mc_dropout: bool = False,
                skip_unused_last_run: bool = False, sparse_bbox=None):
        """features: per-level (1, 256, H, W).  Returns per-level lists of (num_runs, A*C, H, W).

        skip_unused_last_run: the reference's merge (PI:216-222) never reads run N-1 of box_cls,
        box_cls_var and box_reg_var (only box_delta's last run is used, by the epistemic covariance PI:325-331).
        With the flag set those three evaluations of the last run are not computed (their slab in the returned
        tensors is zero): 3 of the 4N subnet evaluations, 7.5 % of the head at N = 10.  Only valid
        together with `merge_quirk=True` in the hot path.

        sparse_bbox (round 5): None, or a callable (logits, logit_vars) -> sparse.LiveBlocks.  The cls side is then evaluated FIRST, the
        callable selects the image's candidates from it (K1f + K2 of the hot path, PI:283-308) and the bbox side -- bbox_subnet,
        bbox_pred, bbox_cov: PR:518-537 evaluates them densely although PI:310-331 reads them at the candidates only -- runs over the
        blocks that can reach a candidate.  Its outputs are defined at the candidates' cells and stale elsewhere; same Philox offsets, so
        the same dropout masks as the dense evaluation."""
        dropout = mc_dropout and self.dropout_rate > 0.0
        n = num_runs
        skip = 1 if (skip_unused_last_run and dropout and n > 1) else 0
        m = n - skip                                               # runs whose cls / cls_var / reg_var are needed

        def padded(t):                                            # (m, C, H, W) -> (n, C, H, W) NCHW planes (what K1 streams),
            if m == n:                                            # last slab untouched; a channels-last trunk output is
                return t.contiguous()                             # transposed by the same copy
            out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            out[:m].copy_(t)
            out[m:].zero_()                                       # never hand out uninitialised memory (a consumer without
            return out                                            # the quirk merge would read it); 1/n of one small tensor

        logits, deltas, logit_vars, delta_covs = [], [], [], []
        cls_copies = m * (2 if self.compute_cls_var else 1)
        box_copies = n + (m if self.compute_bbox_cov else 0)
        wino = (self.takes_wino_path() and features[0].is_cuda and features[0].dtype == torch.float32 and features[0].shape[0] == 1
                and features[0].shape[1] == self.cls_subnet[0].in_channels)
        if wino:
            # every conv of the head on pod_wino_conv3x3: one launch per layer over all levels and all runs
            levels = [(int(f.shape[2]), int(f.shape[3])) for f in features]
            x0 = levels_channels_last(features)                                                    # channels-last, level after level
            grouped = self._grouped_ok(cls_copies, box_copies)
            if sparse_bbox is not None:
                st = self.forward_cls(features, num_runs, mc_dropout, skip_unused_last_run)
                deltas, delta_covs = self.forward_bbox(st, sparse_bbox(st["logits"], st["logit_vars"]))
                return st["logits"], deltas, st["logit_vars"], delta_covs
            if grouped:
                tc, nc, tb, nb = self._trunks_grouped(x0, levels, cls_copies, box_copies, dropout)
            else:
                (tc, nc), (tb, nb) = branches([lambda: self._trunk_all_levels(self.cls_subnet, x0, levels, cls_copies, dropout),
                                               lambda: self._trunk_all_levels(self.bbox_subnet, x0, levels, box_copies, dropout)], "head")
            preds = [c for c in (self.cls_score, self.bbox_pred, self.cls_var if self.compute_cls_var else None, self.bbox_cov if self.compute_bbox_cov else None)
                     if c is not None]
            if grouped and len({self._wino(c).Kpad for c in preds}) == 1 and all(self._wino(c).split for c in preds):
                # the predictors in one launch too
                if dropout:
                    jobs = [(self.cls_score, tc, nc, 0, m, n), (self.bbox_pred, tb, nb, 0, n, n)]
                    jobs += [(self.cls_var, tc, nc, m, m, n)] if self.compute_cls_var else []          # independent dropout draw (Q2)
                    jobs += [(self.bbox_cov, tb, nb, n, m, n)] if self.compute_bbox_cov else []
                    res = self._predict_grouped(jobs, levels)
                else:
                    jobs = [(self.cls_score, tc, 1, 0, 1, 1), (self.bbox_pred, tb, 1, 0, 1, 1)]
                    jobs += [(self.cls_var, tc, 1, 0, 1, 1)] if self.compute_cls_var else []
                    jobs += [(self.bbox_cov, tb, 1, 0, 1, 1)] if self.compute_bbox_cov else []
                    res = self._predict_grouped(jobs, levels)
                    if n > 1:
                        res = [[t.expand(n, -1, -1, -1).contiguous() for t in ts] for ts in res]
                logits, deltas = res[0], res[1]
                logit_vars = res[2] if self.compute_cls_var else []
                delta_covs = res[-1] if self.compute_bbox_cov else []
            elif dropout:
                logits = self._predict_all_levels(self.cls_score, tc, levels, nc, 0, m, n)
                deltas = self._predict_all_levels(self.bbox_pred, tb, levels, nb, 0, n, n)
                if self.compute_cls_var:
                    logit_vars = self._predict_all_levels(self.cls_var, tc, levels, nc, m, m, n)   # independent dropout draw (Q2)
                if self.compute_bbox_cov:
                    delta_covs = self._predict_all_levels(self.bbox_cov, tb, levels, nb, n, m, n)
            else:
                ex = (lambda ts: [t.expand(n, -1, -1, -1).contiguous() for t in ts]) if n > 1 else (lambda ts: ts)
                pred = lambda conv, buf: (lambda: None if conv is None else ex(self._predict_all_levels(conv, buf, levels, 1, 0, 1, 1)))
                logits, deltas, logit_vars, delta_covs = branches([pred(self.cls_score, tc), pred(self.bbox_pred, tb),
                                                                   pred(self.cls_var if self.compute_cls_var else None, tc),
                                                                   pred(self.bbox_cov if self.compute_bbox_cov else None, tb)], "pred")
            return logits, deltas, (logit_vars if self.compute_cls_var else None), (delta_covs if self.compute_bbox_cov else None)
        for level, f in enumerate(features):
            tc = self._trunk(self.cls_subnet, f, cls_copies, dropout, level)
            tb = self._trunk(self.bbox_subnet, f, box_copies, dropout, level)
            if dropout:
                # (a channels-last trunk hands its last activation over as NCHW planes: the A*K / A*4-channel predictor convs
                #  are faster as NCHW Winograd calls than as NHWC implicit GEMMs)
                logits.append(padded(conv_bias_act(self.cls_score, tc[:m])))
                deltas.append(conv_bias_act(self.bbox_pred, tb[:n]).contiguous())
                if self.compute_cls_var:
                    logit_vars.append(padded(conv_bias_act(self.cls_var, tc[m:])))     # independent dropout draw (Q2)
                if self.compute_bbox_cov:
                    delta_covs.append(padded(conv_bias_act(self.bbox_cov, tb[n:])))
            else:
                ex = (lambda t: t.expand(n, -1, -1, -1).contiguous()) if n > 1 else (lambda t: t)
                logits.append(ex(self.cls_score(tc)))
                deltas.append(ex(self.bbox_pred(tb)))
                if self.compute_cls_var:
                    logit_vars.append(ex(self.cls_var(tc)))
                if self.compute_bbox_cov:
                    delta_covs.append(ex(self.bbox_cov(tb)))
        return logits, deltas, (logit_vars if self.compute_cls_var else None), (delta_covs if self.compute_bbox_cov else None)


class ProbabilisticRetinaNet(_TracksStorage):
    """PR:20-166 (inference side only; `losses` PR:168-333 is training and out of scope)."""

    def __init__(self, num_classes=7, dropout_rate=0.0, cls_var_loss="none", cls_var_num_samples=3,
                 bbox_cov_loss="none", bbox_cov_type="diagonal", test_score_thresh=0.05, test_topk_candidates=1000,
                 test_nms_thresh=0.5, max_detections_per_image=100, min_size_test=800, max_size_test=1333):
        super().__init__()
        self.num_classes = num_classes
        self.compute_cls_var = cls_var_loss != "none"                  # PR:29-30
        self.cls_var_num_samples = cls_var_num_samples
        self.compute_bbox_cov = bbox_cov_loss != "none"                # PR:33-34
        self.bbox_cov_dims = 4 if bbox_cov_type == "diagonal" else 10  # PR:37-44
        self.dropout_rate = dropout_rate
        self.use_dropout = dropout_rate != 0.0
        self.test_score_thresh, self.test_topk_candidates = test_score_thresh, test_topk_candidates
        self.test_nms_thresh, self.max_detections_per_image = test_nms_thresh, max_detections_per_image
        self.min_size_test, self.max_size_test = min_size_test, max_size_test
        self.input_format = "BGR"
        self.in_features = ["p3", "p4", "p5", "p6", "p7"]
        self.num_anchors = len(_anchors.ANCHOR_SIZES[0]) * len(_anchors.ASPECT_RATIOS)
        self.bottom_up = ResNet50()
        self.fpn = FPN()
        self.head = ProbabilisticRetinaNetHead(256, self.num_anchors, num_classes, 4, 0.01, dropout_rate,
                                               self.compute_cls_var, self.compute_bbox_cov, self.bbox_cov_dims)
        self.register_buffer("pixel_mean", torch.tensor(PIXEL_MEAN_BGR).view(3, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(PIXEL_STD).view(3, 1, 1), persistent=False)
        self._anchor_cache: Dict[Tuple[int, int], List[torch.Tensor]] = {}
        self.use_graphs = False
        self.max_graphs = 16                                   # (stream, frame shape, flags) entries kept (four frame shapes on each of apply_net's four streams); each owns its activations
        self.graph_after_seen = 0                              # forwards of a (stream, shape, flags) run eagerly before it is captured
        self._graphs: Dict[tuple, tuple] = {}
        self._graph_seen: Dict[tuple, int] = {}
        self._graph_serial = 0
        self._fingerprint_tensors = None
        self._graphs_fingerprint = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: bump_param_generation())

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, image: torch.Tensor) -> torch.Tensor:
        """(3,H,W) BGR uint8/float -> normalised, zero-padded (1,3,H',W') with H',W' % 32 == 0 (PR:96)."""
        x = (image.to(self.device, torch.float32) - self.pixel_mean) / self.pixel_std
        h, w = x.shape[-2:]
        ph, pw = _anchors.padded_size(h, w)
        return F.pad(x, (0, pw - w, 0, ph - h)).unsqueeze(0)

    def anchors_for(self, padded_hw: Tuple[int, int]) -> List[torch.Tensor]:
        if padded_hw not in self._anchor_cache:
            shapes = _anchors.level_shapes(*padded_hw)
            self._anchor_cache[padded_hw] = _anchors.grid_anchors(shapes, device=self.device)
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()   # made once, then read from any stream
        return self._anchor_cache[padded_hw]

    # ---- HIP graphs --------------------------------------------------------------------------------------------------------
    # One image's forward is ~200 launches (MIOpen calls, pod_* kernels, torch element-wise ops) issued from Python: 2.3 - 2.5 ms of
    # host time, which is the whole step of the single-run configurations (BASELINE configs[1], [3]: 2.5 ms of GPU time).  With
    # `enable_graphs()` the forward of a given (stream, frame shape, flags) is captured once into a HIP graph and replayed: one
    # host call per image.  MC-dropout forwards too, when the head runs on the Winograd kernels: their masks' Philox key folds in a
    # device word (`head._epoch`) that the captured forward itself bumps first thing, so every replay draws fresh masks although
    # seed / offset are constants of the captured launches.  Not captured: a dropout forward on the MIOpen head path, anything off
    # the GPU, the parity mode (`dropout_replay`).  The returned tensors belong to the graph: they are valid until the next forward
    # of the same (stream, shape, flags) -- on the same stream, so a consumer enqueued there before that is safe.
    def enable_graphs(self, on: bool = True) -> "ProbabilisticRetinaNet":
        self.use_graphs = bool(on)
        if not on:
            self._drop_graphs()
        return self

    def _drop_graphs(self):
        if self._graphs:
            torch.cuda.synchronize(self.device)          # (a consumer of a graph's tensors may still be queued)
            self._graphs.clear()

    # A captured graph holds raw pointers to the pre-transformed filters (WinoConv.U, Conv1x1.Ws, Stem7x7.Ws) and folded biases of the
    # moment of capture: a graph must never answer for parameters that changed since.  Two things change them: a move / re-allocation
    # (.to(), .cuda(), load_state_dict(assign=True), fold_frozen_bn: all go through _apply / the load hook / module surgery -> the
    # generation counter) and an in-place write (load_state_dict, copy_: bumps the tensors' `_version`).  Both are in the fingerprint;
    # when it moves, every graph is dropped and the next forward re-captures against re-transformed filters.
    def _param_fingerprint(self):
        gen = _PARAM_GENERATION[0]
        ts = self._fingerprint_tensors
        if ts is None or ts[0] != gen:
            ts = self._fingerprint_tensors = (gen, [t for t in list(self.parameters()) + list(self.buffers()) if t is not self.head._epoch])
        return gen, len(ts[1]), sum(t._version for t in ts[1])

    def _cls_eager(self, image: torch.Tensor, n: int, dropout: bool, skip: bool) -> dict:
        feats, padded = self._trunk_eager(image)
        st = self.head.forward_cls(feats, n, mc_dropout=dropout, skip_unused_last_run=skip)
        st.update(padded=padded, image_hw=tuple(image.shape[-2:]), shapes=[tuple(f.shape[-2:]) for f in feats])
        return st

    @torch.no_grad()
    def cls_part(self, image: torch.Tensor, num_mc_dropout_runs: int = -1, skip_unused_last_run: bool = False, mc_dropout: Optional[bool] = None) -> dict:
        """First part of a forward with the sparse bbox tower: trunk + the cls side of the head (captured into a HIP graph like a whole forward
        when graphs are enabled).  Returns the state `bbox_part` continues from; state["partial"] is the HeadOutputs with cls / cls_var set."""
        n = num_mc_dropout_runs if num_mc_dropout_runs > 1 else 1
        if mc_dropout is None:
            mc_dropout = n > 1
        dropout = bool(mc_dropout) and self.use_dropout
        graphs = (self.use_graphs and image.is_cuda and self.device.type == "cuda" and self.head.dropout_replay is None
                  and (not dropout or self.head.takes_wino_path()))
        st = self._forward_graphed(image, n, dropout, skip_unused_last_run, part="cls") if graphs else self._cls_eager(image, n, dropout, skip_unused_last_run)
        return st

    def partial_outputs(self, st: dict, delta=None, reg_var=None) -> HeadOutputs:
        return HeadOutputs(st["logits"], delta, st["logit_vars"], reg_var, self.anchors_for(st["padded"]), st["shapes"], self.num_anchors,
                           self.num_classes, st["image_hw"], last_run_valid=not bool(st["skip"]))

    @torch.no_grad()
    def bbox_part(self, st: dict, live) -> HeadOutputs:
        """Second part: the bbox side over the blocks `live` (sparse.LiveBlocks, made from the candidates the caller selected) lists."""
        delta, reg_var = self.head.forward_bbox(st, live)
        return self.partial_outputs(st, delta, reg_var)

    def _bbox_eager(self, st: dict, sparse_bbox) -> HeadOutputs:
        skipped = bool(st["skip"])
        mk = lambda delta, reg_var: HeadOutputs(st["logits"], delta, st["logit_vars"], reg_var, self.anchors_for(st["padded"]), st["shapes"], self.num_anchors,
                                                self.num_classes, st["image_hw"], last_run_valid=not skipped)
        delta, reg_var = self.head.forward_bbox(st, sparse_bbox(mk(None, None)))
        return mk(delta, reg_var)

    def _forward_graphed(self, image: torch.Tensor, n: int, dropout: bool, skip: bool, part: str = "all"):
        stream = torch.cuda.current_stream(image.device)
        run = self._forward_eager if part == "all" else self._cls_eager
        from . import wino
        fp = self._param_fingerprint()
        if fp != self._graphs_fingerprint:
            self._drop_graphs()
            self._fingerprint_tensors = None                   # (module surgery -- fold_frozen_bn -- also changes WHICH tensors there are)
            self._graphs_fingerprint = self._param_fingerprint()
        # (the kernel selection is part of the key: a graph captured with one convolution kernel must not answer for the other)
        key = (stream.cuda_stream, tuple(image.shape), image.dtype, n, dropout, skip, bool(wino.SPLIT_BF16), CL_BACKBONE, WINO_BACKBONE, part)
        ent = self._graphs.get(key)
        if ent is None:
            # graphs are for (stream, shape) pairs that come back: the first GRAPH_AFTER_SEEN forwards of a key run eagerly (a data set of
            # many frame sizes would otherwise pay two eager forwards + a capture per image -- slower than no graphs at all)
            seen = self._graph_seen.get(key, 0) + 1
            if len(self._graph_seen) > 4096:
                self._graph_seen.clear()
            self._graph_seen[key] = seen
            if seen <= self.graph_after_seen:
                return run(image, n, dropout, skip)
            static_in = image.clone()
            shared_epoch = self.head._epoch
            # every graph owns its epoch word: replays of two graphs on two streams must not read-modify-write one word (the masks
            # would depend on GPU timing).  It starts at (serial of the graph) << 32, so no two graphs ever draw the same masks and
            # the masks of the i-th replay of the j-th captured graph are a function of (seed, j, i) alone.
            self._graph_serial += 1
            epoch = torch.full_like(shared_epoch, self._graph_serial << 32)
            self.head._epoch = epoch
            from . import amax
            try:
                for _ in range(2):                       # eager: MIOpen's solver search, filter transforms, block tables, anchors
                    run(static_in, n, dropout, skip)
                stream.synchronize()
                side = torch.cuda.Stream(device=image.device)          # (capture is not allowed on the legacy default stream)
                side.wait_stream(stream)
                graph = torch.cuda.CUDAGraph()
                # thread_local: only this thread's calls are policed during the capture (an RCCL watchdog thread polling its events
                # must not abort it)
                # (the warm-up forwards attached an abs-max record to a float frame: it is the FIRST frame's, and a capture that found it
                #  would record no pod_absmax -- every replay would scale the stem's split by that frame's range.  ADVICE r5)
                amax.forget(static_in)
                amax.reset()                               # the operand abs-max words of the captured launches come from pools zeroed INSIDE the
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):      # capture: every replay starts from zeroed words
                    if dropout:
                        epoch.add_(1)                      # (captured: every replay starts by moving on to the next set of masks)
                    out = run(static_in, n, dropout, skip)
                stream.wait_stream(side)
            finally:
                amax.reset()                               # (eager launches never max into a graph's words)
                self.head._epoch = shared_epoch
            while len(self._graphs) >= self.max_graphs:          # frames of many different sizes: keep the most recent shapes only
                old_key = next(iter(self._graphs))
                torch.cuda.ExternalStream(old_key[0], device=image.device).synchronize()   # (a consumer of the evicted graph's tensors may
                self._graphs.pop(old_key)                                                  #  still be queued -- on ITS stream, nowhere else)
            ent = (graph, static_in, out, epoch)
        else:
            self._graphs.pop(key)
        self._graphs[key] = ent                                # most recently used last
        graph, static_in, out, _ = ent
        static_in.copy_(image, non_blocking=True)
        graph.replay()
        return out

    def graph_epoch(self, stream=None, **match) -> Optional[torch.Tensor]:
        """The epoch word of the most recently used graph on `stream` (default: the current one); tests and diagnostics."""
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        for key in reversed(list(self._graphs)):
            if key[0] == st and key[4]:
                return self._graphs[key][3]
        return None

    @torch.no_grad()
    def forward(self, image: torch.Tensor, num_mc_dropout_runs: int = -1, skip_unused_last_run: bool = False,
                mc_dropout: Optional[bool] = None, sparse_bbox=None) -> HeadOutputs:
        """Raw anchor-wise output (`return_anchorwise_output=True`, PR:352-361) in NCHW plane layout.
        num_mc_dropout_runs > 1 batches that many dropout-perturbed head evaluations (PR:103-108).
        mc_dropout: dropout active in the head subnets -- the reference's `model.train()` (PI:53-56), which it sets
        whenever MC_DROPOUT.ENABLE is true, also for a single run; default: active iff several runs are requested.
        sparse_bbox: callable (partial HeadOutputs: cls / cls_var set, delta = reg_var = None) -> sparse.LiveBlocks: the bbox side of the
        head is evaluated only where it can reach a candidate (ProbabilisticRetinaNetHead.forward).  The trunk + cls half is replayed as a HIP
        graph (`cls_part`), the bbox half is enqueued eagerly (its live lists are made per image)."""
        n = num_mc_dropout_runs if num_mc_dropout_runs > 1 else 1
        if mc_dropout is None:
            mc_dropout = n > 1
        dropout = bool(mc_dropout) and self.use_dropout
        if sparse_bbox is not None:
            return self._bbox_eager(self.cls_part(image, num_mc_dropout_runs, skip_unused_last_run, mc_dropout), sparse_bbox)
        if (self.use_graphs and image.is_cuda and self.device.type == "cuda" and self.head.dropout_replay is None
                and (not dropout or self.head.takes_wino_path())):
            return self._forward_graphed(image, n, dropout, skip_unused_last_run)
        return self._forward_eager(image, n, dropout, skip_unused_last_run)

    def _cl_backbone(self, x: torch.Tensor) -> bool:
        """The trunk runs channels-last on pod_conv1x1_split + pod_wino_conv3x3_split: GPU, fp32, FrozenBN folded, the split kernels
        selected (POD_WINO_SPLIT=0 -- the fp32-MFMA reference leg -- keeps the 1x1 convolutions on MIOpen as in round 3)."""
        from . import wino
        if not (CL_BACKBONE and FUSE_CONV_TAIL and WINO_BACKBONE and wino.SPLIT_BF16 and x.is_cuda and x.dtype == torch.float32):
            return False
        if not getattr(self, "_cl_ok", False):
            self._cl_ok = self.bottom_up.cl_eligible() and self.fpn.cl_eligible()       # (cached once true: folding is one-way)
        return self._cl_ok

    def _trunk_eager(self, image: torch.Tensor):
        """Frame -> (the five FPN maps, padded (h, w)): everything ahead of the head."""
        if (FUSED_PREPROCESS and image.is_cuda and image.device == self.device and image.dim() == 3 and image.shape[0] == 3 and image.is_contiguous()
                and image.dtype in (torch.uint8, torch.float32) and self._cl_backbone(self.pixel_mean) and self.bottom_up.hip_stem_ok()):
            # the frame as the loader hands it over: pod_stem7x7_split normalises ((x - mean) / std, PR:96) and pads on load
            padded = _anchors.padded_size(int(image.shape[1]), int(image.shape[2]))
            feats = self.fpn.forward_cl(self.bottom_up.forward_cl(None, frame=(image, self.pixel_mean.reshape(3), self.pixel_std.reshape(3), padded)))
        else:
            x = self.preprocess_image(image)
            if self._cl_backbone(x):
                feats = self.fpn.forward_cl(self.bottom_up.forward_cl(x))      # channels-last trunk on pod_conv1x1_split + pod_wino_conv3x3[_split]
            else:
                feats = self.fpn(self.bottom_up(x))
            padded = tuple(x.shape[-2:])
        return feats, padded

    def _head_eager(self, feats, padded, image_hw, n: int, mc_dropout: bool, skip_unused_last_run: bool, sparse_bbox=None) -> HeadOutputs:
        shapes = [tuple(f.shape[-2:]) for f in feats]
        skipped = skip_unused_last_run and n > 1 and bool(mc_dropout) and self.use_dropout
        hook = None
        if sparse_bbox is not None:
            hook = lambda logits, logit_vars: sparse_bbox(HeadOutputs(logits, None, logit_vars, None, self.anchors_for(padded), shapes, self.num_anchors,
                                                                      self.num_classes, tuple(image_hw), last_run_valid=not skipped))
        cls, delta, cls_var, reg_var = self.head(feats, n, mc_dropout=bool(mc_dropout) and self.use_dropout,
                                                 skip_unused_last_run=skip_unused_last_run, sparse_bbox=hook)
        return HeadOutputs(cls, delta, cls_var, reg_var, self.anchors_for(padded), shapes, self.num_anchors,
                           self.num_classes, tuple(image_hw), last_run_valid=not skipped)

    def _forward_eager(self, image: torch.Tensor, n: int, mc_dropout: bool, skip_unused_last_run: bool, sparse_bbox=None) -> HeadOutputs:
        feats, padded = self._trunk_eager(image)
        return self._head_eager(feats, padded, image.shape[-2:], n, mc_dropout, skip_unused_last_run, sparse_bbox)


def resize_test_image(image: torch.Tensor, min_size: int = 800, max_size: int = 1333) -> torch.Tensor:
    """detectron2 ResizeShortestEdge test transform (apply_net.py:83), bilinear, on the device."""
    h, w = image.shape[-2:]
    nh, nw = _anchors.resize_shortest_edge(h, w, min_size, max_size)
    if (nh, nw) == (h, w):
        return image
    return F.interpolate(image.unsqueeze(0).float(), size=(nh, nw), mode="bilinear", align_corners=False).squeeze(0)
