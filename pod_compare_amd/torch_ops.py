"""`torch.ops.pod_mi355x.*`: the C-ABI entry points as registered PyTorch operators (last row).

A maintainer of the reference who prefers operator calls over ctypes gets the hot path as ops that take and return
tensors, run on the CURRENT HIP stream and validate their arguments (`torch._check`: fp32 / int32, contiguous, on one
CUDA device).  The ops are thin: argument checks + the same `libpod_mi355x.so` launches `hotpath.HotPath` makes; there is
no second implementation and no CPU kernel (calling them with CPU tensors raises).

    import pod_compare_amd.torch_ops          # registers the library
    boxes, cov, scores, classes, probs = torch.ops.pod_mi355x.predict(
        box_cls, box_delta, box_cls_var, box_reg_var, anchors, "bayes_od", [750, 1333], [720, 1280], ...)

    predict      everything RetinaNetProbabilisticPredictor.__call__ does after the model forward (PI:86-111): one image's
                 dense head tensors (per level, NCHW planes `(n_runs, A*C, H, W)`, the conv head's own layout) -> detections
    nms_cluster  detectron2 batched_nms as called at PI:554-560 / IU:31-36 -> keep indices
    reg_nll      compute_reg_scores' per-row NLL, scoring_rules.py:68-74
    wino_filter_transform / wino_conv3x3
                 the head's `Conv2d(C, K, 3, padding=1) [+ ReLU + Dropout]` (PR:403-484) for all MC runs and FPN levels in one
                 launch: channels-last activations in, channels-last (trunk) or NCHW planes (predictors) out
    conv1x1_filter_split / conv1x1_split, stem7x7_filter_split / stem7x7_split, maxpool3x3s2_cl, im2col3x3s2_cl
                 the channels-last trunk (round 4): the backbone's / FPN's 1x1 convolutions with bias / residual / ReLU in the store, the
                 7x7 stem taking the frame as loaded (normalisation + padding on load) and its max-pool
"""
from typing import List, Tuple

import torch

from . import hip, hotpath

_LIB = torch.library.Library("pod_mi355x", "DEF")
_LIB.define("predict(Tensor[] box_cls, Tensor[] box_delta, Tensor[] box_cls_var, Tensor[] box_reg_var, Tensor[] anchors, "
            "str mode, int[] image_size, int[] out_size, int num_classes=7, int topk_candidates=1000, float score_thresh=0.05, "
            "float nms_thresh=0.5, int max_detections=100, int cls_var_num_samples=10, float affinity_thresh=0.9, "
            "bool merge_quirk=True, str box_merge_mode='bayesian_inference', str cls_merge_mode='max_score', int draw_id=-1) "
            "-> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_LIB.define("nms_cluster(Tensor boxes, Tensor scores, Tensor classes, float nms_thresh, int max_detections, int num_classes) -> Tensor")
_LIB.define("reg_nll(Tensor means, Tensor covs, Tensor gt) -> Tensor")
_LIB.define("wino_filter_transform(Tensor weight) -> Tensor")
_LIB.define("wino_conv3x3(Tensor src, Tensor U, Tensor? bias, Tensor blocks, int K, int out_elements, bool planes=False, bool relu=False, "
            "float dropout_p=0.0, int seed=0, int offset=0) -> Tensor")
_LIB.define("conv1x1_filter_split(Tensor weight) -> Tensor")
_LIB.define("conv1x1_split(Tensor x, Tensor Ws, Tensor? bias, Tensor? residual, int h, int w, int stride, int cout, bool relu=False, int n_splits=1) -> Tensor")
_LIB.define("stem7x7_filter_split(Tensor weight) -> Tensor")
_LIB.define("stem7x7_split(Tensor frame, Tensor Ws, Tensor? bias, Tensor? mean, Tensor? std, int padded_h, int padded_w, bool relu=True) -> Tensor")
_LIB.define("maxpool3x3s2_cl(Tensor x, int h, int w) -> Tensor")
_LIB.define("im2col3x3s2_cl(Tensor x, int h, int w, bool relu=False) -> Tensor")

_PATHS = {}            # key -> [HotPath, anchor identity, the anchor tensors]; insertion order = LRU order
_MAX_PATHS = 16
_TABLES_OK = {}        # (data_ptr, _version, shape, src pixels, out elements, per-pixel) of block tables already validated


def _check_dense(name: str, ts: List[torch.Tensor], like: List[torch.Tensor]) -> None:
    torch._check(len(ts) == len(like), lambda: "{}: {} levels, box_cls has {}".format(name, len(ts), len(like)))
    for l, (t, ref) in enumerate(zip(ts, like)):
        torch._check(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4, lambda: "{}[{}] must be a CUDA fp32 (runs, A*C, H, W) tensor".format(name, l))
        torch._check(t.shape[0] == ref.shape[0] and tuple(t.shape[2:]) == tuple(ref.shape[2:]) and t.device == ref.device,
                     lambda: "{}[{}]: shape {} does not match box_cls {}".format(name, l, tuple(t.shape), tuple(ref.shape)))


def _predict(box_cls, box_delta, box_cls_var, box_reg_var, anchors, mode, image_size, out_size, num_classes=7, topk_candidates=1000,
             score_thresh=0.05, nms_thresh=0.5, max_detections=100, cls_var_num_samples=10, affinity_thresh=0.9, merge_quirk=True,
             box_merge_mode="bayesian_inference", cls_merge_mode="max_score", draw_id=-1) -> Tuple[torch.Tensor, ...]:
    # (the dispatcher hands a Python kernel only the arguments the caller wrote: the schema's defaults are repeated here)
    torch._check(len(box_cls) >= 1, lambda: "box_cls: at least one FPN level")
    _check_dense("box_cls", box_cls, box_cls)
    _check_dense("box_delta", box_delta, box_cls)
    if box_cls_var:
        _check_dense("box_cls_var", box_cls_var, box_cls)
    if box_reg_var:
        _check_dense("box_reg_var", box_reg_var, box_cls)
    torch._check(len(anchors) == len(box_cls) and len(image_size) == 2 and len(out_size) == 2, lambda: "anchors per level; (h, w) sizes")
    K = int(num_classes)
    torch._check(box_cls[0].shape[1] % K == 0, lambda: "box_cls channels {} are not A * num_classes".format(box_cls[0].shape[1]))
    A = box_cls[0].shape[1] // K
    n_runs = int(box_cls[0].shape[0])
    D = box_reg_var[0].shape[1] // A if box_reg_var else 0
    dev = box_cls[0].device
    shapes = tuple(tuple(t.shape[2:]) for t in box_cls)
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (shapes, A, K, n_runs, bool(box_cls_var), D, str(dev), stream, topk_candidates, score_thresh, nms_thresh, max_detections,
           cls_var_num_samples, affinity_thresh, merge_quirk)
    # The workspace keeps a COPY of the anchors.  The entry also keeps the caller's anchor tensors alive (so that their addresses
    # cannot be handed to other tensors while the identity below is trusted) and re-uploads into the same workspace when a call
    # brings different ones -- detectron2's anchor generator makes new tensors on every forward; that must neither decode
    # against stale anchors nor grow the cache.
    akey = tuple((a.data_ptr(), a._version, tuple(a.shape)) for a in anchors)
    entry = _PATHS.pop(key, None)
    if entry is None:      # one workspace per (geometry, stream), as the predictor keeps them
        p = hotpath.PathParams(num_classes=K, num_anchors=A, topk_candidates=topk_candidates, score_thresh=score_thresh,
                               nms_thresh=nms_thresh, max_detections=max_detections, cls_var_num_samples=cls_var_num_samples,
                               affinity_thresh=affinity_thresh, merge_quirk=merge_quirk)
        entry = [hotpath.HotPath(shapes, anchors, p, n_runs=n_runs, has_cls_var=bool(box_cls_var), cov_dims=D, device=dev), akey, list(anchors)]
    elif entry[1] != akey:
        with torch.cuda.device(dev):
            entry[0].set_anchors(anchors)
        entry[1], entry[2] = akey, list(anchors)
    _PATHS[key] = entry                                   # most recently used last
    while len(_PATHS) > _MAX_PATHS:
        _PATHS.pop(next(iter(_PATHS)))
    hp = entry[0]
    with torch.cuda.device(dev):
        det = hp.run(mode, list(box_cls), list(box_delta), list(box_cls_var) or None, list(box_reg_var) or None,
                     image_size=tuple(image_size), out_size=tuple(out_size), box_merge_mode=box_merge_mode,
                     cls_merge_mode=cls_merge_mode, draw_id=None if draw_id < 0 else int(draw_id))
    m = det.count()      # the one host sync of an image (the operator returns exact-size tensors, like the reference)
    return det.boxes[:m], det.cov[:m], det.scores[:m], det.classes[:m].long(), det.probs[:m]


def _nms_cluster(boxes, scores, classes, nms_thresh, max_detections, num_classes) -> torch.Tensor:
    torch._check(boxes.is_cuda and boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.shape[1] == 4, lambda: "boxes: CUDA fp32 (n, 4)")
    n = int(boxes.shape[0])
    torch._check(tuple(scores.shape) == (n,) and scores.dtype == torch.float32 and scores.device == boxes.device, lambda: "scores: fp32 (n,)")
    torch._check(tuple(classes.shape) == (n,) and classes.device == boxes.device, lambda: "classes: (n,)")
    torch._check(n <= hip.POD_MAX_CANDIDATES, lambda: "at most {} boxes".format(hip.POD_MAX_CANDIDATES))
    lib, P = hip.load(), hip.ptr
    cfg = hip.PodConfig()
    cfg.max_detections, cfg.nms_thresh, cfg.num_classes = int(max_detections), float(nms_thresh), int(num_classes)
    dev = boxes.device
    with torch.cuda.device(dev):
        cap = max(n, 1)
        b = torch.zeros((cap, 4), dtype=torch.float32, device=dev)
        b[:n] = boxes
        s, c = scores.contiguous(), classes.to(torch.int32).contiguous()
        if n == 0:
            s, c = torch.zeros(1, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        keep = torch.empty(hip.POD_MAX_DETECTIONS, dtype=torch.int32, device=dev)
        n_keep = torch.zeros(1, dtype=torch.int32, device=dev)
        nt = torch.tensor([n], dtype=torch.int32, device=dev)
        scratch = torch.zeros(lib.pod_nms_scratch_bytes(cap), dtype=torch.uint8, device=dev)
        hip.check(lib.pod_nms_cluster(cfg, P(nt), cap, P(b), P(s), P(c), P(keep), P(n_keep), P(scratch), hip.current_stream()), "pod_nms_cluster")
        return keep[:int(n_keep.item())].long()


def _reg_nll(means, covs, gt) -> torch.Tensor:
    n = int(means.shape[0])
    for name, t, shape in (("means", means, (n, 4)), ("covs", covs, (n, 4, 4)), ("gt", gt, (n, 4))):
        torch._check(t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == shape, lambda: "{}: CUDA fp32 {}".format(name, shape))
    out = torch.empty(n, dtype=torch.float32, device=means.device)
    with torch.cuda.device(means.device):
        hip.check(hip.load().pod_reg_nll(hip.ptr(means.contiguous()), hip.ptr(covs.contiguous()), hip.ptr(gt.contiguous()), n, hip.ptr(out),
                                         hip.current_stream()), "pod_reg_nll")
    return out


def _wino_filter_transform(weight) -> torch.Tensor:
    torch._check(weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3),
                 lambda: "weight: CUDA fp32 (K, C, 3, 3)")
    K, C = int(weight.shape[0]), int(weight.shape[1])
    torch._check(C % 8 == 0 and (K + 63) // 64 in (1, 2, 4, 8), lambda: "C % 8 == 0 and K <= 512 in {64, 128, 256, 512} after padding")
    U = torch.empty(24 * ((K + 63) // 64 * 64) * C, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        hip.check(hip.load().pod_wino_filter_transform(hip.ptr(weight.contiguous()), hip.ptr(U), K, C, hip.current_stream()),
                  "pod_wino_filter_transform")
    return U


def _wino_conv3x3(src, U, bias, blocks, K, out_elements, planes=False, relu=False, dropout_p=0.0, seed=0, offset=0) -> torch.Tensor:
    """src (pixels, C) channels-last; blocks: the int32 (n, 4) records of include/pod_mi355x.h (pod_compare_amd.wino.block_table builds
    them); K real output channels; the result has out_elements floats: (pixels, round_up(K, 64)) channels-last, or with planes=True the
    NCHW images of K planes each the records' output side describes (pixels nobody writes stay zero)."""
    torch._check(src.is_cuda and src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous(), lambda: "src: contiguous CUDA fp32 (pixels, C)")
    C, Kpad = int(src.shape[1]), (int(K) + 63) // 64 * 64
    torch._check(U.is_cuda and U.dtype == torch.float32 and U.numel() == 24 * Kpad * C, lambda: "U: wino_filter_transform of a (K, C, 3, 3) weight")
    torch._check(blocks.is_cuda and blocks.dtype == torch.int32 and blocks.dim() == 2 and blocks.shape[1] == 4 and blocks.is_contiguous(),
                 lambda: "blocks: contiguous CUDA int32 (n, 4)")
    torch._check(bias is None or (bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == Kpad), lambda: "bias: round_up(K, 64) fp32 values")
    torch._check(planes or out_elements == src.shape[0] * Kpad, lambda: "channels-last output: out_elements == pixels * round_up(K, 64)")
    torch._check(0.0 <= dropout_p < 1.0 and not (planes and dropout_p), lambda: "dropout_p in [0, 1), 0 with planes=True")
    per_px = int(K) if planes else Kpad
    tkey = (blocks.data_ptr(), blocks._version, int(blocks.shape[0]), int(src.shape[0]), int(out_elements), per_px, C)
    if blocks.shape[0] and _TABLES_OK.get(tkey) is not blocks:
        # the kernel reads and writes at offsets taken from the table: every canvas must lie inside the two buffers and inside the
        # kernel's 32-bit byte offsets (buffer resource size; the out-of-image sentinel 0x7FFFFF00 must lie past the canvas).
        # Validated once per table (a device round trip): the entry keeps the table alive, so its address cannot be reused.
        b = blocks.to(torch.int64)
        z, w = b[:, 2], b[:, 3]
        gcols, H, W, n = (z >> 24) & 0xFF, (z >> 12) & 0xFFF, z & 0xFFF, (w >> 24) & 0xFF
        last_in, last_out = b[:, 0] + n * H * W, b[:, 1] + n * H * W
        ok = ((b[:, 0] >= 0) & (b[:, 1] >= 0) & (H > 0) & (W > 0) & (n > 0) & (n <= 127) & (gcols > 0) & (last_in <= src.shape[0])
              & (last_out * per_px <= int(out_elements)) & (((w >> 12) & 0xFFF) * 16 < ((n + gcols - 1) // gcols) * (H + 1))
              & ((w & 0xFFF) * 16 < gcols * (W + 1)) & (n * H * W * max(C, per_px) * 4 <= 2 ** 31 - 256))
        torch._check(bool(ok.all()), lambda: "blocks: a record lies outside src / the output / the 32-bit canvas (record {})".format(int((~ok).nonzero()[0])))
        if len(_TABLES_OK) >= 64:
            _TABLES_OK.pop(next(iter(_TABLES_OK)))
        _TABLES_OK[tkey] = blocks
    out = torch.zeros(int(out_elements), dtype=torch.float32, device=src.device) if planes else \
        torch.empty((src.shape[0], Kpad), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        hip.check(hip.load().pod_wino_conv3x3(hip.ptr(src), hip.ptr(out), hip.ptr(U), hip.ptr(bias), hip.ptr(blocks), int(blocks.shape[0]), C, Kpad,
                                              int(K) if planes else 0, 1 if relu else 0, float(dropout_p), int(seed), int(offset),
                                              None, hip.current_stream()), "pod_wino_conv3x3")
    return out


def _absmax_word(t: torch.Tensor) -> torch.Tensor:
    """The operand abs-max word of a stateless operator call (include/pod_mi355x.h): pod_absmax of the whole tensor, every call -- the
    model's own path (pod_compare_amd/amax.py) takes the word the producing kernel published instead."""
    w = torch.zeros(512, dtype=torch.float32, device=t.device)          # POD_AMAX_FLOATS
    with torch.cuda.device(t.device):
        hip.check(hip.load().pod_absmax(hip.ptr(t), t.numel(), hip.ptr(w), hip.current_stream()), "pod_absmax")
    return w


def _conv1x1_filter_split(weight) -> torch.Tensor:
    """pod_conv1x1_filter_split: a (Cout, Cin, 1, 1) / (Cout, Cin) fp32 weight as two f16 terms per (power-of-two-scaled) value, in the order the
    kernel loads them, + the abs-max trailer (2 * Cout * Cin + 8 int16 words).  Cout % 64 == 0, Cin % 16 == 0."""
    torch._check(weight.is_cuda and weight.dtype == torch.float32 and weight.dim() in (2, 4) and weight.numel() == weight.shape[0] * weight.shape[1],
                 lambda: "weight: CUDA fp32 (Cout, Cin) or (Cout, Cin, 1, 1)")
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    torch._check(cout >= 64 and cout % 64 == 0 and cin >= 16 and cin % 16 == 0, lambda: "Cout % 64 == 0 and Cin % 16 == 0 required")
    ws = torch.empty(2 * cout * cin + 8, dtype=torch.int16, device=weight.device)
    with torch.cuda.device(weight.device):
        hip.check(hip.load().pod_conv1x1_filter_split(hip.ptr(weight.contiguous()), hip.ptr(ws), cout, cin, hip.current_stream()), "pod_conv1x1_filter_split")
    return ws


def _conv1x1_split(x, Ws, bias, residual, h, w, stride, cout, relu=False, n_splits=1) -> torch.Tensor:
    """pod_conv1x1_split on one channels-last image x (h * w, Cin): act(conv1x1(x, stride) + bias + residual) as (h_out * w_out, cout)."""
    torch._check(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.shape[0] == h * w, lambda: "x: contiguous CUDA fp32 (h * w, Cin)")
    cin = int(x.shape[1])
    torch._check(stride in (1, 2) and cout >= 64 and cout % 64 == 0 and cin >= 16 and cin % 16 == 0, lambda: "stride 1 or 2, cout % 64 == 0, Cin % 16 == 0")
    torch._check(Ws.is_cuda and Ws.dtype == torch.int16 and Ws.numel() == 2 * cout * cin + 8, lambda: "Ws: conv1x1_filter_split of a (cout, Cin) weight")
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    torch._check(bias is None or (bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == cout and bias.is_contiguous()), lambda: "bias: cout fp32 values")
    torch._check(residual is None or (residual.is_cuda and residual.dtype == torch.float32 and residual.is_contiguous() and tuple(residual.shape) == (ho * wo, cout)),
                 lambda: "residual: contiguous (h_out * w_out, cout) fp32")
    nks = cin // 16
    torch._check(1 <= n_splits <= 16 and nks % n_splits == 0, lambda: "n_splits must divide Cin / 16 (1 .. 16)")
    y = torch.empty((ho * wo, cout), dtype=torch.float32, device=x.device)
    partials = torch.empty((n_splits, ho * wo, cout), dtype=torch.float32, device=x.device) if n_splits > 1 else None
    with torch.cuda.device(x.device):
        hip.check(hip.load().pod_conv1x1_split(hip.ptr(x), hip.ptr(y), hip.ptr(Ws), hip.ptr(bias), hip.ptr(residual), ho, wo, int(h), int(w), int(stride), cin, int(cout),
                                               1 if relu else 0, int(n_splits), hip.ptr(partials), 0, hip.ptr(_absmax_word(x)), None, hip.current_stream()), "pod_conv1x1_split")
    return y


def _stem7x7_filter_split(weight) -> torch.Tensor:
    torch._check(weight.is_cuda and weight.dtype == torch.float32 and tuple(weight.shape) == (64, 3, 7, 7), lambda: "weight: CUDA fp32 (64, 3, 7, 7)")
    ws = torch.empty(2 * 64 * 192 + 8, dtype=torch.int16, device=weight.device)
    with torch.cuda.device(weight.device):
        hip.check(hip.load().pod_stem7x7_filter_split(hip.ptr(weight.contiguous()), hip.ptr(ws), hip.current_stream()), "pod_stem7x7_filter_split")
    return ws


def _stem7x7_split(frame, Ws, bias, mean, std, padded_h, padded_w, relu=True) -> torch.Tensor:
    """pod_stem7x7_split: frame (3, H, W) uint8 / fp32 -> ((padded_h - 1) // 2 + 1) * ((padded_w - 1) // 2 + 1) pixels x 64 channels, channels-last;
    mean / std (3 values each) given: normalised on load."""
    torch._check(frame.is_cuda and frame.dtype in (torch.uint8, torch.float32) and frame.dim() == 3 and frame.shape[0] == 3 and frame.is_contiguous(),
                 lambda: "frame: contiguous CUDA uint8 / fp32 (3, H, W)")
    hi, wi = int(frame.shape[1]), int(frame.shape[2])
    torch._check(padded_h >= hi and padded_w >= wi and padded_h <= 16384 and padded_w <= 16384, lambda: "padded extent must contain the frame (<= 16384)")
    torch._check(Ws.is_cuda and Ws.dtype == torch.int16 and Ws.numel() == 2 * 64 * 192 + 8, lambda: "Ws: stem7x7_filter_split of the (64, 3, 7, 7) weight")
    torch._check((mean is None) == (std is None), lambda: "mean and std: both or neither")
    for t in (mean, std):
        torch._check(t is None or (t.is_cuda and t.dtype == torch.float32 and t.numel() == 3 and t.is_contiguous()), lambda: "mean / std: 3 contiguous CUDA fp32 values")
    torch._check(bias is None or (bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == 64 and bias.is_contiguous()), lambda: "bias: 64 fp32 values")
    ho, wo = (int(padded_h) - 1) // 2 + 1, (int(padded_w) - 1) // 2 + 1
    y = torch.empty((ho * wo, 64), dtype=torch.float32, device=frame.device)
    # in_amax >= max |(x - mean) / std|: (max |x| + max |mean|) / min |std|, max |x| = 255 for a uint8 frame
    bound = torch.full((512,), 255.0, dtype=torch.float32, device=frame.device) if frame.dtype == torch.uint8 else _absmax_word(frame)
    if mean is not None:
        bound = (bound + mean.abs().max()) / std.abs().min()
    with torch.cuda.device(frame.device):
        hip.check(hip.load().pod_stem7x7_split(hip.ptr(frame), 1 if frame.dtype == torch.uint8 else 0, hi, wi, hip.ptr(mean), hip.ptr(std), hip.ptr(y), hip.ptr(Ws),
                                               hip.ptr(bias), int(padded_h), int(padded_w), 1 if relu else 0, hip.ptr(bound), None, hip.current_stream()), "pod_stem7x7_split")
    return y


def _maxpool3x3s2_cl(x, h, w) -> torch.Tensor:
    torch._check(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.shape[0] == h * w and x.shape[1] % 4 == 0 and x.shape[1] >= 4,
                 lambda: "x: contiguous CUDA fp32 (h * w, C), C % 4 == 0")
    y = torch.empty((((h - 1) // 2 + 1) * ((w - 1) // 2 + 1), x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.check(hip.load().pod_maxpool3x3s2_cl(hip.ptr(x), hip.ptr(y), int(h), int(w), int(x.shape[1]), hip.current_stream()), "pod_maxpool3x3s2_cl")
    return y


def _im2col3x3s2_cl(x, h, w, relu=False) -> torch.Tensor:
    """The patch matrix of a 3x3 / stride 2 / padding 1 convolution (FPN's p6 / p7): conv1x1_split on it with the weight laid out
    (Cout, ty, tx, Cin) is the convolution."""
    torch._check(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.shape[0] == h * w and x.shape[1] % 4 == 0 and x.shape[1] >= 4,
                 lambda: "x: contiguous CUDA fp32 (h * w, C), C % 4 == 0")
    y = torch.empty((((h - 1) // 2 + 1) * ((w - 1) // 2 + 1), 9 * x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.check(hip.load().pod_im2col3x3s2_cl(hip.ptr(x), hip.ptr(y), int(h), int(w), int(x.shape[1]), 1 if relu else 0, hip.current_stream()), "pod_im2col3x3s2_cl")
    return y


_IMPL = torch.library.Library("pod_mi355x", "IMPL")
_IMPL.impl("im2col3x3s2_cl", _im2col3x3s2_cl, "CUDA")
_IMPL.impl("conv1x1_filter_split", _conv1x1_filter_split, "CUDA")
_IMPL.impl("conv1x1_split", _conv1x1_split, "CUDA")
_IMPL.impl("stem7x7_filter_split", _stem7x7_filter_split, "CUDA")
_IMPL.impl("stem7x7_split", _stem7x7_split, "CUDA")
_IMPL.impl("maxpool3x3s2_cl", _maxpool3x3s2_cl, "CUDA")
_IMPL.impl("predict", _predict, "CUDA")
_IMPL.impl("nms_cluster", _nms_cluster, "CUDA")
_IMPL.impl("reg_nll", _reg_nll, "CUDA")
_IMPL.impl("wino_filter_transform", _wino_filter_transform, "CUDA")
_IMPL.impl("wino_conv3x3", _wino_conv3x3, "CUDA")
