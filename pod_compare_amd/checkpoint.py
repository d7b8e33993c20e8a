"""Checkpoint loading for the predictor's models (row a1 / a15 of SURVEY 8a).

Replaces the reference's
    DetectionCheckpointer(model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=True)     PI:78-84
and, for ensembles, the same call on `<parent of OUTPUT_DIR>/random_seed_<s>` for every member              PI:59-77.

`resume_or_load(..., resume=True)` semantics (detectron2 / fvcore Checkpointer, restated): if
`<save_dir>/last_checkpoint` exists its content names the file to load inside `save_dir`; otherwise
`cfg.MODEL.WEIGHTS` is loaded; an empty path loads nothing (random init).  Files are `.pth` (torch.save of
`{"model": state_dict, ...}` or of a bare state dict) or `.pkl` (pickled `{"model": {name: ndarray}}`; detectron2's
own names, or the Caffe2 names of the ImageNet-pretrained MSRA backbones, which are renamed here).

detectron2 names -> pod_compare_amd.modeling.ProbabilisticRetinaNet:
    backbone.bottom_up.stem.conv1.{weight, norm.*}              bottom_up.stem.{0.weight, 1.*}
    backbone.bottom_up.res<s>.<b>.<conv>.{weight, norm.*}       bottom_up.res<s>.<b>.<conv>.{0.weight, 1.*}
    backbone.fpn_lateral<3|4|5>, backbone.fpn_output<3|4|5>      fpn.lateral.<0|1|2>, fpn.output.<0|1|2>
    backbone.top_block.p6 / p7                                   fpn.p6 / fpn.p7
    head.cls_subnet.<i>, head.bbox_subnet.<i>                    head.cls_subnet.<j>, head.bbox_subnet.<j>
        the reference's subnets are nn.Sequential(conv, ReLU[, Dropout]) x 4 (PR:403-427): the j-th conv sits at
        index 3j with dropout, 2j without; the j-th conv found in the file is taken, whatever its stride
    head.cls_score / bbox_pred / cls_var / bbox_cov              same names
Buffers that carry no weights (anchor_generator.cell_anchors.*, pixel_mean, pixel_std) are ignored.

Loading happens on the UNFOLDED model (conv + FrozenBatchNorm2d pairs); `modeling.fold_frozen_bn` runs afterwards.
"""
import os
import pickle
import re
import warnings
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import modeling

_BN_FIELDS = ("weight", "bias", "running_mean", "running_var")
_IGNORED = re.compile(r"^(anchor_generator\.|pixel_mean$|pixel_std$|backbone\.bottom_up\.stem\.fc|backbone\.bottom_up\.linear)")


class CheckpointError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------------
# key map
# ---------------------------------------------------------------------------------------------------
def detectron2_to_local_keys(model: "modeling.ProbabilisticRetinaNet", subnet_stride: int) -> Dict[str, str]:
    """detectron2 state-dict key -> key of `model.state_dict()` (unfolded model).  `subnet_stride`: distance between
    the conv indices of the head's nn.Sequential subnets (3 with nn.Dropout entries, 2 without)."""
    out: Dict[str, str] = {}

    def conv_bn(d2: str, local: str):
        out[d2 + ".weight"] = local + ".0.weight"
        for f in _BN_FIELDS:
            out[d2 + ".norm." + f] = local + ".1." + f

    def conv(d2: str, local: str):
        out[d2 + ".weight"] = local + ".weight"
        out[d2 + ".bias"] = local + ".bias"

    conv_bn("backbone.bottom_up.stem.conv1", "bottom_up.stem")
    for stage in ("res2", "res3", "res4", "res5"):
        for b, block in enumerate(getattr(model.bottom_up, stage)):
            base_d2, base = "backbone.bottom_up.%s.%d" % (stage, b), "bottom_up.%s.%d" % (stage, b)
            if block.shortcut is not None:
                conv_bn(base_d2 + ".shortcut", base + ".shortcut")
            for c in ("conv1", "conv2", "conv3"):
                conv_bn(base_d2 + "." + c, base + "." + c)
    for i, lvl in enumerate((3, 4, 5)):
        conv("backbone.fpn_lateral%d" % lvl, "fpn.lateral.%d" % i)
        conv("backbone.fpn_output%d" % lvl, "fpn.output.%d" % i)
    conv("backbone.top_block.p6", "fpn.p6")
    conv("backbone.top_block.p7", "fpn.p7")
    for sub in ("cls_subnet", "bbox_subnet"):
        for j in range(len(getattr(model.head, sub))):
            conv("head.%s.%d" % (sub, j * subnet_stride), "head.%s.%d" % (sub, j))
    for name in ("cls_score", "bbox_pred", "cls_var", "bbox_cov"):
        if getattr(model.head, name, None) is not None:
            conv("head." + name, "head." + name)
    return out


def _subnet_stride(keys) -> int:
    """2 or 3, from the conv indices present in the file (PR:403-427); 3 if the file has no head."""
    idx = sorted({int(m.group(1)) for k in keys for m in [re.match(r"^head\.cls_subnet\.(\d+)\.weight$", k)] if m})
    if len(idx) >= 2:
        return idx[1] - idx[0]
    return 3


_C2_BLOCK = {"branch2a": "conv1", "branch2b": "conv2", "branch2c": "conv3", "branch1": "shortcut"}


def caffe2_to_detectron2_keys(sd: Dict[str, object]) -> Dict[str, object]:
    """Names of the ImageNet-pretrained MSRA ResNets (`R-50.pkl`: conv1_w, res_conv1_bn_s, res2_0_branch2a_w,
    res2_0_branch2a_bn_b, ...) -> detectron2 backbone names.  Only the ResNet trunk is covered (that is all these
    files hold besides the ImageNet classifier, which is dropped)."""
    out = {}
    for k, v in sd.items():
        m = re.match(r"^res(\d)_(\d+)_(branch2a|branch2b|branch2c|branch1)_(w|bn_s|bn_b)$", k)
        if m:
            base = "backbone.bottom_up.res%s.%s.%s" % (m.group(1), m.group(2), _C2_BLOCK[m.group(3)])
        elif k in ("conv1_w", "res_conv1_bn_s", "res_conv1_bn_b"):
            base, m = "backbone.bottom_up.stem.conv1", re.match(r"^.*?_(w|bn_s|bn_b)$", k)
        else:
            continue                                   # fc1000_*, pred_*: not part of a detector
        out[base + {"w": ".weight", "bn_s": ".norm.weight", "bn_b": ".norm.bias"}[m.group(m.lastindex)]] = v
    return out


# ---------------------------------------------------------------------------------------------------
# files
# ---------------------------------------------------------------------------------------------------
def read_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
    """The model state dict stored in a `.pth` / `.pkl` file, detectron2 names, CPU tensors."""
    if "://" in path:
        raise CheckpointError("{}: remote checkpoints are not fetched (no network); give a local path".format(path))
    if not os.path.isfile(path):
        raise CheckpointError("checkpoint {} does not exist".format(path))
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        sd = data["model"] if isinstance(data, dict) and "model" in data else (data["blobs"] if isinstance(data, dict) and "blobs" in data else data)
        if any(k in sd for k in ("conv1_w", "res_conv1_bn_s")):
            sd = caffe2_to_detectron2_keys(sd)
    else:
        try:        # detectron2 checkpoints hold tensors, numbers and strings only: the restricted unpickler is enough
            data = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            # older files pickle numpy scalars / argparse namespaces next to the weights.  Full unpickling executes code from the
            # file: only for files the operator vouches for (POD_TRUSTED_CHECKPOINTS=1), never as a silent fallback.
            if os.environ.get("POD_TRUSTED_CHECKPOINTS") != "1":
                raise CheckpointError("{}: not readable by the restricted unpickler ({}); if the file is trusted, set "
                                      "POD_TRUSTED_CHECKPOINTS=1 to allow full unpickling".format(path, str(e).splitlines()[0])) from e
            data = torch.load(path, map_location="cpu", weights_only=False)
        sd = data["model"] if isinstance(data, dict) and "model" in data and isinstance(data["model"], dict) else data
    out = {}
    for k, v in sd.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v.copy())
        if not torch.is_tensor(v):
            continue
        out[k[7:] if k.startswith("module.") else k] = v.detach().cpu()      # DistributedDataParallel prefix
    return out


def resolve_checkpoint(save_dir: Optional[str], weights: str) -> str:
    """resume_or_load(weights, resume=True): `<save_dir>/last_checkpoint` wins over `weights`; '' = nothing to load."""
    if save_dir:
        tag = os.path.join(save_dir, "last_checkpoint")
        if os.path.isfile(tag):
            with open(tag, "r") as f:
                name = f.read().strip()
            return os.path.join(save_dir, name)
    return weights or ""


# ---------------------------------------------------------------------------------------------------
# load
# ---------------------------------------------------------------------------------------------------
def load_detectron2_state_dict(model: "modeling.ProbabilisticRetinaNet", sd: Dict[str, torch.Tensor],
                               strict: bool = False) -> Tuple[List[str], List[str]]:
    """Copies a detectron2-named state dict into the (unfolded) model.  Returns (missing local keys, unexpected
    file keys).  Shape mismatches always raise; missing / unexpected keys raise only with strict=True (a backbone-only
    file legitimately lacks the head)."""
    if any(isinstance(m, torch.nn.Identity) for m in model.bottom_up.stem):
        raise CheckpointError("load weights BEFORE modeling.fold_frozen_bn(model): the FrozenBN statistics are already folded")
    kmap = detectron2_to_local_keys(model, _subnet_stride(sd.keys()))
    own = model.state_dict()
    loaded, unexpected = set(), []
    staged = {}
    for k, v in sd.items():
        if _IGNORED.match(k):
            continue
        local = kmap.get(k)
        if local is None or local not in own:
            unexpected.append(k)
            continue
        if tuple(own[local].shape) != tuple(v.shape):
            raise CheckpointError("{} -> {}: shape {} in the file, {} in the model".format(k, local, tuple(v.shape), tuple(own[local].shape)))
        staged[local] = v
        loaded.add(local)
    # FrozenBatchNorm2d files without statistics (Caffe2 affine-only BN): mean 0, var 1 - eps, i.e. scale = weight
    for local in list(loaded):
        if local.endswith(".1.weight"):
            stem = local[:-len("weight")]
            if stem + "running_mean" in own and stem + "running_mean" not in loaded:
                staged[stem + "running_mean"] = torch.zeros_like(own[stem + "running_mean"])
                staged[stem + "running_var"] = torch.ones_like(own[stem + "running_var"]) - 1e-5
                loaded.update((stem + "running_mean", stem + "running_var"))
    with torch.no_grad():
        for local, v in staged.items():
            own[local].copy_(v.to(own[local].dtype))
    missing = sorted(k for k in own if k not in loaded)
    if strict and (missing or unexpected):
        raise CheckpointError("strict load failed: missing {} unexpected {}".format(missing[:8], unexpected[:8]))
    return missing, sorted(unexpected)


def to_detectron2_state_dict(model: "modeling.ProbabilisticRetinaNet", with_dropout_entries: Optional[bool] = None) -> Dict[str, torch.Tensor]:
    """The (unfolded) model's weights under detectron2's names, e.g. to hand a model built here to the reference's
    DetectionCheckpointer.  with_dropout_entries: head subnets laid out as Sequential(conv, ReLU, Dropout) (PR:420-424);
    default: whether the model uses dropout."""
    if with_dropout_entries is None:
        with_dropout_entries = model.use_dropout
    kmap = detectron2_to_local_keys(model, 3 if with_dropout_entries else 2)
    own = model.state_dict()
    return {d2: own[local].detach().clone() for d2, local in kmap.items() if local in own}


def load_model_weights(model, save_dir: Optional[str], weights: str, strict: bool = False) -> str:
    """PI:78-84 for one model: resolve, read, load.  Returns the path loaded ('' = none: random init kept)."""
    path = resolve_checkpoint(save_dir, weights)
    if not path:
        return ""
    missing, unexpected = load_detectron2_state_dict(model, read_checkpoint_file(path), strict=strict)
    n_own = len(model.state_dict())
    for part in ("bottom_up.", "head."):
        own_part = [k for k in model.state_dict() if k.startswith(part)]
        if own_part and all(k in missing for k in own_part):
            # a renamed head / backbone would otherwise leave that whole part at its random init behind a mere warning
            if part == "bottom_up." or len(missing) == n_own:
                raise CheckpointError("checkpoint {}: no {} tensor of the model was found in the file (names: {} ...)".format(
                    path, part.rstrip("."), sorted(unexpected)[:4]))
            warnings.warn("checkpoint {} holds NO tensor of the model's {} (a backbone-only file?): that part keeps its random init".format(
                path, part.rstrip(".")))
    if missing or unexpected:
        warnings.warn("checkpoint {}: {} model tensors not in the file (e.g. {}), {} file tensors unused (e.g. {})".format(
            path, len(missing), missing[:3], len(unexpected), unexpected[:3]))
    return path
