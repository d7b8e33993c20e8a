#!/usr/bin/env python3
"""Generate tests/golden/head_*.npz by running the REFERENCE's own model classes in this container.

    python oracle/make_golden_model.py        # needs /root/reference (read-only) -- build container only

`probabilistic_modeling.probabilistic_retinanet` is imported where it lies (oracle/refimport.py; detectron2 / fvcore names
served by oracle/refstub -- the base classes there carry no arithmetic the head uses: the reference re-creates every layer
itself).  For each head variant (plain / dropout / reg_cls_var / reg_cls_var_dropout, diagonal + full covariance):

  * `ProbabilisticRetinaNet(cfg)` is constructed with a feature-replaying stand-in backbone; the constants its constructor
    leaves in the head (PR:447-484) are recorded, then seeded weights are loaded (oracle/model_fixture.py);
  * eval mode: `head(features)` (PR:486-537) and `model(inputs, return_anchorwise_output=True)` (PR:110-112, 335-361);
  * MC mode, as PI:53-56 sets it up (`model.train()`): `model(inputs, return_anchorwise_output=True,
    num_mc_dropout_runs=N)` (PR:103-108) with `torch.nn.functional.dropout` serving seeded keep-masks that are recorded in
    call order and tagged (subnet, evaluation, run, level, layer) by forward hooks on the reference's own modules.

A fixture is data: parameters, seed, input checksums, bit-packed masks, the reference's outputs.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import model_fixture as mf  # noqa: E402
from oracle.refimport import load_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 4101


def ns(**kw):
    return types.SimpleNamespace(**kw)


def sha(tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


class ReplayBackbone(torch.nn.Module):
    """Returns the FPN features it was handed (the ResNet-FPN is not part of row a1)."""

    def __init__(self, features):
        super().__init__()
        self.names = ["p3", "p4", "p5", "p6", "p7"]
        self.features = features

    def output_shape(self):
        from detectron2.layers import ShapeSpec
        return {n: ShapeSpec(channels=mf.CHANNELS, stride=s) for n, s in zip(self.names, mf.STRIDES)}

    def forward(self, x):
        return dict(zip(self.names, self.features))


def make_cfg(v, backbone):
    return ns(
        STUB_BACKBONE=backbone,
        SOLVER=ns(STEPS=(1, 2)),
        MODEL=ns(
            DEVICE="cpu",
            RETINANET=ns(NUM_CLASSES=mf.NUM_CLASSES, NUM_CONVS=mf.NUM_CONVS, PRIOR_PROB=0.01, IN_FEATURES=["p3", "p4", "p5", "p6", "p7"]),
            ANCHOR_GENERATOR=ns(SIZES=mf.ANCHOR_SIZES, ASPECT_RATIOS=mf.ASPECT_RATIOS, OFFSET=0.0),
            PROBABILISTIC_MODELING=ns(
                DROPOUT_RATE=v["dropout_rate"],
                CLS_VAR_LOSS=ns(NAME="loss_attenuation" if v["cls_var"] else "none", NUM_SAMPLES=10),
                BBOX_COV_LOSS=ns(NAME="negative_log_likelihood" if v["bbox_cov"] else "none", NUM_SAMPLES=1000,
                                 COVARIANCE_TYPE="diagonal" if v["cov_dims"] == 4 else "full"))))


class MaskRecorder:
    """Serves `F.dropout` and remembers which of the reference's modules asked."""

    def __init__(self, head, n_levels, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.masks, self.index = [], []
        self.n_levels, self.offset = n_levels, 0
        self.current = None                      # (subnet, how many times that subnet has been evaluated before)
        self.layer = 0
        self.seq = {0: 0, 1: 0}
        for sid, sub in enumerate((head.cls_subnet, head.bbox_subnet)):
            sub.register_forward_pre_hook(self._enter(sid))

    def _enter(self, sid):
        def hook(module, inputs):
            self.current = (sid, self.seq[sid])
            self.seq[sid] += 1
            self.layer = 0
        return hook

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = torch.rand(x.shape, generator=self.g) >= p
        sid, seq = self.current
        self.index.append([sid, seq, self.layer, self.offset, keep.numel()])
        self.masks.append(keep)
        self.offset += keep.numel()
        self.layer += 1
        return mf.apply_mask(x, keep, p)


def stack_runs(lst, n_levels, runs):
    """The reference's lists run over `features * N`: entry run * L + level -> per level an (N, ...) array."""
    return [torch.cat([lst[r * n_levels + l] for r in range(runs)]).numpy() for l in range(n_levels)]


def run_variant(pr, name, v):
    torch.manual_seed(SEED)
    feats = mf.seeded_features(SEED)
    L = len(feats)
    model = pr.ProbabilisticRetinaNet(make_cfg(v, ReplayBackbone(feats)))
    head = model.head
    assert type(head) is pr.ProbabilisticRetinaNetHead
    out = {}
    # what the reference's constructor leaves behind (PR:447-484)
    ctor = {}
    for k, p in head.state_dict().items():
        ctor[k] = dict(shape=list(p.shape), mean=float(p.double().mean()), std=float(p.double().std()) if p.numel() > 1 else 0.0)
    mf.load_seeded_state(head, SEED)
    inputs = [{"image": torch.zeros(3, mf.LEVELS[0][0] * 8, mf.LEVELS[0][1] * 8)}]

    # --- eval mode ---------------------------------------------------------------------------------------------------
    model.eval()
    with torch.no_grad():
        logits, deltas, logit_vars, delta_covs = head(feats)
        raw = model(inputs, return_anchorwise_output=True)
    for l in range(L):
        out["eval_logits_l%d" % l] = logits[l].numpy()
        out["eval_bbox_reg_l%d" % l] = deltas[l].numpy()
        if logit_vars is not None:
            out["eval_logits_var_l%d" % l] = logit_vars[l].numpy()
        if delta_covs is not None:
            out["eval_bbox_cov_l%d" % l] = delta_covs[l].numpy()
        out["anchors_l%d" % l] = raw["anchors"][l].tensor.numpy()
        for key in ("box_cls", "box_delta", "box_cls_var", "box_reg_var"):
            if raw[key] is not None:
                out["eval_%s_l%d" % (key, l)] = raw[key][l].numpy()
    none_keys = [k for k in ("box_cls_var", "box_reg_var") if raw[k] is None]

    # --- MC mode: model.train() as PI:53-56, N runs through the head only (PR:103-108) -------------------------------------
    index = np.zeros((0, 7), np.int64)
    packed = np.zeros(0, np.uint8)
    if v["dropout_rate"] > 0.0:
        import torch.nn.functional as F
        model.train()
        rec = MaskRecorder(head, L, SEED + 1)
        real = F.dropout
        F.dropout = rec.dropout
        try:
            with torch.no_grad():
                raw = model(inputs, return_anchorwise_output=True, num_mc_dropout_runs=mf.MC_RUNS)
        finally:
            F.dropout = real
        assert len(raw["anchors"]) == L * mf.MC_RUNS
        for key in ("box_cls", "box_delta", "box_cls_var", "box_reg_var"):
            if raw[key] is not None:
                assert len(raw[key]) == L * mf.MC_RUNS
                for l, a in enumerate(stack_runs(raw[key], L, mf.MC_RUNS)):
                    out["mc_%s_l%d" % (key, l)] = a
        # subnet evaluation sequence number -> (evaluation, run, level): per feature the reference evaluates a subnet once for the
        # mean predictor and once more for the variance predictor if there is one (PR:518-523)
        per_feature = {0: 2 if v["cls_var"] else 1, 1: 2 if v["bbox_cov"] else 1}
        rows = []
        for sid, seq, layer, off, n in rec.index:
            feat, ev = divmod(seq, per_feature[sid])
            run, level = divmod(feat, L)
            rows.append([sid, ev, run, level, layer, off, n])
        index = np.asarray(rows, np.int64)
        assert index[:, 2].max() == mf.MC_RUNS - 1 and index[:, 4].max() == mf.NUM_CONVS - 1
        packed = mf.pack_masks(rec.masks)
    meta = dict(variant=name, seed=SEED, channels=mf.CHANNELS, levels=[list(s) for s in mf.LEVELS], mc_runs=mf.MC_RUNS,
                num_classes=mf.NUM_CLASSES, num_anchors=9, ctor=ctor, state_keys=list(head.state_dict().keys()),
                none_outputs=none_keys, features_sha=sha(feats), params_sha=sha(list(head.state_dict().values())),
                reference_lines="PR:95-112, PR:335-361, PR:365-537", **v)
    out["meta"] = np.array(json.dumps(meta))
    out["mask_bits"] = packed
    out["mask_index"] = index
    return out


def main():
    load_reference()
    import importlib
    pr = importlib.import_module("probabilistic_modeling.probabilistic_retinanet")
    os.makedirs(OUT, exist_ok=True)
    for name, v in mf.VARIANTS.items():
        arrays = run_variant(pr, name, v)
        path = os.path.join(OUT, "head_%s.npz" % name)
        np.savez_compressed(path, **arrays)
        print(name, os.path.getsize(path), "bytes", len(arrays), "arrays")


if __name__ == "__main__":
    main()
