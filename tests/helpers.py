"""Shared test plumbing: golden-fixture loading, input regeneration, tolerances."""
import glob
import hashlib
import json
import os

import numpy as np
import torch

from pod_compare_amd import synthetic
from pod_compare_amd.anchors import padded_size

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# "within 1e-4 on box means/covariances" (BASELINE.json north_star): |a-b| <= 1e-4 * max(1, |b|)
RTOL = 1e-4
ATOL = 1e-4


def sha(tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def fixture_paths(prefix=""):
    skip = ("unit_functions.npz", "eval_matching.npz", "eval_metrics.npz")     # not whole-predictor fixtures
    return sorted(p for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz"))
                  if not p.endswith(skip) and not os.path.basename(p).startswith("head_"))      # head_*: model fixtures (row a1)


def fixture_id(path):
    return os.path.basename(path)[:-4]


class Golden:
    """One whole-predictor fixture written by oracle/make_golden.py."""

    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"]))
        self.spec = self.meta["spec"]
        self.name = self.meta["name"]

    def t(self, key):
        return torch.from_numpy(self.z[key])

    def has(self, key):
        return key in self.z.files

    def head_outputs(self):
        """Regenerates the seeded inputs and checks them against the stored checksum."""
        s = self.spec
        ho = synthetic.planted_head_outputs(tuple(self.meta["padded"]), s["runs"], seed=self.meta["seed"],
                                            num_boxes=s.get("num_boxes", 8), with_cls_var=s["cls_var"],
                                            with_reg_var=s["reg_var"], cov_dims=s.get("cov_dims", 4),
                                            mode=s.get("synth_mode", "planted"))
        ts = list(ho.cls) + list(ho.delta) + (ho.cls_var or []) + (ho.reg_var or [])
        assert sha(ts) == self.meta["input_sha"], "synthetic generator drifted from the golden fixture"
        return ho

    def eps_source(self):
        return synthetic.SeededNormals(self.meta["eps_seed"])

    def check_eps(self, tensors):
        assert [list(t.shape) for t in tensors] == self.meta["eps_shapes"]
        assert sha(tensors) == self.meta["eps_sha"], "eps stream drifted from the golden fixture"


def assert_close(a, b, what="", rtol=RTOL, atol=ATOL):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, "{}: shape {} vs {}".format(what, tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return
    err = (a - b).abs()
    bound = torch.clamp(rtol * b.abs(), min=atol)      # defaults: 1e-4 * max(1, |b|), the bar DESIGN.md and smoke() state
    bad = err > bound
    assert not bool(bad.any()), "{}: {} of {} elements off; worst |d|={:.3e} at ref={:.6g}".format(
        what, int(bad.sum()), a.numel(), float(err.max()), float(b.reshape(-1)[err.reshape(-1).argmax()]))
