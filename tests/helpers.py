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


def is_predictor_fixture(path) -> bool:
    """A whole-predictor fixture (oracle/make_golden.py) is recognised by WHAT IT HOLDS -- a JSON `meta` member carrying a `spec` --
    never by what its file name is not: any other .npz dropped into tests/golden/ (unit vectors, evaluation fixtures, model fixtures,
    calibration fixtures ...) is simply not one."""
    try:
        with np.load(path, allow_pickle=False) as z:
            if "meta" not in z.files:
                return False
            meta = json.loads(str(z["meta"]))
    except Exception:
        return False
    return isinstance(meta, dict) and "spec" in meta and "name" in meta and "input_sha" in meta


def fixture_paths(prefix=""):
    return sorted(p for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")) if is_predictor_fixture(p))


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


# What the parity bar actually uses: every assert_close call records the worst absolute error and the worst error in units of its
# bound, per (test, quantity); tests/conftest.py writes them out at the end of a GPU session (gpurun_out/parity_errors.{json,md};
# the committed copy is profiles/r04_parity_errors.md).
PARITY_LOG = {}


def _record(what, err, bound, b):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
    k = (test, what)
    flat_e, flat_b, flat_r = err.reshape(-1), bound.reshape(-1), b.reshape(-1)
    i = int(torch.argmax(flat_e / flat_b))
    rec = {"n": int(flat_e.numel()), "max_abs": float(flat_e.max()), "max_rel_to_max1ref": float((flat_e / flat_r.abs().clamp(min=1.0)).max()),
           "worst_fraction_of_bound": float(flat_e[i] / flat_b[i]), "at_ref": float(flat_r[i]), "bound_there": float(flat_b[i])}
    old = PARITY_LOG.get(k)
    if old is None or rec["worst_fraction_of_bound"] > old["worst_fraction_of_bound"]:
        rec["n"] += old["n"] if old else 0
        PARITY_LOG[k] = rec
    else:
        old["n"] += rec["n"]
        old["max_abs"] = max(old["max_abs"], rec["max_abs"])
        old["max_rel_to_max1ref"] = max(old["max_rel_to_max1ref"], rec["max_rel_to_max1ref"])


def assert_close(a, b, what="", rtol=RTOL, atol=ATOL, matrix_scale=False):
    """|a - b| <= max(atol, rtol |b|) element-wise.  With the defaults that is `north_star`'s "within 1e-4 on box means / covariances"
    read RELATIVE TO max(1, |ref|): the quantities are fp32 moments of 1000 samples of coordinates up to ~1 300 px (and covariances up
    to ~1e4 px^2) -- an absolute 1e-4 on a 1 300-px coordinate is below the spacing of fp32 numbers there (1.2e-4), so no fp32
    computation, the reference's own on another BLAS included, could meet it.  How much of the bar is used: profiles/r04_parity_errors.md."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, "{}: shape {} vs {}".format(what, tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return
    err = (a - b).abs()
    bound = torch.clamp(rtol * b.abs(), min=atol)      # defaults: 1e-4 * max(1, |b|), the bar DESIGN.md and smoke() state
    if matrix_scale and b.dim() >= 2:
        # COVARIANCE MATRICES (round 6, found by the full-size index tests): an entry is (a difference of) sums of 1000 fp32 products of the
        # size of the matrix's variances, so its rounding noise scales with the LARGEST entry of its matrix, not with itself -- a p7 anchor's
        # box has variances of ~1 500 px^2 and correlations near 0: an off-diagonal entry of 0.3 carries 2e-4 of summation-order noise (1e-7
        # of the variances; the reference's own sgemm on another BLAS differs by as much).  Bound = rtol * max(1, max |ref matrix|); the
        # element-wise fraction is still recorded (parity_errors.md: `cov ... elementwise`).
        _record(what + " (elementwise reading)", err, bound, b)
        scale = b.abs().amax(dim=(-2, -1), keepdim=True).clamp(min=1.0)
        bound = torch.maximum(bound, (rtol * scale).expand_as(bound))
    _record(what, err, bound, b)
    bad = err > bound
    assert not bool(bad.any()), "{}: {} of {} elements off; worst |d|={:.3e} at ref={:.6g}".format(
        what, int(bad.sum()), a.numel(), float(err.max()), float(b.reshape(-1)[err.reshape(-1).argmax()]))


def write_parity_report(directory):
    if not PARITY_LOG:
        return
    os.makedirs(directory, exist_ok=True)
    rows = [dict(test=t, quantity=w, **r) for (t, w), r in sorted(PARITY_LOG.items())]
    with open(os.path.join(directory, "parity_errors.json"), "w") as f:
        json.dump(rows, f, indent=1)
    by_q = {}
    for r in rows:
        q = by_q.setdefault(r["quantity"], dict(tests=0, n=0, max_abs=0.0, max_rel=0.0, frac=0.0, where=""))
        q["tests"] += 1
        q["n"] += r["n"]
        q["max_abs"] = max(q["max_abs"], r["max_abs"])
        q["max_rel"] = max(q["max_rel"], r["max_rel_to_max1ref"])
        if r["worst_fraction_of_bound"] > q["frac"]:
            q["frac"], q["where"] = r["worst_fraction_of_bound"], r["test"]
    with open(os.path.join(directory, "parity_errors.md"), "w") as f:
        f.write("| quantity | tests | elements | worst abs. error | worst error / max(1, abs ref) | worst fraction of its bound | in |\n|---|---|---|---|---|---|---|\n")
        for k, q in sorted(by_q.items()):
            f.write("| %s | %d | %d | %.3e | %.3e | %.3f | `%s` |\n" % (k, q["tests"], q["n"], q["max_abs"], q["max_rel"], q["frac"], q["where"]))
