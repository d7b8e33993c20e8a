"""Guards for the test suite itself, run on CPU so that the container-side `pytest -q` catches what only `-m gpu` would otherwise
meet on the driver's box (round 4: a stray .npz in tests/golden/ stopped the driver's GPU run at test 86 of 387)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import GPU_ORDER
from tests.helpers import GOLDEN, Golden, fixture_paths, is_predictor_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu_ids():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ids = [l.strip() for l in r.stdout.splitlines() if "::" in l]
    assert len(ids) >= 380, len(ids)
    return ids


def test_every_fixture_a_gpu_parametrisation_names_loads_as_a_predictor_fixture(gpu_ids):
    """Each `[...]` id of a collected GPU test that names a file in tests/golden/ is opened the way the test will open it."""
    names = {os.path.basename(p)[:-4] for p in os.listdir(GOLDEN) if p.endswith(".npz")}
    used = set()
    for i in gpu_ids:
        m = re.search(r"\[(.*)\]$", i)
        if not m:
            continue
        for tok in [m.group(1)] + m.group(1).split("-"):
            if tok in names:
                used.add((i.split("::")[0], tok))
    assert len(used) >= 40, sorted(used)
    for mod, name in sorted(used):
        p = os.path.join(GOLDEN, name + ".npz")
        src = open(os.path.join(ROOT, mod)).read()
        if "Golden(" in src and not name.startswith("head_"):
            g = Golden(p)
            assert g.name and g.spec and is_predictor_fixture(p), (mod, name)
        else:
            with np.load(p, allow_pickle=False) as z:
                assert z.files, (mod, name)


def test_fixture_names_written_into_test_sources_exist_and_load():
    """Fixtures the tests name literally (`"cfg3_bayes_od_mc10_s31.npz"`, `FIXTURE = "..."`) rather than through a parametrisation."""
    pat = re.compile(r"[\"']([A-Za-z0-9_]+)\.npz[\"']")
    seen = 0
    for f in sorted(os.listdir(os.path.join(ROOT, "tests"))):
        if not f.endswith(".py") or f == os.path.basename(__file__):
            continue
        src = open(os.path.join(ROOT, "tests", f)).read()
        for name in set(pat.findall(src)):
            p = os.path.join(GOLDEN, name + ".npz")
            assert os.path.exists(p), (f, name)
            if is_predictor_fixture(p):
                Golden(p)
            seen += 1
    assert seen >= 10


def test_predictor_fixture_selection_is_by_content():
    paths = fixture_paths()
    assert len(paths) == 27, [os.path.basename(p) for p in paths]
    for p in paths:
        g = Golden(p)
        assert g.meta["input_sha"] and "runs" in g.spec
    others = sorted(set(os.listdir(GOLDEN)) - {os.path.basename(p) for p in paths})
    assert all(not is_predictor_fixture(os.path.join(GOLDEN, o)) for o in others if o.endswith(".npz"))
    for k in range(1, 6):      # every BASELINE config has its two small goldens
        assert len(fixture_paths("cfg%d_" % k)) == 2, k
    assert len(fixture_paths("full_cfg")) == 5 and len(fixture_paths("full_worst_cfg3")) == 1      # every BASELINE config at R = 193 374


def test_gpu_run_order_puts_the_contract_first(gpu_ids):
    mods = []
    for i in gpu_ids:
        m = os.path.splitext(os.path.basename(i.split("::")[0]))[0]
        if not mods or mods[-1] != m:
            mods.append(m)
    assert len(mods) == len(set(mods)), mods          # each module is one contiguous stretch
    known = [m for m in mods if m in GPU_ORDER]
    assert known == [m for m in GPU_ORDER if m in known], mods
    assert set(mods) <= set(GPU_ORDER), sorted(set(mods) - set(GPU_ORDER))      # a new GPU test module must be given a place
    first = [i for i in gpu_ids if "test_hip_parity" in i][:12]
    assert all("test_hip_matches_reference_golden[cfg" in i or "[full_cfg" in i or "[full_worst_cfg" in i for i in first), first
