"""Import the reference's pure-Python hot-path modules in THIS container.

TEST INFRASTRUCTURE ONLY. `/root/reference` does not exist on the GPU box and is never
read by the product path, `-m gpu` tests, `smoke()` or `bench.py`; this module is used by
`oracle/make_golden.py` (fixture generation) and by CPU tests that are skipped when the
reference tree is absent.

The reference needs detectron2 / cv2, which are not installed (no network): the
`oracle/refstub` stand-in restates the handful of public detectron2 calls the path uses.
Nothing from `/root/reference` is copied; it is imported where it lies.
"""
import importlib
import os
import sys
import types

REFERENCE_SRC = os.environ.get("POD_REFERENCE_SRC", "/root/reference/src")
_STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refstub")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "probabilistic_inference", "probabilistic_inference.py"))


def load_reference():
    """Returns (probabilistic_inference module, inference_utils module, modeling_utils module)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at {}".format(REFERENCE_SRC))
    if _STUB not in sys.path:
        sys.path.insert(0, _STUB)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(1, REFERENCE_SRC)
    # The reference's `core` package pulls in the visualiser (matplotlib/cv2 GUI); seed dummies.
    for name in ("core", "core.visualization_tools", "core.visualization_tools.probabilistic_visualizer"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["core.visualization_tools.probabilistic_visualizer"].ProbabilisticVisualizer = object
    pi = importlib.import_module("probabilistic_inference.probabilistic_inference")
    iu = importlib.import_module("probabilistic_inference.inference_utils")
    mu = importlib.import_module("probabilistic_modeling.modeling_utils")
    return pi, iu, mu


def load_reference_evaluation():
    """Returns (evaluation_utils module, scoring_rules module) of the reference's core/evaluation_tools (SURVEY f-1).
    `ujson`, `tqdm`-free stand-ins and dummy `core.datasets` modules are seeded; the reference's own files are imported
    by path so that its `core/__init__.py` (project paths) is not needed."""
    import importlib.util
    import json
    load_reference()
    sys.modules.setdefault("ujson", json)
    for name in ("core.datasets", "core.datasets.metadata", "core.evaluation_tools"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["core"].datasets = sys.modules["core.datasets"]
    sys.modules["core.datasets"].metadata = sys.modules["core.datasets.metadata"]
    mods = []
    for fname in ("evaluation_utils", "scoring_rules"):
        path = os.path.join(REFERENCE_SRC, "core", "evaluation_tools", fname + ".py")
        spec = importlib.util.spec_from_file_location("core.evaluation_tools." + fname, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)
