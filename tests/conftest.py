import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """On a GPU box: what the parity tolerances actually used (tests/helpers.py: PARITY_LOG) -> gpurun_out/parity_errors.{json,md}."""
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from tests import helpers
        helpers.write_parity_report(os.path.join(ROOT, "gpurun_out"))
    except Exception as e:  # pragma: no cover  (a report must never fail a test run)
        print("parity report not written:", e)
