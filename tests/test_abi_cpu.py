"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/pod_mi355x.h declares; the ctypes mirrors have the C layout.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

from pod_compare_amd import build, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pod_mi355x.h")


TEST_HEADER = os.path.join(ROOT, "include", "pod_mi355x_test.h")


def declared_symbols(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|size_t)\s+(pod_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    return build.build_library()


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(hip.EXPORTS)
    assert declared_symbols(TEST_HEADER) == sorted(hip.TEST_EXPORTS)          # test support lives in its own header (round 6)
    assert not set(hip.TEST_EXPORTS) & set(declared_symbols())


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols() + declared_symbols(TEST_HEADER):
        assert hasattr(lib, name), name
    assert lib.pod_abi_version() == hip.POD_ABI_VERSION


def test_binding_loads(lib_path):
    lib = hip.load()
    assert lib.pod_nms_scratch_bytes(5000) >= 4 * 16 * (1 + hip.POD_MAX_DETECTIONS)   # flag + per-class survivor lists
    assert lib.pod_nms_scratch_bytes(0) == 0


def test_struct_layout_matches_c(lib_path, tmp_path):
    """sizeof/offsetof of the ctypes mirrors against the C header (compiled with gcc)."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pod_mi355x.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(PodLevel), offsetof(PodLevel, run_stride_cls),'
                   ' offsetof(PodLevel, H), sizeof(PodConfig), offsetof(PodConfig, score_thresh), offsetof(PodConfig, box_weights),'
                   ' offsetof(PodConfig, philox_seed), sizeof(PodWorkspace), offsetof(PodWorkspace, cand_run_delta),'
                   ' offsetof(PodWorkspace, m_probs), offsetof(PodWorkspace, n_capacity), sizeof(PodDetections),'
                   ' offsetof(PodDetections, n_det));'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(PodConvSet), offsetof(PodConvSet, in_amax), offsetof(PodConvSet, offset),'
                   ' offsetof(PodConvSet, k_planes), sizeof(PodWinoConv), offsetof(PodWinoConv, p), offsetof(PodWinoConv, epoch),'
                   ' offsetof(PodWinoConv, split_stride), offsetof(PodWinoConv, sets)); return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(hip.PodLevel), hip.PodLevel.run_stride_cls.offset, hip.PodLevel.H.offset, ctypes.sizeof(hip.PodConfig),
            hip.PodConfig.score_thresh.offset, hip.PodConfig.box_weights.offset, hip.PodConfig.philox_seed.offset,
            ctypes.sizeof(hip.PodWorkspace), hip.PodWorkspace.cand_run_delta.offset, hip.PodWorkspace.m_probs.offset,
            hip.PodWorkspace.n_capacity.offset, ctypes.sizeof(hip.PodDetections), hip.PodDetections.n_det.offset,
            ctypes.sizeof(hip.PodConvSet), hip.PodConvSet.in_amax.offset, hip.PodConvSet.offset.offset, hip.PodConvSet.k_planes.offset,
            ctypes.sizeof(hip.PodWinoConv), hip.PodWinoConv.p.offset, hip.PodWinoConv.epoch.offset, hip.PodWinoConv.split_stride.offset,
            hip.PodWinoConv.sets.offset]
    assert got == want


def test_invalid_arguments_are_rejected_without_a_gpu(lib_path):
    """Argument validation happens on the host before any launch: safe to exercise on CPU."""
    lib = hip.load()
    cfg = hip.PodConfig()
    assert lib.pod_reset_counters(None, 4, None) == -1
    assert lib.pod_level_topk(cfg, None, None, None, None, None, None, None, None, None) == -1
    assert lib.pod_reg_nll(None, None, None, 3, None, None) == -1
    assert lib.pod_run_image(cfg, None, None, 0, 0, 0, 10, 10, 10, 10, None, None) == -1
    assert lib.pod_nms_cluster(cfg, None, 8, None, None, None, None, None, None, None) == -1
    d = hip.PodWinoConv()
    assert lib.pod_wino_conv3x3_split(ctypes.byref(d), None) == -1 and lib.pod_wino_conv3x3_split(None, None) == -1
    assert lib.pod_absmax(None, 4, None, None) == -1
    assert lib.pod_wino_filter_split_bytes(64, 32) == 2 * 24 * 64 * 32 * 2 + 16 and lib.pod_wino_filter_split_bytes(64, 8) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setenv("POD_MI355X_LIB", "/nonexistent/libpod_mi355x.so")
    monkeypatch.setattr(hip, "_lib", None)
    with pytest.raises(hip.PodError):
        hip.load()
