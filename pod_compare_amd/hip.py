"""ctypes binding of the C ABI declared in include/pod_mi355x.h.

The product path has NO CPU fallback: if the HIP library is missing or a kernel entry point
returns an error, a RuntimeError is raised.  Tensors are passed as raw device pointers
(`tensor.data_ptr()`), the stream as the current torch HIP stream handle -- torch is plumbing
(device memory + streams), the kernels are ours.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p
from typing import Optional

import torch

from .build import LIB

POD_ABI_VERSION = 14
POD_MAX_LEVELS = 8
POD_MAX_CLASSES = 16
POD_MAX_RUNS = 64
POD_MAX_TOPK = 2048
POD_MAX_PROP_SAMPLES = 1024
POD_MAX_CLS_SAMPLES = 64
POD_MAX_CANDIDATES = 8192
POD_MAX_DETECTIONS = 128

EXPORTS = ("pod_abi_version", "pod_mc_merge_score", "pod_maybe_words", "pod_score_maybe", "pod_merge_score_fused", "pod_reset_counters", "pod_level_topk", "pod_gather_candidates", "pod_gather_decode",
           "pod_decode_cov", "pod_nms_scratch_bytes", "pod_nms_cluster", "pod_bayes_fuse", "pod_anchor_stats_merge",
           "pod_ensemble_append", "pod_ensemble_merge",
           "pod_finalize", "pod_reg_nll", "pod_relu_dropout", "pod_bias_act", "pod_bias_act_to_nchw", "pod_bias_act_to_nhwc", "pod_expand_dropout", "pod_match_groundtruth", "pod_run_image", "pod_run_image_part",
           "pod_absmax", "pod_wino_filter_transform", "pod_wino_conv3x3", "pod_wino_filter_split_bytes", "pod_wino_filter_transform_split", "pod_wino_conv3x3_split", "pod_sparse_reach", "pod_sparse_live_blocks", "pod_wino_reduce", "pod_conv1x1_filter_split_bytes", "pod_conv1x1_filter_split", "pod_conv1x1_split", "pod_reduce_partials", "pod_stem7x7_filter_split", "pod_stem7x7_split", "pod_maxpool3x3s2_cl", "pod_im2col3x3s2_cl")
# include/pod_mi355x_test.h: test support (the dumps of the in-kernel draws and of the f16 split) -- exported for tests/ and tools/, not part of the boundary
TEST_EXPORTS = ("pod_dump_cls_normals", "pod_dump_box_normals", "pod_debug_f16_split2")
POD_MODE_STANDARD_NMS, POD_MODE_BAYES_OD, POD_MODE_ANCHOR_STATISTICS = 0, 1, 2


class PodLevel(Structure):
    _fields_ = [("cls", c_void_p), ("cls_var", c_void_p), ("delta", c_void_p), ("reg_var", c_void_p), ("eps_cls", c_void_p),
                ("run_stride_cls", c_int64), ("run_stride_delta", c_int64), ("run_stride_reg", c_int64),
                ("H", c_int32), ("W", c_int32), ("anchor_base", c_int32), ("reserved", c_int32)]


class PodConfig(Structure):
    _fields_ = [("n_levels", c_int32), ("n_runs", c_int32), ("num_anchors", c_int32), ("num_classes", c_int32),
                ("cov_dims", c_int32), ("has_cls_var", c_int32), ("merge_quirk", c_int32), ("cls_samples", c_int32),
                ("prop_samples", c_int32), ("topk", c_int32), ("max_detections", c_int32),
                ("score_thresh", c_float), ("nms_thresh", c_float), ("affinity_thresh", c_float),
                ("box_weights", c_float * 4), ("philox_seed", c_uint64)]


class PodWorkspace(Structure):
    """include/pod_mi355x.h: PodWorkspace (device pointers of the per-geometry workspace, caller-owned)."""
    _fields_ = [(n, c_void_p) for n in (
        "anchors", "mean_cls", "mean_cls_var", "mean_delta", "mean_reg_var", "cand_keys", "cand_count", "maybe_bits",
        "sel_keys", "sel_count", "cat_keys", "cat_level", "probs_dense", "n_total", "cand_anchor_idx", "cand_level", "cand_class", "cand_score", "cand_probs",
        "cand_delta", "cand_reg_var", "cand_anchor", "cand_run_delta", "boxes", "cov", "keep", "n_keep", "nms_scratch",
        "m_boxes", "m_cov", "m_scores", "m_classes", "m_probs")] + [("n_capacity", c_int32), ("reserved", c_int32)]


class PodDetections(Structure):
    """include/pod_mi355x.h: PodDetections (pod_finalize's outputs)."""
    _fields_ = [(n, c_void_p) for n in ("boxes", "cov", "scores", "classes", "probs", "records", "n_det")]


class PodConvSet(Structure):
    """include/pod_mi355x.h: PodConvSet (one convolution of a pod_wino_conv3x3_split launch)."""
    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("Us", c_void_p), ("bias", c_void_p), ("in_amax", c_void_p), ("out_amax", c_void_p),
                ("offset", c_uint64), ("first_block", c_int32), ("replicas", c_int32), ("k_planes", c_int32), ("reserved", c_int32)]


class PodWinoConv(Structure):
    """include/pod_mi355x.h: PodWinoConv."""
    _fields_ = [("blocks", c_void_p), ("n_blocks", c_int32), ("n_sets", c_int32), ("C", c_int32), ("K", c_int32), ("relu", c_int32), ("p", c_float),
                ("seed", c_uint64), ("epoch", c_void_p), ("n_splits", c_int32), ("form", c_int32), ("split_stride", c_int64), ("live_blocks", c_void_p),
                ("sets", PodConvSet * 4)]


class PodError(RuntimeError):
    pass


_lib = None


def library_path() -> str:
    return os.environ.get("POD_MI355X_LIB", LIB)


def load() -> ctypes.CDLL:
    """Loads libpod_mi355x.so (built by `python -m pod_compare_amd.build`); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise PodError("HIP library {} not found: run `python -m pod_compare_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback for the hot path.".format(path))
    lib = ctypes.CDLL(path)
    for name in EXPORTS + TEST_EXPORTS:
        if not hasattr(lib, name):
            raise PodError("{} does not export {}".format(path, name))
    P = c_void_p
    lib.pod_abi_version.restype = ctypes.c_int
    lib.pod_nms_scratch_bytes.restype = c_size_t
    lib.pod_nms_scratch_bytes.argtypes = [c_int32]
    lib.pod_reset_counters.argtypes = [P, c_int32, P]
    lib.pod_mc_merge_score.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, P, P, P, P, P, P, P]
    lib.pod_maybe_words.argtypes = [POINTER(PodConfig), POINTER(PodLevel)]
    lib.pod_maybe_words.restype = c_int64
    lib.pod_score_maybe.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, P, P, P, P, P, P]
    lib.pod_merge_score_fused.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, P, P, P, P, P]
    lib.pod_level_topk.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, P, P, P, P, P, P, P]
    lib.pod_gather_candidates.argtypes = [POINTER(PodConfig), POINTER(PodLevel)] + [P] * 16
    lib.pod_gather_decode.argtypes = [POINTER(PodConfig), POINTER(PodLevel)] + [P] * 19
    lib.pod_decode_cov.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, c_int32, P, P, P, P, P, P, P, c_int32, P, P, P]
    lib.pod_nms_cluster.argtypes = [POINTER(PodConfig), P, c_int32, P, P, P, P, P, P, P]
    lib.pod_bayes_fuse.argtypes = [POINTER(PodConfig), P, P, P, P, P, P, P, P, c_int32, c_int32, P, P, P, P, P, P]
    lib.pod_anchor_stats_merge.argtypes = [POINTER(PodConfig), P, P, P, P, P, P, P, P, P, P, P, P, P]
    lib.pod_ensemble_append.argtypes = [POINTER(PodConfig), P, P, P, P, P, P, c_int32, P, P, P, P, P, P]
    lib.pod_ensemble_merge.argtypes = [POINTER(PodConfig), P, c_int32, P, P, P, P, P, P, P, P, P, P, P, P]
    lib.pod_finalize.argtypes = [POINTER(PodConfig)] + [P] * 7 + [c_float] * 4 + [P] * 7 + [P]
    lib.pod_reg_nll.argtypes = [P, P, P, c_int32, P, P]
    lib.pod_match_groundtruth.argtypes = [P, P, P, P, c_int32, P, P, P, c_int32, c_int32, c_float, c_float, P, P, P, P, P, P]
    lib.pod_relu_dropout.argtypes = [P, c_int64, c_float, c_uint64, c_uint64, P]
    lib.pod_bias_act_to_nchw.argtypes = [P, P, P, c_int64, c_int32, c_int64, c_int32, c_float, c_uint64, c_uint64, P]
    lib.pod_bias_act_to_nhwc.argtypes = [P, P, P, c_int64, c_int32, c_int64, c_int32, P]
    lib.pod_expand_dropout.argtypes = [P, P, c_int64, c_int32, c_float, c_uint64, c_uint64, P, P]
    lib.pod_wino_filter_transform.argtypes = [P, P, c_int32, c_int32, P]
    lib.pod_wino_conv3x3.argtypes = [P, P, P, P, P, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_uint64, c_uint64, P, P]
    lib.pod_wino_filter_split_bytes.argtypes = [c_int32, c_int32]
    lib.pod_wino_filter_split_bytes.restype = c_int64
    lib.pod_wino_filter_transform_split.argtypes = [P, P, c_int32, c_int32, P]
    lib.pod_wino_conv3x3_split.argtypes = [POINTER(PodWinoConv), P]
    lib.pod_absmax.argtypes = [P, c_int64, P, P]
    lib.pod_sparse_reach.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, P, P, P, P, P]
    lib.pod_sparse_live_blocks.argtypes = [POINTER(PodConfig), POINTER(PodLevel), P, P, c_int32, P, c_int32, c_int32, P, P]
    lib.pod_debug_f16_split2.argtypes = [P, c_float, P, c_int64, P]
    lib.pod_wino_reduce.argtypes = [P, c_int32, c_int64, P, P, c_int64, c_int32, c_int32, c_int32, P, P]
    lib.pod_reduce_partials.argtypes = [P, c_int32, c_int64, P, P, P, c_int64, c_int32, c_int32, P, P]
    lib.pod_stem7x7_filter_split.argtypes = [P, P, P]
    lib.pod_stem7x7_split.argtypes = [P, c_int32, c_int32, c_int32, P, P, P, P, P, c_int32, c_int32, c_int32, P, P, P]
    lib.pod_maxpool3x3s2_cl.argtypes = [P, P, c_int32, c_int32, c_int32, P]
    lib.pod_im2col3x3s2_cl.argtypes = [P, P, c_int32, c_int32, c_int32, c_int32, P]
    lib.pod_conv1x1_filter_split.argtypes = [P, P, c_int32, c_int32, P]
    lib.pod_conv1x1_split.argtypes = [P, P, P, P, P] + [c_int32] * 9 + [P, c_int32, P, P, P]
    lib.pod_conv1x1_filter_split_bytes.argtypes = [c_int32, c_int32]
    lib.pod_conv1x1_filter_split_bytes.restype = c_int64
    lib.pod_bias_act.argtypes = [P, P, P, P, c_int64, c_int32, c_int64, c_int32, c_float, c_uint64, c_uint64, P]
    lib.pod_dump_cls_normals.argtypes = [POINTER(PodConfig), POINTER(PodLevel), c_int32, P, P]
    lib.pod_dump_box_normals.argtypes = [POINTER(PodConfig), P, c_int32, P, P]
    lib.pod_run_image.argtypes = [POINTER(PodConfig), POINTER(PodLevel), POINTER(PodWorkspace), c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, POINTER(PodDetections), P]
    lib.pod_run_image_part.argtypes = [POINTER(PodConfig), POINTER(PodLevel), POINTER(PodWorkspace), c_int32, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, POINTER(PodDetections), c_int32, P]
    for name in EXPORTS + TEST_EXPORTS:
        if name not in ("pod_abi_version", "pod_nms_scratch_bytes", "pod_maybe_words", "pod_wino_filter_split_bytes", "pod_conv1x1_filter_split_bytes"):
            getattr(lib, name).restype = ctypes.c_int
    if lib.pod_abi_version() != POD_ABI_VERSION:
        raise PodError("ABI version mismatch: library {} vs binding {}".format(lib.pod_abi_version(), POD_ABI_VERSION))
    _lib = lib
    return lib


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "pod_mi355x kernels need contiguous tensors"
    return t.data_ptr()


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise PodError("{} failed with code {} ({})".format(what, rc, {-1: "invalid argument", -2: "HIP launch error"}.get(rc, "?")))
