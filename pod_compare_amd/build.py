"""Builds the gfx950 HIP library of the hot path in-tree (pod_compare_amd/lib/libpod_mi355x.so).

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  `python -m pod_compare_amd.build [--force]`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpod_mi355x.so")
SOURCES = ["k1_mc_merge_score.hip", "k1f_merge_score_fused.hip", "k2_topk_gather.hip", "k3_decode_cov.hip", "k4_nms.hip", "k5_cluster_merge.hip",
           "k8_model_ops.hip", "k9_eval_match.hip", "k10_debug_dump.hip", "k11_wino_conv.hip", "k12_wino_conv_split.hip", "k13_conv1x1_split.hip", "k14_stem_conv.hip", "k15_sparse_blocks.hip", "pod_run.hip"]
HEADERS = [os.path.join(CSRC, "pod_device.h"), os.path.join(CSRC, "pod_candidate.h"), os.path.join(CSRC, "pod_wino.h"), os.path.join(os.path.dirname(HERE), "include", "pod_mi355x.h"),
           os.path.join(os.path.dirname(HERE), "include", "pod_mi355x_test.h"), os.path.join(CSRC, "pod_experiments.h")]
# -ffp-contract=off: the CPU reference rounds after every op; index parity needs the same fp32 values.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]


# per-source extras.  k12: its transform arithmetic is slotted behind bf16 MFMAs, where a packed fp32 instruction (what the SLP vectoriser
# makes of neighbouring scalar operations) costs ~40 cycles and a scalar one nothing (tools/mfma_bf16_split.hip)
SOURCE_FLAGS = {"k12_wino_conv_split.hip": ["-fno-slp-vectorize"], "k16_wino_conv_split8.hip": ["-fno-slp-vectorize"], "k13_conv1x1_split.hip": ["-fno-slp-vectorize"], "k14_stem_conv.hip": ["-fno-slp-vectorize"]}


def _flags(tagged: bool):
    """Experiment / diagnostics defines apply to TAGGED builds only: the shipped library is always the plain source.
    POD_EXTRA_DEFINES="-DPOD_WINO_ELIM=3 ..."; POD_TRACE=1: phase time stamps inside kernels (csrc/pod_device.h, k11)."""
    f = list(FLAGS)
    if tagged and os.environ.get("POD_EXTRA_DEFINES"):
        f.extend(os.environ["POD_EXTRA_DEFINES"].split())
    if tagged and os.environ.get("POD_TRACE") == "1":
        f.append("-DPOD_TRACE")
    return f


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, tag: str = "") -> str:
    """tag (or env POD_BUILD_TAG): an experiment / diagnostics build kept beside the shipped one in lib/<tag>/ (load it with
    POD_MI355X_LIB=pod_compare_amd/lib/<tag>/libpod_mi355x.so); the shipped library is only ever built without a tag.
    POD_TAG_SOURCES="k11_wino_conv.hip ..." limits what a tagged build recompiles (with its extra defines); the other
    objects are the shipped ones."""
    shipped = tag == "__shipped__"
    tag = "" if shipped else (tag or os.environ.get("POD_BUILD_TAG", ""))
    if not tag and not shipped and (os.environ.get("POD_TRACE") == "1" or os.environ.get("POD_EXTRA_DEFINES")):
        raise RuntimeError("POD_TRACE / POD_EXTRA_DEFINES only apply to tagged builds (set POD_BUILD_TAG=<name>; load the result with "
                           "POD_MI355X_LIB=pod_compare_amd/lib/<name>/libpod_mi355x.so): the shipped library is always the plain source")
    libdir = os.path.join(LIBDIR, tag) if tag else LIBDIR
    lib = os.path.join(libdir, "libpod_mi355x.so")
    os.makedirs(libdir, exist_ok=True)
    only = os.environ.get("POD_TAG_SOURCES", "").split() if tag else []
    if only:
        build_library(verbose=verbose, tag="__shipped__")
    objs = []
    sources = list(SOURCES)
    k16 = bool(tag) and os.environ.get("POD_WITH_K16") == "1"
    if k16:      # round 6's experiment (tools/experiments/k16_wino_conv_split8.hip: the eight-wavefront form of pod_wino_conv3x3_split; measured, not shipped)
        sources.append(os.path.join("..", "..", "tools", "experiments", "k16_wino_conv_split8.hip"))
    for src in sources:
        s = os.path.join(CSRC, src)
        src = os.path.basename(src)
        o = os.path.join(libdir if (not only or src in only) else LIBDIR, src.replace(".hip", ".o"))
        if only and src not in only:
            objs.append(o)
            continue
        if force or _stale(o, [s] + HEADERS):
            cmd = [_hipcc()] + _flags(bool(tag)) + (["-DPOD_WITH_K16", "-I" + CSRC] if k16 else []) + SOURCE_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
