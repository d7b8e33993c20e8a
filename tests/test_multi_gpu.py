"""First contact with RCCL before the driver's scaling run.  One test runs everywhere: bench.py's multi-rank code path on backend nccl
with ONE rank (POD_BENCH_FORCE_DIST=1 under torch.distributed.run: process group on device_id, device check, flush all_gather, barriers).
The others are skipped unless the box exposes >= 2 GPUs (the gpurun boxes have one).
bench.py --gpus 2 on backend nccl (image sharding, one all_gather flush), apply_net on two nccl ranks, and
config 5 with one ensemble member per rank over RCCL point-to-point (>= 6 GPUs; the batch_isend_irecv path the gloo tests cannot see)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
ENV = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")


def bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=ENV, timeout=1500, check=True,
                         stdout=subprocess.PIPE, universal_newlines=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_bench_multi_rank_path_on_nccl_with_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(34500 + os.getpid() % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-diagnostics"]
    out = subprocess.run(cmd, cwd=ROOT, env=dict(ENV, POD_BENCH_FORCE_DIST="1"), timeout=900, check=True, stdout=subprocess.PIPE,
                         universal_newlines=True).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "nccl" and line["config"]["rank_devices"] == [0]
    assert line["flush_ms"] is not None and line["flush_ms"] < 1000.0 and len(line["per_rank_images_per_s"]) == 1 and line["value"] > 0


@pytest.mark.skipif(N_GPU < 2, reason="needs 2 GPUs (RCCL over xGMI)")
def test_bench_two_ranks_on_nccl():
    one = bench("--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-diagnostics")
    two = bench("--gpus", "2", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-diagnostics")
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and two["config"]["collective_backend"] == "nccl"
    assert sorted(two["config"]["rank_devices"]) == [0, 1]                       # every rank its own GPU
    assert len(two["per_rank_images_per_s"]) == 2 and two["flush_ms"] is not None
    assert two["value"] > 1.5 * one["value"]                                      # weak scaling: two GPUs, twice the images


@pytest.mark.skipif(N_GPU < 2, reason="needs 2 GPUs (RCCL over xGMI)")
def test_apply_net_two_ranks_on_nccl(tmp_path):
    out = str(tmp_path / "two.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(36500 + os.getpid() % 2000), "-m", "pod_compare_amd.apply_net", "--num-images", "6", "--random-init",
           "--output", out, "--flush-every", "2"]
    subprocess.check_call(cmd, cwd=ROOT, env=ENV, timeout=900)
    ids = [d["image_id"] for d in json.load(open(out))]
    assert ids == sorted(ids) and set(ids) <= set(range(6))


@pytest.mark.skipif(N_GPU < 6, reason="needs 6 GPUs: the fixture's five member ranks + a merge-only rank")
def test_ensemble_per_gpu_on_nccl(tmp_path):
    """tests/test_ensemble_dist_gpu.py's pipeline on real devices: every member on its own GPU, the packed rows over RCCL
    point-to-point (batch_isend_irecv on device buffers, rotating destination), every merge rank reproduces the reference."""
    from tests import test_ensemble_dist_gpu as t
    t.run_pipeline(tmp_path, backend="nccl", own_gpu=True)
