"""BASELINE configs[4] in its 8-GPU topology, end to end on ONE GPU: 6 processes (5 ensemble members + 1 merge-only rank,
all on cuda:0) run apply_net.EnsemblePerGpu -- member forward, packed rows exchanged through the pipelined point-to-point
schedule (gloo, rows staged through pinned host memory because gloo cannot move device buffers; RCCL moves them directly),
K1 streaming the received (5, packed) buffer in place as run-strided views, K2..K7 on the rotating merge rank.
Members are deterministic stand-ins that return run s of the golden cfg5 fixture, so every image's detections must equal
the REFERENCE's recorded output for that fixture (PI:483-505), whichever rank merged it."""
import json
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import GOLDEN, Golden

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pod_compare_amd", "configs")
FIXTURE = "cfg5_ensembles_pre_nms_s51"


class _Member:
    """Ensemble member `s`: returns run s of the fixture's head tensors (what PI:498-500 gets from model s)."""

    def __init__(self, ho, s):
        from pod_compare_amd.probabilistic_inference import run_slice
        self.out = run_slice(ho, s)
        self.cls_var_num_samples, self.test_topk_candidates, self.test_score_thresh = 10, 1000, 0.05
        self.test_nms_thresh, self.max_detections_per_image = 0.5, 100

    def __call__(self, image):
        return self.out


class _RestartingEps:
    """The fixture's replay stream, restarted whenever a new image begins (first request = the level-0 class draws)."""

    def __init__(self, g):
        self.g, self.src, self.first = g, None, None

    def __call__(self, shape):
        shape = tuple(shape)
        if self.first is None:
            self.first = shape
        if shape == self.first:
            self.src = self.g.eps_source()
        return self.src(shape)


def _worker(rank, world, port, num_images, tmp, backend="gloo", own_gpu=False):
    from pod_compare_amd import apply_net, config
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev_index = rank if own_gpu else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    g = Golden(os.path.join(GOLDEN, FIXTURE + ".npz"))
    cfg = config.setup_config(CFG + "/BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var.yaml", CFG + "/Inference/ensembles_pre_nms.yaml")
    cfg.MODEL.DEVICE = "cuda:%d" % dev_index
    M = g.spec["runs"]
    ho = g.head_outputs().to("cuda") if rank < M else None
    runner = apply_net.EnsemblePerGpu(cfg, rank, world, frame_hw=tuple(g.meta["out"]), net_hw=tuple(g.meta["image"]),
                                      model=_Member(ho, rank) if rank < M else None)
    assert runner.pipe.host_staged == (backend != "nccl") and (runner.model is None) == (rank >= M)
    runner.predictor.eps_fn = _RestartingEps(g)
    dummy = torch.zeros((3,) + tuple(g.meta["out"]), dtype=torch.uint8, device="cuda")
    results = {}

    def grab(i, det):
        m = det.count()
        results[i] = dict(boxes=det.boxes[:m].cpu().tolist(), cov=det.cov[:m].cpu().tolist(), scores=det.scores[:m].cpu().tolist(),
                          classes=det.classes[:m].cpu().tolist())

    with torch.no_grad():
        ids, recs, cnts = runner.run(num_images, lambda i: dummy, on_detections=grab)
    assert sorted(results) == ids == [i for i in range(num_images) if i % world == rank]
    with open(os.path.join(tmp, "det_%d.json" % rank), "w") as f:
        json.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


def run_pipeline(tmp_path, backend="gloo", own_gpu=False):
    g = Golden(os.path.join(GOLDEN, FIXTURE + ".npz"))
    world, num_images = g.spec["runs"] + 1, 8
    port = 34500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, num_images, str(tmp_path), backend, own_gpu), nprocs=world, join=True)
    ref_b, ref_c, ref_s, ref_k = g.t("pred_boxes"), g.t("pred_boxes_covariance"), g.t("scores"), g.t("pred_classes")
    seen = []
    for r in range(world):
        out = json.load(open(tmp_path / ("det_%d.json" % r)))
        for i, d in out.items():
            seen.append(int(i))
            assert d["classes"] == ref_k.tolist(), (r, i)
            b, c, s = torch.tensor(d["boxes"]), torch.tensor(d["cov"]), torch.tensor(d["scores"])
            assert float((b - ref_b).abs().max()) <= 1e-4 * max(1.0, float(ref_b.abs().max()))
            assert bool(((c - ref_c).abs() <= 1e-4 * ref_c.abs().clamp(min=1.0)).all())
            assert float((s - ref_s).abs().max()) <= 2e-6
    assert sorted(seen) == list(range(num_images))


def test_one_member_per_rank_pipeline_reproduces_the_reference_on_every_merge_rank(tmp_path):
    run_pipeline(tmp_path)
