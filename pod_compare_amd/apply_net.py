"""Inference driver: the reference's apply_net.py loop (AN:82-102), image-sharded over the GPUs of one node.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m pod_compare_amd.apply_net \
        --config-file <model.yaml> --inference-config <inference.yaml> --num-images 64 --output results.json

The reference pins inference to one process (AN:113-114).  Here rank r handles images r, r+world, ...
(independent units, batch 1 as AN:35, a few of them in flight on separate HIP streams); each rank keeps its detections
as fixed-stride records in HBM (K7) and ONE collective per flush (every --flush-every images per rank) gathers counts +
records to every rank (`all_gather`; backend "nccl" =
RCCL over xGMI on GPUs, "gloo" in the CPU tests of this sharding/re-ordering logic).  Rank 0 restores the
image order and writes `coco_instances_results.json` (AN:100-102 format).
"""
import argparse
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .inference_utils import record_width, records_to_json

BDD_CAT_MAP = {i: i + 1 for i in range(7)}   # contiguous id -> dataset id (core/datasets/metadata.py, ids 1..7)


def shard_indices(num_images: int, rank: int, world: int) -> List[int]:
    """Images of rank `rank`: i = rank (mod world)."""
    return list(range(rank, num_images, world))


def gather_records(local_ids: Sequence[int], local_counts: torch.Tensor, local_records: torch.Tensor, num_images: int,
                   world: int, per_rank: Optional[int] = None) -> Tuple[List[int], torch.Tensor, torch.Tensor]:
    """One flush: every rank contributes (n_local, max_det, width) records + (n_local,) counts, padded to the
    common per-rank maximum (`per_rank`, default ceil(num_images / world)) so a single all_gather moves them.
    Returns (image ids, counts, records) in image order."""
    if per_rank is None:
        per_rank = -(-num_images // world)
    n_local = len(local_ids)
    dev = local_records.device
    ids = torch.full((per_rank,), -1, dtype=torch.int64, device=dev)
    ids[:n_local] = torch.as_tensor(list(local_ids), dtype=torch.int64, device=dev)
    cnt = torch.zeros((per_rank,), dtype=torch.int32, device=dev)
    cnt[:n_local] = local_counts.to(torch.int32)
    rec = torch.zeros((per_rank,) + tuple(local_records.shape[1:]), dtype=local_records.dtype, device=dev)
    rec[:n_local] = local_records
    if world > 1:
        all_ids = [torch.empty_like(ids) for _ in range(world)]
        all_cnt = [torch.empty_like(cnt) for _ in range(world)]
        all_rec = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(all_ids, ids)
        dist.all_gather(all_cnt, cnt)
        dist.all_gather(all_rec, rec)
        ids, cnt, rec = torch.cat(all_ids), torch.cat(all_cnt), torch.cat(all_rec)
    valid = ids >= 0
    ids, cnt, rec = ids[valid], cnt[valid], rec[valid]
    order = torch.argsort(ids)
    return ids[order].tolist(), cnt[order], rec[order]


def results_json(ids: Sequence[int], counts: torch.Tensor, records: torch.Tensor, num_classes: int,
                 cat_map: Optional[Dict[int, int]] = None) -> List[dict]:
    out = []
    counts = counts.cpu().tolist()
    records = records.cpu()
    for i, image_id in enumerate(ids):
        out.extend(records_to_json(records[i], counts[i], image_id, num_classes, cat_map))
    return out


def _run_ensemble_per_gpu(cfg, args, rank, world):
    """Config 5 (`ensembles_pre_nms.yaml`): one ensemble member per rank, point-to-point exchange of the dense head
    tensors onto the image's merge rank (= the rank that owns the image in the sharded order), K1..K7 there."""
    from . import anchors, ensemble_dist, modeling, synthetic
    from .probabilistic_inference import RetinaNetProbabilisticPredictor, build_model
    seeds = list(cfg.PROBABILISTIC_INFERENCE.ENSEMBLES.RANDOM_SEED_NUMS)
    M = len(seeds)
    if world < M:
        raise SystemExit("--ensemble-per-gpu needs at least %d ranks (one per ensemble member), got %d" % (M, world))
    model = None
    if rank < M:
        torch.manual_seed(int(seeds[rank]))          # PI:59-77 loads random_seed_<s> checkpoints; random-init stands in
        model = build_model(cfg)
    cfg.PROBABILISTIC_INFERENCE.ENSEMBLES.BOX_MERGE_MODE = "pre_nms"
    predictor = RetinaNetProbabilisticPredictor(cfg, model=model if model is not None else object(), model_list=[object()] * M)
    if model is None:
        # merge-only rank: the predictor only needs the model's test-time attributes
        predictor.model = build_model(cfg)
    dev = torch.device(cfg.MODEL.DEVICE)
    net_hw = anchors.resize_shortest_edge(720, 1280, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    recs, cnts = [], []
    layout = stacked = like = None
    with torch.no_grad():
        for i in range(args.num_images):
            dst = ensemble_dist.merge_rank(i, world)
            frame = synthetic.synthetic_frame(i, device=dev)
            image = modeling.resize_test_image(frame, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
            packed = None
            ho = None
            if rank < M:
                ho = model(image)
            if layout is None:
                like = ho if ho is not None else predictor.model(image)
                layout = ensemble_dist.MemberLayout.of(like)
                stacked = torch.empty((M, layout.total), dtype=torch.float32, device=dev)
            if rank < M:
                packed = layout.pack(ho)
            ensemble_dist.exchange_members(packed, stacked if rank == dst else None, M, dst, rank)
            if rank == dst:
                input_im = [{"image": image, "height": frame.shape[1], "width": frame.shape[2], "image_id": i}]
                predictor._run("standard_nms", input_im, layout.views(stacked, like))     # PI:502-505
                recs.append(predictor.last_detections.records)
                cnts.append(predictor.last_detections.n_det)
    return recs, cnts


def main(argv=None):
    from . import synthetic
    from .config import setup_config
    from .probabilistic_inference import build_predictor
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--config-file", default=os.path.join(here, "configs/BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml"))
    ap.add_argument("--inference-config", default=os.path.join(here, "configs/Inference/bayes_od_mc_dropout.yaml"))
    ap.add_argument("--num-images", type=int, default=8)
    ap.add_argument("--random-seed", type=int, default=0)
    ap.add_argument("--output", default="coco_instances_results.json")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams per GPU; consecutive images of a rank go to different streams (batch 1 per stream, AN:35)")
    ap.add_argument("--flush-every", type=int, default=64,
                    help="images per rank between two gathers of the device-resident records (SURVEY 8e: ~0.74 MB per rank and flush)")
    ap.add_argument("--ensemble-per-gpu", action="store_true",
                    help="BASELINE config 5: rank s < M runs ensemble member s, dense pre-NMS tensors meet on a rotating merge rank")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    cfg = setup_config(args.config_file, args.inference_config, args.random_seed)
    cfg.MODEL.DEVICE = "cuda:%d" % local_rank
    K = cfg.MODEL.RETINANET.NUM_CLASSES
    from . import modeling
    mine = shard_indices(args.num_images, rank, world)
    recs, cnts = [], []
    if args.ensemble_per_gpu:
        recs, cnts = _run_ensemble_per_gpu(cfg, args, rank, world)
    torch.manual_seed(args.random_seed)
    predictor = build_predictor(cfg) if not args.ensemble_per_gpu else None
    if predictor is not None:
        predictor.return_device = True      # no per-image host sync: records and counts stay in HBM until the gather
    # images are independent units: keep a few in flight on separate HIP streams so one image's low-occupancy backbone
    # stretches overlap another image's head convs (+12 % images/s on one MI355X); the predictor keeps a workspace per stream
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(max(1, args.streams) - 1)]
    width = record_width(K)
    dev = cfg.MODEL.DEVICE
    F = max(1, args.flush_every)
    n_mine = len(shard_indices(args.num_images, 0, world))          # rank 0 owns the most images: every rank flushes as often
    all_ids: List[int] = []
    all_cnt, all_rec = [], []

    def flush(chunk_ids, recs, cnts):
        """ONE collective: this chunk's records of every rank, image order restored (device memory stays bounded)."""
        for st in streams[1:]:
            streams[0].wait_stream(st)          # the gather reads every stream's records
        rec = torch.stack(recs) if recs else torch.zeros((0, 128, width), device=dev)
        cnt = torch.stack(cnts) if cnts else torch.zeros((0,), dtype=torch.int32, device=dev)
        ids, cnt, rec = gather_records(chunk_ids, cnt, rec, args.num_images, world, per_rank=F)
        all_ids.extend(ids)
        all_cnt.append(cnt.cpu())
        all_rec.append(rec.cpu())

    if args.ensemble_per_gpu:
        flush_ids = list(mine)
        for a in range(0, max(n_mine, 1), F):
            flush(flush_ids[a:a + F], recs[a:a + F], cnts[a:a + F])
    else:
        chunk_ids: List[int] = []
        recs, cnts = [], []
        with torch.no_grad():
            for j in range(n_mine):
                if j < len(mine):
                    i = mine[j]
                    with torch.cuda.stream(streams[j % len(streams)]):
                        frame = synthetic.synthetic_frame(i, device=dev)
                        image = modeling.resize_test_image(frame, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
                        input_im = [{"image": image, "height": frame.shape[1], "width": frame.shape[2], "image_id": i}]
                        det = predictor(input_im)
                        chunk_ids.append(i)
                        recs.append(det.records)
                        cnts.append(det.n_det)
                    if os.environ.get("POD_SYNC_EACH_IMAGE") == "1":    # debugging aid: serialise the streams
                        torch.cuda.synchronize()
                if (j + 1) % F == 0 or j + 1 == n_mine:
                    flush(chunk_ids, recs, cnts)
                    chunk_ids, recs, cnts = [], [], []
    ids = all_ids
    order = sorted(range(len(ids)), key=lambda q: ids[q])
    ids = [ids[q] for q in order]
    cnt = torch.cat(all_cnt)[order] if all_cnt else torch.zeros((0,), dtype=torch.int32)
    rec = torch.cat(all_rec)[order] if all_rec else torch.zeros((0, 128, width))
    if rank == 0:
        with open(args.output, "w") as fp:
            json.dump(results_json(ids, cnt, rec, K, BDD_CAT_MAP), fp, indent=4, separators=(",", ": "))
        print("wrote %s: %d images, %d detections" % (args.output, len(ids), int(cnt.sum())))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
