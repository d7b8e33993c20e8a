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
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .inference_utils import record_width, records_to_json

# core/datasets/metadata.py as DATA: the thing classes of the data sets the reference registers (setup_datasets.py:36-117); dataset ids are
# 1 .. len(classes) in this order (metadata.py:9-15), contiguous ids 0 .. len - 1.  Lyft re-uses BDD's map (setup_datasets.py:117).
THING_CLASSES = {"bdd": ["car", "bus", "truck", "person", "rider", "bike", "motor"], "kitti": ["car", "person"]}
THING_CLASSES["lyft"] = THING_CLASSES["bdd"]


def _family(dataset: str) -> str:
    for fam in THING_CLASSES:
        if dataset.startswith(fam + "_") or dataset == fam:
            return fam
    raise ValueError("Unknown data set {!r}: known families are {} (e.g. bdd_val, kitti_val).".format(dataset, sorted(THING_CLASSES)))


def category_mapping(train_dataset: str, test_dataset: str) -> Dict[int, int]:
    """AN:52-80: model (contiguous, TRAIN data set) class id -> category_id of the TEST data set's annotations.
    Same classes (BDD -> BDD, BDD -> Lyft): the test set's contiguous -> dataset id map flipped (AN:60-63).  BDD -> KITTI: only the classes
    KITTI annotates survive, through metadata.BDD_TO_KITTI_CONTIGUOUS_ID (AN:70-73, 78-79); a detection of any other class gets -1 and is
    dropped by instances_to_json (IU:466-471).  Any other pair: the reference BUILDS a ValueError without raising it (AN:76-77)
    and then fails on an undefined name; here it is raised.  (COCO / VOC: their 80- / 20-class tables are not carried -- out of BASELINE's
    scope -- and are reported as such.)"""
    train, test = _family(train_dataset), _family(test_dataset)
    test_ids = {i: i + 1 for i in range(len(THING_CLASSES[test]))}               # contiguous -> dataset id of the test set
    if THING_CLASSES[train] == THING_CLASSES[test]:
        return test_ids
    if train == "bdd" and test == "kitti":
        to_kitti = {THING_CLASSES["bdd"].index(c): THING_CLASSES["kitti"].index(c) for c in THING_CLASSES["kitti"]}    # BDD_TO_KITTI_CONTIGUOUS_ID
        return {bdd: test_ids[kitti] for bdd, kitti in to_kitti.items()}
    raise ValueError("Cannot generate category mapping dictionary. Please check if training and inference datasets are compatible.")


BDD_CAT_MAP = category_mapping("bdd_train", "bdd_val")   # contiguous id -> dataset id (ids 1..7)


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
        if dist.get_backend() != "nccl":      # functional runs on gloo: collectives on host copies (RCCL gathers device buffers)
            ids, cnt, rec = ids.cpu(), cnt.cpu(), rec.cpu()
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


class CocoImages:
    """The reference's `build_detection_test_loader(cfg, dataset_name)` (AN:83-84) for a COCO-format image list, restated
    without detectron2: what detectron2's DatasetMapper(is_train=False) hands the predictor for every entry of `images` --
    the file read with PIL as RGB and flipped to BGR (cfg.INPUT.FORMAT), ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST) applied
    with PIL's bilinear filter on the uint8 HWC array (detectron2's ResizeTransform does exactly that for uint8 images), then a
    (3, H, W) uint8 tensor; 'height' / 'width' are the ORIGINAL size from the json (the output resolution, PI:106-107) and
    'image_id' the dataset's id."""

    def __init__(self, json_path: str, image_root: str, min_size: int = 800, max_size: int = 1333):
        with open(json_path, "r") as f:
            data = json.load(f)
        self.images = list(data["images"])
        self.root, self.min_size, self.max_size = image_root, int(min_size), int(max_size)

    def __len__(self) -> int:
        return len(self.images)

    def image_id(self, i: int):
        return self.images[i]["id"]

    def __getitem__(self, i: int) -> dict:
        import numpy as np
        from PIL import Image
        from . import anchors
        rec = self.images[i]
        with Image.open(os.path.join(self.root, rec["file_name"])) as im:
            im = im.convert("RGB")
            h, w = im.height, im.width
            nh, nw = anchors.resize_shortest_edge(h, w, self.min_size, self.max_size)
            if (nh, nw) != (h, w):
                im = im.resize((nw, nh), Image.BILINEAR)
            arr = np.asarray(im)[:, :, ::-1]                       # RGB -> BGR
        image = torch.as_tensor(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        return {"image": image, "height": int(rec.get("height", h)), "width": int(rec.get("width", w)), "image_id": rec["id"],
                "file_name": rec["file_name"]}


class Prefetched:
    """The worker side of the reference's test loader (detectron2's build_detection_test_loader runs the mapper on
    cfg.DATALOADER.NUM_WORKERS processes): entries `dataset[i]` for `indices`, read / decoded / resized up to `depth` entries ahead on
    `workers` host threads (PIL drops the GIL while it decodes and resizes) and handed out IN ORDER; the uint8 frame is moved to pinned
    memory so that `.to(device, non_blocking=True)` is an asynchronous copy on the image's stream.  One decode thread feeds ~60 frames of
    1280x720 per second; a GPU takes 130 - 650.  workers = 0: the plain loop.  An exception of a worker surfaces at the entry it belongs to."""

    def __init__(self, dataset, indices: Sequence[int], workers: int = 4, depth: int = 0, pin: bool = True):
        self.dataset, self.indices, self.workers = dataset, list(indices), max(0, int(workers))
        self.depth = int(depth) if depth > 0 else 2 * max(1, self.workers)
        self.pin = bool(pin) and torch.cuda.is_available()

    def _load(self, i: int) -> dict:
        d = self.dataset[i]
        if self.pin:
            d["image"] = d["image"].pin_memory()
        return d

    def __len__(self) -> int:
        return len(self.indices)

    def __iter__(self):
        if self.workers == 0:
            for i in self.indices:
                yield i, self._load(i)
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="pod-loader") as pool:
            pending = deque()
            it = iter(self.indices)
            try:
                for i in it:
                    pending.append((i, pool.submit(self._load, i)))
                    if len(pending) >= self.depth:
                        break
                while pending:
                    i, fut = pending.popleft()
                    entry = fut.result()
                    nxt = next(it, None)
                    if nxt is not None:
                        pending.append((nxt, pool.submit(self._load, nxt)))
                    yield i, entry
            finally:
                for _, fut in pending:
                    fut.cancel()


class EnsemblePerGpu:
    """Config 5 (`ensembles_pre_nms.yaml`) on one node: ensemble member s lives on rank s < M, every member rank runs the
    conv net on the SAME image, the dense pre-NMS head tensors meet on the image's merge rank (= the rank that owns the
    image in the sharded order) through `ensemble_dist.MemberPipeline` (two images in flight), K1..K7 run there
    (PI:495-505).  Ranks >= M only merge; they never build a model: the predictor reads the model's test-time
    attributes from `model_test_attributes(cfg)` and the level geometry comes from the anchor generator."""

    def __init__(self, cfg, rank: int, world: int, frame_hw=(720, 1280), net_hw=None, model=None):
        """frame_hw: output resolution ('height' / 'width' of input_im); net_hw: network-input size (default: the test
        transform of frame_hw, AN:83); model: this rank's member model (default: built and loaded per PI:59-77)."""
        from . import anchors, ensemble_dist, modeling
        from .probabilistic_inference import RetinaNetProbabilisticPredictor, build_model, ensemble_member_dir, model_test_attributes
        from .synthetic import HeadOutputs
        seeds = list(cfg.PROBABILISTIC_INFERENCE.ENSEMBLES.RANDOM_SEED_NUMS)
        self.M, self.rank, self.world, self.cfg = len(seeds), rank, world, cfg
        if world < self.M:
            raise SystemExit("one ensemble member per rank needs at least %d ranks, got %d" % (self.M, world))
        self.dev = torch.device(cfg.MODEL.DEVICE)
        self.model = model if rank < self.M else None
        if rank < self.M and self.model is None:
            state = torch.random.get_rng_state()
            torch.manual_seed(int(seeds[rank]))          # a member without a checkpoint is a random-init model seeded with its number
            self.model = build_model(cfg, save_dir=ensemble_member_dir(cfg, seeds[rank]))      # PI:59-77
            torch.random.set_rng_state(state)
        cfg.PROBABILISTIC_INFERENCE.ENSEMBLES.BOX_MERGE_MODE = "pre_nms"
        self.predictor = RetinaNetProbabilisticPredictor(cfg, model=self.model if self.model is not None else model_test_attributes(cfg),
                                                         model_list=[object()] * self.M)
        self.predictor.return_device = True
        self.frame_hw = tuple(frame_hw)
        self.net_hw = tuple(net_hw) if net_hw is not None else \
            anchors.resize_shortest_edge(frame_hw[0], frame_hw[1], cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
        padded = anchors.padded_size(*self.net_hw)
        shapes = anchors.level_shapes(*padded)
        pm = cfg.MODEL.PROBABILISTIC_MODELING
        A, K = len(anchors.ANCHOR_SIZES[0]) * len(anchors.ASPECT_RATIOS), cfg.MODEL.RETINANET.NUM_CLASSES
        D = 0 if pm.BBOX_COV_LOSS.NAME == "none" else (4 if pm.BBOX_COV_LOSS.COVARIANCE_TYPE == "diagonal" else 10)
        self.layout = ensemble_dist.MemberLayout(shapes, A, K, D, pm.CLS_VAR_LOSS.NAME != "none")
        self.like = HeadOutputs(None, None, None, None, anchors.grid_anchors(shapes, device=self.dev), shapes, A, K, self.net_hw)
        self.pipe = ensemble_dist.MemberPipeline(self.layout, self.M, rank, world, self.dev)
        self._resize = lambda frame: modeling.resize_test_image(frame, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)

    def run(self, num_images: int, frame_of, on_detections=None):
        """`frame_of(i)` -> uint8 (3, H, W) frame of image i (every rank sees the same frames).  Returns this rank's
        ([image ids], [records], [counts]) -- device tensors, no host sync."""
        ids, recs, cnts = [], [], []
        h, w = self.frame_hw
        shape_only = torch.empty((3,) + tuple(self.net_hw), device="meta")     # the path only reads the network-input size (IU:39-41)

        def forward(i):
            return self.model(self._resize(frame_of(i)))

        def merge(i, stacked):
            input_im = [{"image": shape_only, "height": h, "width": w, "image_id": i}]
            det = self.predictor._run("standard_nms", input_im, self.layout.views(stacked, self.like))     # PI:502-505
            ids.append(i); recs.append(det.records); cnts.append(det.n_det)
            if on_detections is not None:
                on_detections(i, det)

        self.pipe.run(num_images, forward, merge)
        return ids, recs, cnts


def _run_ensemble_per_gpu(cfg, args, rank, world):
    from . import synthetic
    runner = EnsemblePerGpu(cfg, rank, world)
    _, recs, cnts = runner.run(args.num_images, lambda i: synthetic.synthetic_frame(i, device=runner.dev))
    return recs, cnts


def main(argv=None):
    from . import synthetic
    from .config import setup_config
    from .probabilistic_inference import build_predictor
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--config-file", default=os.path.join(here, "configs/BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml"))
    ap.add_argument("--inference-config", default=os.path.join(here, "configs/Inference/bayes_od_mc_dropout.yaml"))
    ap.add_argument("--num-images", type=int, default=8, help="synthetic 1280x720 frames (ignored with --coco-json)")
    ap.add_argument("--coco-json", default="", help="COCO-format json whose `images` are run (the reference's test data loader, AN:83-84)")
    ap.add_argument("--image-root", default="", help="directory of the files named in --coco-json")
    ap.add_argument("--train-dataset", default="bdd_train", help="cfg.DATASETS.TRAIN[0] of the model (AN:53): with --test-dataset it fixes the category map (AN:52-80)")
    ap.add_argument("--test-dataset", default="bdd_val", help="the data set the detections are written for (AN:55, args.test_dataset)")
    ap.add_argument("--dense-bbox", action="store_true",
                    help="evaluate bbox_subnet / bbox_pred / bbox_cov on every cell, in the reference's order (PR:518-537).  Default since round 6: the "
                         "SPARSE order -- cls side first, candidates selected (PI:283-308), bbox side only over the blocks that can reach a candidate "
                         "(PI:310-331 reads nothing else; pod_compare_amd/sparse.py): the same detections, a function of the image alone "
                         "(tests/test_sparse_tower_gpu.py), 1.3 x the images/s with MC dropout.  --ensemble-per-gpu (one member per GPU) is always dense")
    ap.add_argument("--sparse-bbox", action="store_true", help="(the default; kept for round-5 command lines)")
    ap.add_argument("--random-seed", type=int, default=0)
    ap.add_argument("--output", default="coco_instances_results.json")
    ap.add_argument("--no-graphs", action="store_true", help="issue every launch of a forward from Python instead of replaying a HIP graph per (stream, frame shape)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams per GPU; consecutive images of a rank go to different streams (batch 1 per stream, AN:35); 0 = 2 with "
                         "MC dropout (several runs per image), 3 for an in-process ensemble, 4 otherwise (profiles/r05_streams_sweep.txt)")
    ap.add_argument("--loader-workers", type=int, default=-1,
                    help="host threads that read / decode / resize the files of --coco-json ahead of the GPU (the reference's DATALOADER.NUM_WORKERS); "
                         "-1 = the config's value, 0 = in the main loop")
    ap.add_argument("--flush-every", type=int, default=64,
                    help="images per rank between two gathers of the device-resident records (SURVEY 8e: ~0.74 MB per rank and flush)")
    ap.add_argument("--ensemble-per-gpu", action="store_true",
                    help="BASELINE config 5: rank s < M runs ensemble member s, dense pre-NMS tensors meet on a rotating merge rank")
    ap.add_argument("--binary-output", default="", help="also write the detections as a binary sidecar (inference_utils.write_binary_results)")
    ap.add_argument("--data-dir", default="", help="the reference's core.data_dir(): OUTPUT_DIR = <data-dir>/<dataset>/<family>/<config>/"
                                                   "random_seed_<seed>, whose last_checkpoint is loaded (CS:170-182, PI:59-84)")
    ap.add_argument("--weights", default=None, help="overrides MODEL.WEIGHTS (a local .pth / .pkl with detectron2 names)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl = RCCL over xGMI (the real path); gloo for a "
                                                      "functional run, e.g. several ranks on one GPU with --share-gpu")
    ap.add_argument("--share-gpu", action="store_true", help="functional check only: every rank uses cuda:0")
    ap.add_argument("--random-init", action="store_true",
                    help="synthetic runs: clear MODEL.WEIGHTS / OUTPUT_DIR and keep the seeded random initialisation")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    if world > 1:                 # enqueue + loader threads of this rank on its GPU's NUMA node (hostbind.py; POD_BIND_NUMA=0: off)
        from . import hostbind
        b = hostbind.bind_rank_to_gpu_numa(local_rank)
        print("rank %d: cuda:%d pci %s numa node %s -> %s" % (rank, local_rank, b["pci"], b["numa_node"], b["cpus"] if b["bound"] else "not bound (%s)" % b.get("why_not", "off")))
    cfg = setup_config(args.config_file, args.inference_config, args.random_seed, data_dir=args.data_dir, is_testing=bool(args.data_dir))
    if args.weights is not None:
        cfg.MODEL.WEIGHTS = args.weights
    if args.random_init:
        cfg.MODEL.WEIGHTS, cfg.OUTPUT_DIR = "", ""
    cfg.MODEL.DEVICE = "cuda:%d" % local_rank
    K = cfg.MODEL.RETINANET.NUM_CLASSES
    from . import modeling
    dataset = CocoImages(args.coco_json, args.image_root, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST) if args.coco_json else None
    if dataset is not None:
        if args.ensemble_per_gpu:
            raise SystemExit("--ensemble-per-gpu runs on the synthetic frames only")
        args.num_images = len(dataset)
    mine = shard_indices(args.num_images, rank, world)
    recs, cnts = [], []
    if args.ensemble_per_gpu:
        recs, cnts = _run_ensemble_per_gpu(cfg, args, rank, world)
    torch.manual_seed(args.random_seed)
    predictor = build_predictor(cfg) if not args.ensemble_per_gpu else None
    if predictor is not None:
        predictor.return_device = True      # no per-image host sync: records and counts stay in HBM until the gather
        # every image is evaluated on ONE GPU here (image sharding): the sparse order is the default; POD_SPARSE_BBOX=0 / --dense-bbox give the dense one
        predictor.sparse_bbox_tower = not bool(getattr(args, "dense_bbox", False)) and os.environ.get("POD_SPARSE_BBOX", "1") != "0"
        for m in [predictor.model] + list(predictor.model_list):
            if isinstance(m, modeling.ProbabilisticRetinaNet):
                m.enable_graphs(not getattr(args, "no_graphs", False))      # one host call per forward instead of ~200 launches
                m.graph_after_seen = 2                                      # (a frame size is captured once it has come back: data sets of many sizes stay eager)
    # images are independent units: keep a few in flight on separate HIP streams so one image's low-occupancy backbone
    # stretches overlap another image's head convs (+12 % images/s on one MI355X); the predictor keeps a workspace per stream
    mc = getattr(predictor, "mc_dropout_enabled", False) and getattr(predictor, "num_mc_dropout_runs", 1) > 1
    n_streams = args.streams if args.streams > 0 else (2 if mc else 3 if len(getattr(predictor, "model_list", [])) > 1 else 4)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_streams - 1)]
    width = record_width(K)
    dev = cfg.MODEL.DEVICE
    F = max(1, args.flush_every)
    n_mine = len(shard_indices(args.num_images, 0, world))          # rank 0 owns the most images: every rank flushes as often
    all_ids: List[int] = []
    all_cnt, all_rec = [], []
    flushed: List[Tuple[float, int]] = []

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
        flushed.append((time.perf_counter(), len(all_ids)))       # (the host copies above waited for the records)

    torch.cuda.synchronize()
    t_loop = time.perf_counter()
    if args.ensemble_per_gpu:
        flush_ids = list(mine)
        for a in range(0, max(n_mine, 1), F):
            flush(flush_ids[a:a + F], recs[a:a + F], cnts[a:a + F])
    else:
        chunk_ids: List[int] = []
        recs, cnts = [], []
        workers = args.loader_workers if args.loader_workers >= 0 else int(cfg.DATALOADER.NUM_WORKERS)
        loaded = iter(Prefetched(dataset, mine, workers=workers)) if dataset is not None else None
        with torch.no_grad():
            for j in range(n_mine):
                if j < len(mine):
                    i = mine[j]
                    if loaded is not None:
                        _, d = next(loaded)                             # resized on the host exactly as detectron2's mapper does, a few entries ahead
                    with torch.cuda.stream(streams[j % len(streams)]):
                        if dataset is not None:
                            input_im = [{"image": d["image"].to(dev, non_blocking=True), "height": d["height"], "width": d["width"],
                                         "image_id": i}]                # position in the list: the dataset's id is restored below
                        else:
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
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t_loop
    ids = all_ids
    order = sorted(range(len(ids)), key=lambda q: ids[q])
    ids = [ids[q] for q in order]
    cnt = torch.cat(all_cnt)[order] if all_cnt else torch.zeros((0,), dtype=torch.int32)
    rec = torch.cat(all_rec)[order] if all_rec else torch.zeros((0, 128, width))
    if dataset is not None:
        ids = [dataset.image_id(i) for i in ids]
    if rank == 0:
        with open(args.output, "w") as fp:
            json.dump(results_json(ids, cnt, rec, K, category_mapping(args.train_dataset, args.test_dataset)), fp, indent=4, separators=(",", ": "))
        if args.binary_output:
            from .inference_utils import write_binary_results
            write_binary_results(args.binary_output, ids, cnt, rec, K)
        print("wrote %s: %d images, %d detections" % (args.output, len(ids), int(cnt.sum())))
        if not args.ensemble_per_gpu and t_loop > 0:           # (first images included: eager forwards, then the graph captures)
            steady = ""
            if len(flushed) >= 2 and flushed[-1][0] > flushed[0][0]:
                steady = "; %.1f images/s after the first flush" % ((flushed[-1][1] - flushed[0][1]) / (flushed[-1][0] - flushed[0][0]))
            print("inference loop: %d images on %d rank(s) in %.2f s = %.1f images/s%s (%d stream(s) per GPU%s)" % (
                len(ids), world, t_loop, len(ids) / t_loop, steady, n_streams,
                ", %d loader thread(s)" % workers if dataset is not None else ", synthetic frames made on the device"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
