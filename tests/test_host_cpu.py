"""CPU tests of the host logic: config loading, result format, image sharding + gather (gloo, world_size 2)."""
import glob
import json
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pod_compare_amd import apply_net, config, inference_utils
from pod_compare_amd.anchors import grid_anchors, level_shapes, padded_size, resize_shortest_edge
from pod_compare_amd.structures import Boxes, Instances

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = os.path.join(os.path.dirname(HERE), "pod_compare_amd", "configs")
REF_CFG = "/root/reference/src/configs"


def test_baseline_geometry():
    """1280x720 -> 750x1333 -> padded 768x1344 -> R = 193374 anchors."""
    hw = resize_shortest_edge(720, 1280)
    assert hw == (750, 1333) and padded_size(*hw) == (768, 1344)
    shapes = level_shapes(768, 1344)
    assert shapes == [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]
    anchors = grid_anchors(shapes)
    assert sum(a.shape[0] for a in anchors) == 193374
    # (h, w, a) order, a = size-major x ratio (0.5, 1, 2); first cell of p3
    assert torch.allclose(anchors[0][0], torch.tensor([-22.6274, -11.3137, 22.6274, 11.3137]), atol=1e-4)
    assert torch.allclose(anchors[0][9], anchors[0][0] + torch.tensor([8.0, 0.0, 8.0, 0.0]))


def test_configs_load_with_reference_keys():
    cfg = config.setup_config(os.path.join(CFG, "BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml"),
                              os.path.join(CFG, "Inference/bayes_od_mc_dropout.yaml"))
    pi = cfg.PROBABILISTIC_INFERENCE
    assert pi.INFERENCE_MODE == "bayes_od" and pi.AFFINITY_THRESHOLD == 0.9
    assert pi.MC_DROPOUT.ENABLE is True and pi.MC_DROPOUT.NUM_RUNS == 10
    assert pi.BAYES_OD.CLS_MERGE_MODE == "max_score" and pi.BAYES_OD.BOX_MERGE_MODE == "bayesian_inference"
    pm = cfg.MODEL.PROBABILISTIC_MODELING
    assert pm.DROPOUT_RATE == 0.2 and pm.CLS_VAR_LOSS.NUM_SAMPLES == 10 and pm.BBOX_COV_LOSS.COVARIANCE_TYPE == "diagonal"
    assert cfg.MODEL.RETINANET.NUM_CLASSES == 7 and cfg.MODEL.META_ARCHITECTURE == "ProbabilisticRetinaNet"
    # defaults of core/setup.py:90-133 when the inference yaml is silent
    d = config.setup_config(os.path.join(CFG, "BDD-Detection/retinanet/retinanet_R_50_FPN_1x.yaml"), os.path.join(CFG, "Inference/standard_nms.yaml"))
    assert d.PROBABILISTIC_INFERENCE.AFFINITY_THRESHOLD == 0.7 and d.PROBABILISTIC_INFERENCE.MC_DROPOUT.NUM_RUNS == 1
    assert d.PROBABILISTIC_INFERENCE.ENSEMBLES.RANDOM_SEED_NUMS == [0, 1000, 2000, 3000, 4000]


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present")
def test_own_yamls_equal_reference_yamls():
    """Every model x inference YAML pair gives the same MODEL / PROBABILISTIC_INFERENCE values as the reference's
    own files (read in place; the `eval` tag of Base-RetinaNet.yaml:8 is replaced by literal sizes, never evaluated)."""
    def flat(d, p=""):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(flat(v, p + k + "."))
            else:
                out[p + k] = [list(x) if isinstance(x, (list, tuple)) else x for x in v] if isinstance(v, (list, tuple)) else v
        return out
    n = 0
    for m in glob.glob(REF_CFG + "/BDD-Detection/retinanet/retinanet*.yaml"):
        for i in glob.glob(REF_CFG + "/Inference/*.yaml"):
            a = config.setup_config(m, i)
            b = config.setup_config(m.replace(REF_CFG, CFG), i.replace(REF_CFG, CFG))
            for sect in ("MODEL", "PROBABILISTIC_INFERENCE"):
                assert flat(a[sect]) == flat(b[sect]), (m, i)
            n += 1
    assert n == 32


def test_unknown_meta_architecture_and_mode_raise():
    from pod_compare_amd import probabilistic_inference as pinf
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    with pytest.raises(ValueError):
        pinf.build_predictor(cfg, model=object())


def test_instances_to_json_format():
    inst = Instances((720, 1280))
    inst.pred_boxes = Boxes(torch.tensor([[10., 20., 110., 220.], [5., 5., 6., 7.]]))
    inst.scores = torch.tensor([0.9, 0.4])
    inst.pred_classes = torch.tensor([2, 9])          # class 9 is not in the map -> dropped (IU:477-479, 491)
    inst.pred_cls_probs = torch.rand(2, 7)
    c = torch.rand(2, 4, 4)
    inst.pred_boxes_covariance = c @ c.transpose(1, 2)
    js = inference_utils.instances_to_json(inst, 42, {i: i + 1 for i in range(7)})
    assert len(js) == 1 and set(js[0]) == {"image_id", "category_id", "bbox", "score", "cls_prob", "bbox_covar"}
    assert js[0]["image_id"] == 42 and js[0]["category_id"] == 3 and js[0]["bbox"] == [10.0, 20.0, 100.0, 200.0]
    t = torch.tensor([[1., 0, 0, 0], [0, 1., 0, 0], [-1., 0, 1., 0], [0, -1., 0, 1.]])
    assert torch.allclose(torch.tensor(js[0]["bbox_covar"]), t @ inst.pred_boxes_covariance[0] @ t.t(), atol=1e-6)
    assert inference_utils.instances_to_json(Instances((1, 1), pred_boxes=Boxes(torch.zeros(0, 4))), 1, {}) == []


def test_shard_indices_cover_all_images_once():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in apply_net.shard_indices(21, r, world))
        assert seen == list(range(21))


def _gather_worker(rank, world, port, num_images, tmp, flush_every=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, md = 7, 4
    width = inference_utils.record_width(K)
    mine = apply_net.shard_indices(num_images, rank, world)
    rec = torch.zeros((len(mine), md, width))
    cnt = torch.zeros((len(mine),), dtype=torch.int32)
    for j, i in enumerate(mine):
        cnt[j] = i % (md + 1)
        for d in range(int(cnt[j])):
            rec[j, d, :4] = torch.tensor([i, d, 10.0, 20.0])
            rec[j, d, 4] = 1.0 / (1 + d)
            rec[j, d, 5] = (i + d) % K
    if flush_every is None:
        ids, c, r = apply_net.gather_records(mine, cnt, rec, num_images, world)
    else:   # the driver's periodic flush: every rank joins every collective, chunks of `flush_every` images per rank
        n_flush = len(apply_net.shard_indices(num_images, 0, world))
        ids, cs, rs = [], [], []
        for a in range(0, n_flush, flush_every):
            i2, c2, r2 = apply_net.gather_records(mine[a:a + flush_every], cnt[a:a + flush_every], rec[a:a + flush_every], num_images,
                                                  world, per_rank=flush_every)
            ids.extend(i2); cs.append(c2); rs.append(r2)
        order = sorted(range(len(ids)), key=lambda q: ids[q])
        ids = [ids[q] for q in order]
        c, r = torch.cat(cs)[order], torch.cat(rs)[order]
    if rank == 0:
        js = apply_net.results_json(ids, c, r, K, apply_net.BDD_CAT_MAP)
        with open(os.path.join(tmp, "out.json"), "w") as f:
            json.dump({"ids": ids, "counts": c.tolist(), "js": js}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_images,flush_every", [(7, None), (8, None), (7, 2), (9, 3)])
def test_two_rank_gather_restores_image_order(tmp_path, num_images, flush_every):
    """world_size-2 gloo run of the sharding + all_gather + re-ordering logic (ragged shards when num_images is odd), in one
    gather or in the driver's periodic flushes."""
    port = 29500 + (os.getpid() % 2000) + num_images + 10 * (flush_every or 0)
    mp.spawn(_gather_worker, args=(2, port, num_images, str(tmp_path), flush_every), nprocs=2, join=True)
    out = json.load(open(tmp_path / "out.json"))
    assert out["ids"] == list(range(num_images))
    assert out["counts"] == [i % 5 for i in range(num_images)]
    assert len(out["js"]) == sum(i % 5 for i in range(num_images))
    first = [d for d in out["js"] if d["image_id"] == 3]
    assert [d["bbox"][1] for d in first] == [0.0, 1.0, 2.0] and first[0]["category_id"] == 4


# ---- config 5 topology: one seed per rank, dense pre-NMS exchange ------------------------------------------------

def test_member_layout_roundtrip():
    from pod_compare_amd import ensemble_dist, synthetic
    ho = synthetic.planted_head_outputs((96, 128), 3, seed=3, num_boxes=4)
    from pod_compare_amd.probabilistic_inference import run_slice
    lay = ensemble_dist.MemberLayout.of(ho)
    assert lay.total % 4 == 0 and all(off % 4 == 0 for off, *_ in lay.offsets.values())
    stacked = torch.stack([lay.pack(run_slice(ho, r)) for r in range(3)])
    v = lay.views(stacked, ho)
    for name in ("cls", "delta", "cls_var", "reg_var"):
        for a, b in zip(getattr(v, name), getattr(ho, name)):
            assert a.shape == b.shape and torch.equal(a, b)
            assert a.stride(0) == lay.total and a[1].is_contiguous()


def _exchange_worker(rank, world, port, n_members, tmp):
    from pod_compare_amd import ensemble_dist, synthetic
    from pod_compare_amd.probabilistic_inference import run_slice
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ho = synthetic.planted_head_outputs((96, 128), n_members, seed=9, num_boxes=4)      # run s plays member s
    lay = ensemble_dist.MemberLayout.of(ho)
    ok = True
    for image in range(4):
        dst = ensemble_dist.merge_rank(image, world)
        packed = lay.pack(run_slice(ho, rank)) + float(image) if rank < n_members else None
        stacked = torch.zeros((n_members, lay.total)) if rank == dst else None
        ensemble_dist.exchange_members(packed, stacked, n_members, dst, rank)
        if rank == dst:
            v = lay.views(stacked, ho)
            for name in ("cls", "delta", "cls_var", "reg_var"):
                for a, b in zip(getattr(v, name), getattr(ho, name)):
                    ok = ok and torch.equal(a, b + float(image))
    with open(os.path.join(tmp, "ok_%d" % rank), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_ensemble_exchange_three_ranks_two_members(tmp_path):
    """gloo, world 3, 2 member ranks, merge rank rotating over all 3 (incl. a non-member rank)."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(3, port, 2, str(tmp_path)), nprocs=3, join=True)
    assert [open(tmp_path / ("ok_%d" % r)).read() for r in range(3)] == ["1", "1", "1"]


def _pipeline_worker(rank, world, port, n_members, num_images, tmp):
    from pod_compare_amd import ensemble_dist, synthetic
    from pod_compare_amd.probabilistic_inference import run_slice
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ho = synthetic.planted_head_outputs((96, 128), n_members, seed=9, num_boxes=4)      # run s plays member s
    lay = ensemble_dist.MemberLayout.of(ho)
    pipe = ensemble_dist.MemberPipeline(lay, n_members, rank, world, "cpu")
    log, ok = [], True

    def forward(i):
        log.append(("forward", i))
        m = run_slice(ho, rank)
        add = lambda lst: [t + float(i) for t in lst]
        return synthetic.HeadOutputs(add(m.cls), add(m.delta), add(m.cls_var), add(m.reg_var), m.anchors, m.shapes, m.num_anchors,
                                     m.num_classes, m.image_size)

    def merge(i, stacked):
        nonlocal ok
        log.append(("merge", i))
        v = lay.views(stacked, ho)                      # the strided (members, A*C, H, W) views K1 streams
        for name in ("cls", "delta", "cls_var", "reg_var"):
            for a, b in zip(getattr(v, name), getattr(ho, name)):
                ok = ok and a.stride(0) == lay.total and torch.equal(a, b + float(i))

    pipe.run(num_images, forward, merge)
    with open(os.path.join(tmp, "log_%d.json" % rank), "w") as f:
        json.dump({"ok": ok, "log": log}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_members,num_images", [(3, 2, 7), (2, 2, 5)])
def test_member_pipeline_keeps_two_images_in_flight(tmp_path, world, n_members, num_images):
    """The pipelined config-5 exchange (PI:495-505 replaced): every merge sees exactly its image's member rows through the
    strided views, every image is merged once on its rotating merge rank, and a member rank enqueues the forward of image
    i+1 BEFORE it merges image i (so image i's rows travel underneath that forward)."""
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_pipeline_worker, args=(world, port, n_members, num_images, str(tmp_path)), nprocs=world, join=True)
    merged = []
    for r in range(world):
        out = json.load(open(tmp_path / ("log_%d.json" % r)))
        assert out["ok"]
        log = [tuple(e) for e in out["log"]]
        mine = [i for kind, i in log if kind == "merge"]
        assert mine == [i for i in range(num_images) if i % world == r]
        merged += mine
        if r < n_members:
            assert [i for kind, i in log if kind == "forward"] == list(range(num_images))
            for i in mine:
                if i + 1 < num_images:
                    assert log.index(("forward", i + 1)) < log.index(("merge", i))
        else:
            assert all(kind == "merge" for kind, _ in log)
    assert sorted(merged) == list(range(num_images))


def test_binary_sidecar_round_trip_equals_the_json_records(tmp_path):
    """SURVEY f-2: the binary sidecar of coco_instances_results.json carries the same detections as the JSON file."""
    K, md = 7, 128
    g = torch.Generator().manual_seed(8)
    ids = [5, 9, 12]
    counts = torch.tensor([3, 0, 100], dtype=torch.int32)
    rec = torch.randn(3, md, inference_utils.record_width(K), generator=g)
    rec[:, :, 5] = torch.randint(0, K, (3, md), generator=g).float()
    path = str(tmp_path / "results.podr")
    inference_utils.write_binary_results(path, ids, counts, rec, K)
    ids2, counts2, rec2, k2 = inference_utils.read_binary_results(path)
    assert ids2 == ids and k2 == K and torch.equal(counts2, counts)
    for i, c in enumerate(counts.tolist()):          # the rows K7 wrote travel unchanged; rows behind the count (never written: torch.empty) are zeroed,
        assert torch.equal(rec2[i, :c], rec[i, :c]) and float(rec2[i, c:].abs().sum()) == 0.0      # so the file is a function of the detections alone
    want = apply_net.results_json(ids, counts, rec, K, apply_net.BDD_CAT_MAP)
    got = inference_utils.binary_results_to_json(path, apply_net.BDD_CAT_MAP)
    assert got == want and len(got) == 103
    assert os.path.getsize(path) < 0.35 * len(json.dumps(want, indent=4))          # ~4x smaller than the indented JSON
    with pytest.raises(ValueError):
        open(path, "r+b").write(b"XXXX")
        inference_utils.read_binary_results(path)


def test_coco_image_list_is_mapped_like_detectron2s_test_loader(tmp_path):
    """AN:83-84 restated (apply_net.CocoImages): RGB file -> BGR uint8 (3, H', W'), ResizeShortestEdge with PIL's bilinear filter on
    the uint8 array (what detectron2's ResizeTransform does), original height / width and the dataset's image id passed on."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(72, 128, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.png")
    Image.fromarray(rgb[:40, :50]).save(tmp_path / "b.png")
    spec = {"images": [{"id": 901, "file_name": "a.png", "height": 72, "width": 128}, {"id": 17, "file_name": "b.png", "height": 40, "width": 50}]}
    (tmp_path / "set.json").write_text(json.dumps(spec))
    ds = apply_net.CocoImages(str(tmp_path / "set.json"), str(tmp_path), min_size=80, max_size=120)
    assert len(ds) == 2 and ds.image_id(1) == 17
    a = ds[0]
    nh, nw = resize_shortest_edge(72, 128, 80, 120)                  # the long side caps the scale: 120 / 128
    assert (nh, nw) == (68, 120) and tuple(a["image"].shape) == (3, nh, nw) and a["image"].dtype == torch.uint8
    want = np.asarray(Image.fromarray(rgb).resize((nw, nh), Image.BILINEAR))[:, :, ::-1].transpose(2, 0, 1)
    assert np.array_equal(a["image"].numpy(), want)
    assert (a["height"], a["width"], a["image_id"]) == (72, 128, 901)
    b = ds[1]
    assert tuple(b["image"].shape) == (3, 80, 100) and (b["height"], b["width"], b["image_id"]) == (40, 50, 17)


def test_prefetched_loader_keeps_order_contents_and_errors(tmp_path):
    """apply_net.Prefetched (the reference's DATALOADER.NUM_WORKERS side of the test loader): entries come back in the order asked for,
    equal to the plain loop's, for any number of threads and look-ahead; a worker's exception surfaces at its entry."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(5)
    images = []
    for k in range(9):
        h, w = 30 + 3 * k, 50 + 2 * k
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(tmp_path / ("f%d.png" % k))
        images.append({"id": 100 - k, "file_name": "f%d.png" % k, "height": h, "width": w})
    (tmp_path / "set.json").write_text(json.dumps({"images": images}))
    ds = apply_net.CocoImages(str(tmp_path / "set.json"), str(tmp_path), min_size=40, max_size=80)
    order = [7, 0, 3, 3, 8, 1]
    plain = [(i, ds[i]) for i in order]
    for workers, depth in ((0, 0), (1, 1), (3, 2), (4, 0), (8, 16)):
        got = list(apply_net.Prefetched(ds, order, workers=workers, depth=depth, pin=False))
        assert [i for i, _ in got] == order
        for (_, a), (_, b) in zip(got, plain):
            assert torch.equal(a["image"], b["image"]) and (a["height"], a["width"], a["image_id"]) == (b["height"], b["width"], b["image_id"])
    assert list(apply_net.Prefetched(ds, [], workers=2)) == []
    images.append({"id": 1, "file_name": "missing.png", "height": 1, "width": 1})
    (tmp_path / "set.json").write_text(json.dumps({"images": images}))
    ds = apply_net.CocoImages(str(tmp_path / "set.json"), str(tmp_path), min_size=40, max_size=80)
    it = iter(apply_net.Prefetched(ds, [0, 9, 1], workers=2, pin=False))
    assert next(it)[0] == 0
    with pytest.raises(FileNotFoundError):
        next(it)


def test_wino_block_table_canvases():
    """Host side of pod_wino_conv3x3: level-major pixel offsets and the block records {first in pixel, first out pixel,
    grid_cols<<24 | H<<12 | W, n_images<<24 | by<<12 | bx} (include/pod_mi355x.h): the images of a level stand in a grid on one canvas,
    one zero row / column apart.  Decoding the records the way the kernel does must give every output pixel of every image exactly once."""
    from pod_compare_amd.wino import block_table, canvas_layout, level_pixel_offsets
    levels = [(23, 40), (6, 10), (16, 32)]
    assert level_pixel_offsets(levels, 3) == [0, 3 * 920, 3 * 920 + 3 * 60, 3 * 920 + 3 * 60 + 3 * 512]

    def decode(table):
        """-> list of (in pixel, out pixel) of every image pixel the blocks cover"""
        pairs = []
        for pin, pout, geo, blk in table.tolist():
            gcols, H, W = (geo >> 24) & 0xFF, (geo >> 12) & 0xFFF, geo & 0xFFF
            n, by, bx = (blk >> 24) & 0x7F, (blk >> 12) & 0xFFF, blk & 0xFFF
            assert gcols >= 1 and n >= 1
            hit = 0
            for cy in range(16 * by, 16 * by + 16):
                gr, y = divmod(cy, H + 1)
                for cx in range(16 * bx, 16 * bx + 16):
                    gc, x = divmod(cx, W + 1)
                    img = gr * gcols + gc
                    if y < H and x < W and gc < gcols and img < n:
                        pairs.append((pin + img * H * W + y * W + x, pout + img * H * W + y * W + x))
                        hit += 1
            assert hit > 0, "a block that covers no image pixel is wasted work"
        return pairs

    t = block_table(levels, 2, "cpu", in_copies=5, in_first=1, out_copies=3)
    assert t.dtype == torch.int32 and t.shape[1] == 4
    pairs = decode(t)
    ioffs, ooffs = level_pixel_offsets(levels, 5), level_pixel_offsets(levels, 3)
    want = [(ioffs[l] + (1 + c) * h * w + p, ooffs[l] + c * h * w + p) for l, (h, w) in enumerate(levels) for c in range(2) for p in range(h * w)]
    assert sorted(pairs) == sorted(want)                     # every pixel once, read from images in_first.. and written to images 0..
    assert len({tuple(r) for r in t.tolist()}) == t.shape[0]
    assert block_table(levels, 2, "cpu", in_copies=5, in_first=1, out_copies=3) is t      # cached
    assert t.pod_pixels == 2 * (920 + 60 + 512) and t.pod_levels == 3
    # 19 maps of 6 x 11 share at most 8 blocks (one canvas per image: 19)
    rows, cols, blocks = canvas_layout(6, 11, 19)
    assert rows * cols >= 19 and len(blocks) <= 8
    # the benchmark launch: 19 runs x 5 levels in 1624 blocks (one canvas per image: 1767), each pixel once
    bench_levels = [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]
    big = block_table(bench_levels, 19, "cpu")
    assert big.shape[0] == 1624
    got = decode(big)
    assert len(got) == big.pod_pixels and len({p for _, p in got}) == big.pod_pixels and all(a == b for a, b in got)


def test_category_mapping_follows_the_reference_tables():
    """AN:52-80 + core/datasets/metadata.py: BDD -> BDD / Lyft is the identity + 1, BDD -> KITTI keeps car and person only, an
    incompatible pair raises (the reference builds the ValueError and forgets to raise it)."""
    import pytest
    from pod_compare_amd.apply_net import category_mapping
    from pod_compare_amd.inference_utils import records_to_json
    assert category_mapping("bdd_train", "bdd_val") == {i: i + 1 for i in range(7)}
    assert category_mapping("bdd_train", "lyft_val") == {i: i + 1 for i in range(7)}
    assert category_mapping("bdd_train", "kitti_val") == {0: 1, 3: 2}           # car -> 1, person -> 2; bus / truck / rider / bike / motor: dropped
    assert category_mapping("kitti_train", "kitti_val") == {0: 1, 1: 2}
    with pytest.raises(ValueError):
        category_mapping("kitti_train", "bdd_val")
    with pytest.raises(ValueError):
        category_mapping("coco_2017_train", "voc_2012_val")
    # a KITTI run drops the classes KITTI does not annotate (IU:466-471: category -1 is skipped)
    import torch
    rec = torch.zeros(3, 4 + 1 + 1 + 7 + 16)
    rec[:, 2:4] = 10.0
    rec[:, 4] = torch.tensor([0.9, 0.8, 0.7])
    rec[:, 5] = torch.tensor([0.0, 1.0, 3.0])                                   # car, bus, person
    out = records_to_json(rec, 3, 5, 7, category_mapping("bdd_train", "kitti_val"))
    assert [d["category_id"] for d in out] == [1, 2]


def test_host_binding_reads_the_numa_node_from_sysfs(tmp_path):
    """hostbind: the GPU's NUMA node and that node's CPU list come from sysfs; a box without the entries is left alone (and says so)."""
    from pod_compare_amd import hostbind
    sysfs = tmp_path / "sys"
    (sysfs / "bus/pci/devices/0000:c1:00.0").mkdir(parents=True)
    (sysfs / "bus/pci/devices/0000:c1:00.0/numa_node").write_text("1\n")
    (sysfs / "devices/system/node/node1").mkdir(parents=True)
    (sysfs / "devices/system/node/node1/cpulist").write_text("2-3,66-67\n")
    assert hostbind.numa_node_of("0000:c1:00.0", str(sysfs)) == 1
    assert hostbind.cpus_of_node(1, str(sysfs)) == [2, 3, 66, 67]
    assert hostbind.numa_node_of("0000:c2:00.0", str(sysfs)) is None and hostbind.numa_node_of(None, str(sysfs)) is None
    (sysfs / "bus/pci/devices/0000:c3:00.0").mkdir(parents=True)
    (sysfs / "bus/pci/devices/0000:c3:00.0/numa_node").write_text("-1\n")
    assert hostbind.numa_node_of("0000:c3:00.0", str(sysfs)) is None          # single-node host: the kernel reports -1
    assert hostbind._parse_cpulist("0-2,8,10-11") == [0, 1, 2, 8, 10, 11]
