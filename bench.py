#!/usr/bin/env python3
"""images/sec of the probabilistic-inference path (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W] [--config cfg3] [--no-cnn] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches this script under torch.distributed.run with N
ranks (one per GPU); inside a launch, WORLD_SIZE must equal --gpus (a mismatch is an error, never a silent 1-rank run).

A "step" is one image through predictor(input_im): ResNet-50-FPN + probabilistic RetinaNet head on PyTorch-ROCm
(MC-dropout runs batched), then the hand-written HIP hot path K1..K7 in native-RNG mode (in-kernel Philox).  Frames
(uint8 1280x720) and the planted head tensors are resident in HBM before the timed region.  Random-init weights give
p ~= 0.01 < 0.05, i.e. no detections, so -- as SURVEY 8(d) prescribes -- the conv net is run and timed on
the frame, its output is discarded, and the hot path consumes seeded planted-object head tensors of the same shape
(rotated over `--images` distinct sets, 170 MB each at N = 10, so consecutive steps never re-read a cache-resident
buffer).

Images shard over ranks (rank r takes images r, r+world, ...: weak scaling, per-GPU work fixed); the only collective is
one RCCL all_gather of the fixed-stride detection records at the end of the timed region.  `--config cfg5
--ensemble-per-gpu` is BASELINE configs[4] in its 8-GPU topology instead: one ensemble member per rank, the dense
pre-NMS tensors exchanged point-to-point onto a rotating merge rank (pod_compare_amd.ensemble_dist.MemberPipeline).
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pod_compare_amd import anchors as A  # noqa: E402
from pod_compare_amd import hotpath, modeling, synthetic  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1..4]; configs[0] (CPU plumbing) is a parity-test case
    "cfg2": dict(name="retinanet_R_50_FPN_1x_reg_cls_var + bayes_od.yaml", mode="bayes_od", runs=1, cls_var=True, reg_var=True,
                 dropout=0.0, yaml=("retinanet_R_50_FPN_1x_reg_cls_var.yaml", "bayes_od.yaml")),
    "cfg3": dict(name="retinanet_R_50_FPN_1x_reg_cls_var_dropout + bayes_od_mc_dropout.yaml (N=10 MC samples)",
                 mode="bayes_od", runs=10, cls_var=True, reg_var=True, dropout=0.2,
                 yaml=("retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml", "bayes_od_mc_dropout.yaml")),
    "cfg4": dict(name="retinanet_R_50_FPN_1x + anchor_statistics.yaml", mode="anchor_statistics", runs=1, cls_var=False,
                 reg_var=False, dropout=0.0, yaml=("retinanet_R_50_FPN_1x.yaml", "anchor_statistics.yaml")),
    # BASELINE configs[4]: all 5 members on ONE GPU by default; --ensemble-per-gpu = one member per rank (>= 5 ranks)
    "cfg5": dict(name="5-seed ensembles_pre_nms.yaml (reg_cls_var)", mode="ensembles", runs=5,
                 cls_var=True, reg_var=True, dropout=0.0, members=5, yaml=("retinanet_R_50_FPN_1x_reg_cls_var.yaml", "ensembles_pre_nms.yaml")),
}
FRAME_HW = (720, 1280)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix peak (no xf32 / tf32 on gfx950)
BF16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 / f16 matrix peak (16 x the fp32 rate)
SPLIT_PRODUCTS = 3           # 16-bit partial products per fp32 product of the split kernels (round 5: 2-way f16 splits; rounds 3-4: 6, 3-way bf16)


def k1_algorithmic_bytes(R, K, D, N, has_cls_var, quirk, dense_box=True):
    """4*R*C*(N+1), C = 2K+4+D channels per anchor, N runs read + merged tensors written.
    The reference's merge (PI:216-222) never reads the last run, so with the quirk on only N-1 runs are
    streamed: the smaller figure is used so the fraction is never flattered.  N = 1: score pass only.
    dense_box=False: the product path, where box_delta / box_reg_var are merged at the candidates by K2b and K1
    streams the C = 2K class channels only."""
    C = K * (2 if has_cls_var else 1) + ((4 + D) if dense_box else 0)
    if N == 1:
        return 4 * R * K * (2 if has_cls_var else 1)
    reads = (N - 1) if quirk else N
    return 4 * R * C * (reads + 1)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--images", type=int, default=4, help="distinct planted head-tensor sets kept in HBM")
    ap.add_argument("--synth", default="planted", choices=["planted", "worst"])
    ap.add_argument("--boxes", type=int, default=24, help="planted objects per image (SURVEY 8d: 24); more objects = more candidates")
    ap.add_argument("--no-cnn", action="store_true", help="time the HIP hot path only (diagnostic; not the headline)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams per GPU, images round-robin (batch 1 per stream as in AN:35; SURVEY 8d); 0 = the config's default: 2 for the "
                         "MC-dropout config (its step is K12's head launches; a third image in flight only adds contention: 101.6 - 102.6 against "
                         "99.6 - 99.7 images/s, profiles/r04_experiments.md), 4 for the single-model configs (profiles/r05_streams_sweep.txt: "
                         "570 / 628 / 646 / 616 images/s with 2 / 3 / 4 / 6 streams on cfg2), 3 for the in-process ensemble")
    ap.add_argument("--ensemble-per-gpu", action="store_true",
                    help="cfg5 only: one ensemble member per rank (needs --gpus >= 5), exchange pipelined over RCCL p2p")
    ap.add_argument("--fp32-mfma", action="store_true",
                    help="3x3 convolutions on pod_wino_conv3x3 (fp32 matrix instructions) instead of pod_wino_conv3x3_split (every fp32 product "
                         "formed from 2-way f16 splits of the scaled operands on the f16 matrix cores, fp32 accumulate: the production kernel "
                         "since round 4, contract in tests/test_wino_conv_gpu.py); the other kernel is always measured as a second leg "
                         "(`fp32_mfma` / `split_f16`)")
    ap.add_argument("--split-bf16", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-graphs", action="store_true", help="issue every launch of the model forward from Python instead of replaying a HIP graph "
                                                             "per (stream, shape)")
    ap.add_argument("--sparse-bbox", action="store_true",
                    help="time the SPARSE bbox tower as the step (never the headline `value` of a default run, which stays the dense head and carries the "
                         "sparse figures beside it): cls side first, candidates selected, bbox side over the blocks that can reach one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the K1 / conv / NLL / worst-case legs after the timed region")
    ap.add_argument("--cpu-images", type=int, default=256, help="upper bound; the CPU leg stops after ~12 s of CPU work")
    ap.add_argument("--k1-traffic-bytes", type=float, default=None,
                    help="HBM bytes per K1 launch from the rocprofv3 PMC passes (profiles/): 2*FETCH_SIZE + WRITE_SIZE")
    a = ap.parse_args()
    a.split_bf16 = not a.fp32_mfma
    return a


def relaunch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (torch.distributed.run, one per GPU)."""
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("POD_BENCH_SHARE_GPU") != "1":
        raise SystemExit("bench.py --gpus %d: this node exposes %d GPU(s) (POD_BENCH_SHARE_GPU=1 + POD_BENCH_BACKEND=gloo runs "
                         "all ranks on cuda:0 as a functional check only)" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def conv_census(model, img, N, quirk, dev):
    """FLOPs of one image's conv net from the layer shapes it actually runs, and the time of every conv call on ONE
    stream (HIP events around each F.conv2d): the part of a step that is MIOpen's, priced against the fp32 MFMA peak.
    Categories: the NHWC trunk of large maps (MIOpen implicit GEMM), the other 3x3 convs (Winograd), 1x1 (GEMM), stem."""
    import torch.nn.functional as F
    real = F.conv2d
    calls = []

    def probe(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real(x, w, b, stride, padding, dilation, groups)
        e1.record()
        kh, kw = int(w.shape[2]), int(w.shape[3])
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3] * (w.shape[1]) * kh * kw
        nhwc = x.dim() == 4 and x.shape[1] > 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        cat = "stem_7x7" if kh == 7 else ("1x1" if kh == 1 else ("3x3_nhwc_trunk" if nhwc else "3x3_nchw"))
        calls.append((cat, flops, e0, e1))
        return y

    from pod_compare_amd import wino
    real_wino = wino.WinoConv.__call__

    real_planes = wino.WinoConv.planes_of_one_image
    inside = [False]

    def probe_planes(self, src, dst, table, **kw):      # a backbone / FPN convolution: one launch, or (small maps) split launch + reduce
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        inside[0] = True
        try:
            y = real_planes(self, src, dst, table, **kw)
        finally:
            inside[0] = False
        e1.record()
        calls.append(("3x3_backbone_fpn_winograd_hip", 2.0 * table.pod_pixels * self.C * self.K * 9, e0, e1))
        return y

    def probe_wino(self, src, dst, table, **kw):
        if inside[0]:
            return real_wino(self, src, dst, table, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real_wino(self, src, dst, table, **kw)
        e1.record()
        # a head launch covers all FPN levels x runs; a backbone / FPN launch is one NCHW feature map (modeling.wino_conv_nchw)
        cat = "3x3_head_winograd_hip" if table.pod_levels > 1 else "3x3_backbone_fpn_winograd_hip"
        calls.append((cat, 2.0 * table.pod_pixels * self.C * self.K * 9, e0, e1))   # direct-convolution FLOPs
        return y

    real_rep = wino.WinoConv.replicas

    def probe_rep(self, src, dst, table, replicas, **kw):     # the first conv of an MC-dropout subnet: evaluated once, stored `replicas` times
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real_rep(self, src, dst, table, replicas, **kw)
        e1.record()
        calls.append(("3x3_head_winograd_hip", 2.0 * src.shape[0] * self.C * self.K * 9, e0, e1))
        return y

    real_grouped = wino.grouped_launch

    def probe_grouped(sets, **kw):                       # layer l of both subnets / the four predictors in one launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_grouped(sets, **kw)
        e1.record()
        flops = sum(2.0 * (st["src"].shape[0] if st.get("replicas", 0) else st["table"].pod_pixels) * st["conv"].C * st["conv"].K * 9 for st in sets)
        calls.append(("3x3_head_winograd_hip", flops, e0, e1))

    real_cl = wino.WinoConv.channels_last_of_one_image

    def probe_cl(self, src, table, **kw):               # the same convolutions in the channels-last backbone
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        inside[0] = True
        try:
            y = real_cl(self, src, table, **kw)
        finally:
            inside[0] = False
        e1.record()
        calls.append(("3x3_backbone_fpn_winograd_hip", 2.0 * table.pod_pixels * self.C * self.K * 9, e0, e1))
        return y

    from pod_compare_amd import conv1x1
    real_c1 = conv1x1.Conv1x1.__call__

    def probe_c1(self, x, h, w, **kw):                   # pod_conv1x1_split (+ its reduce launch on small maps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real_c1(self, x, h, w, **kw)
        e1.record()
        calls.append(("1x1_hip", 2.0 * y.shape[0] * self.C * self.K, e0, e1))
        return y

    real_stem = conv1x1.Stem7x7.__call__

    def probe_stem(self, x, **kw):                       # pod_stem7x7_split (frame normalisation + padding + conv + ReLU)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real_stem(self, x, **kw)
        e1.record()
        calls.append(("stem_7x7_hip", 2.0 * y[1] * y[2] * 64 * 147, e0, e1))
        return y

    reps = 3
    conv1x1.Stem7x7.__call__ = probe_stem
    wino.WinoConv.replicas = probe_rep
    wino.grouped_launch = probe_grouped
    F.conv2d = probe
    wino.WinoConv.__call__ = probe_wino
    wino.WinoConv.planes_of_one_image = probe_planes
    wino.WinoConv.channels_last_of_one_image = probe_cl
    conv1x1.Conv1x1.__call__ = probe_c1
    graphs = getattr(model, "use_graphs", False)
    model.use_graphs = False                              # a replayed graph calls none of the probes: the census is of the eager forward
    try:
        with torch.no_grad():
            for _ in range(reps):
                model(img, num_mc_dropout_runs=N, skip_unused_last_run=quirk)
        torch.cuda.synchronize()
    finally:
        model.use_graphs = graphs
        F.conv2d = real
        wino.WinoConv.__call__ = real_wino
        wino.WinoConv.planes_of_one_image = real_planes
        wino.WinoConv.channels_last_of_one_image = real_cl
        conv1x1.Conv1x1.__call__ = real_c1
        conv1x1.Stem7x7.__call__ = real_stem
        wino.WinoConv.replicas = real_rep
        wino.grouped_launch = real_grouped
    out = {}
    for cat, flops, e0, e1 in calls:
        d = out.setdefault(cat, {"calls": 0, "gflop": 0.0, "ms": 0.0})
        d["calls"] += 1
        d["gflop"] += flops / 1e9
        d["ms"] += e0.elapsed_time(e1)
    for d in out.values():
        d["calls"] //= reps
        d["gflop"] /= reps
        d["ms"] /= reps
        d["tflops"] = d["gflop"] / d["ms"] if d["ms"] > 0 else None
    for w in (out.get("3x3_head_winograd_hip"), out.get("3x3_backbone_fpn_winograd_hip")):
      if w:   # F(2,3) x F(4,3): 24 multiplies where the direct form has 72, tiles of partial 16x16 blocks included in the time only
        w["note"] = ("pod_wino_conv3x3; gflop / tflops are DIRECT-convolution FLOPs (the model's arithmetic), the matrix cores "
                     "execute 24/72 of them: mfma_tflops_executed is what to hold against the 157.3 TFLOP/s peak")
        w["mfma_tflops_executed"] = w["tflops"] * 24.0 / 72.0 if w["tflops"] else None
    return out


def head_conv_roofline(model, net_hw, N, quirk, dev):
    """pod_wino_conv3x3 on its largest launch of a step (one bbox_subnet layer: every MC run of every FPN level), timed with
    HIP events on the launch stream.  fp32 Winograd, F(2,3) down the rows x F(4,3) along the columns: per 2x4 output tile and (c, k)
    pair the matrix cores execute 24 multiply-adds where the direct convolution has 72, so `achieved` (executed MFMA FLOPs of the real tiles / time) is what
    to hold against the fp32 MFMA peak, and `direct_equivalent_tflops` is the rate in the model's own arithmetic."""
    from pod_compare_amd import wino
    head = model.head
    from pod_compare_amd import anchors as A
    levels = [tuple(int(v) for v in hw) for hw in A.level_shapes((net_hw[0] + 31) // 32 * 32, (net_hw[1] + 31) // 32 * 32)]
    skip = 1 if (quirk and N > 1 and head.dropout_rate > 0.0) else 0
    copies = N + ((N - skip) if head.compute_bbox_cov else 0)
    conv = head._wino(head.bbox_subnet[1])
    table = wino.block_table(levels, copies, dev)
    src = torch.randn(table.pod_pixels, conv.C, device=dev)
    dst = torch.empty(table.pod_pixels, conv.Kpad, device=dev)
    for _ in range(8):          # (the chip needs a few launches of this kernel to settle its clock)
        conv(src, dst, table, relu=True, dropout_p=head.dropout_rate, seed=1, offset=0)
    torch.cuda.synchronize()
    B, nb = 8, 6
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nb)]
    torch.cuda._sleep(3_000_000)
    for a, b in evs:
        a.record()
        for _ in range(B):
            conv(src, dst, table, relu=True, dropout_p=head.dropout_rate, seed=1, offset=0)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) / B for a, b in evs)
    avg = sum(ms) / len(ms)
    tiles = copies * sum(((h + 1) // 2) * ((w + 3) // 4) for h, w in levels)        # 2 x 4 output tiles, 24 Winograd positions each
    mfma_flop = 2.0 * 24 * tiles * conv.C * conv.K * (SPLIT_PRODUCTS if conv.split else 1)      # split kernel: 3 f16 partial products per fp32 product
    peak = BF16_MFMA_PEAK_TF if conv.split else FP32_MFMA_PEAK_TF
    direct_flop = 2.0 * 9 * table.pod_pixels * conv.C * conv.K
    traffic = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_wino_split_traffic.json" if conv.split else "*_wino_traffic.json")), reverse=True):
        t = json.load(open(path))
        if t.get("levels") == [list(x) for x in levels] and t.get("copies") == copies:
            traffic = t.get("traffic_bytes")
            break
    return {"kernel": "%s: conv3x3 256->256 + bias + ReLU + dropout, %d runs x %d levels in one launch" % (
                "pod_wino_conv3x3_split (k_wino_conv3x3_split, f16 matrix cores: 3 partial products per fp32 product)" if conv.split
                else "pod_wino_conv3x3 (k_wino_conv3x3, fp32 matrix cores)", copies, len(levels)),
            "bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": mfma_flop / avg / 1e9,
            "frac": mfma_flop / avg / 1e9 / peak, "direct_equivalent_tflops": direct_flop / avg / 1e9,
            "algorithmic_flop": mfma_flop, "direct_flop": direct_flop, "avg_launch_us": 1e3 * avg, "min_launch_us": 1e3 * ms[0],
            "tiles": tiles, "tiles_executed_with_block_padding": int(table.shape[0]) * 32, "traffic": traffic,
            # each fp32 product counted ONCE (the fp32-MFMA kernel's accounting), against the fp32 matrix peak: the figure that compares
            # the two kernels -- the split kernel spends 3 f16 partial products per fp32 product and is limited by the power cap, not by
            # issue slots (profiles/r05_k12_elimination.txt: 25 % fewer instructions per chunk bought no wall time)
            "fp32_products_tflops": mfma_flop / (SPLIT_PRODUCTS if conv.split else 1) / avg / 1e9,
            "fp32_products_vs_fp32_mfma_peak": mfma_flop / (SPLIT_PRODUCTS if conv.split else 1) / avg / 1e9 / FP32_MFMA_PEAK_TF,
            "share_of_step": "the head's 12 launches of this kernel are ~90 % of a step's GPU time (conv_census.by_kind)"}


def run_ensemble_per_gpu(args, spec, world, rank, dev):
    """cfg5 in its BASELINE topology (one seed per GPU): K timed images through apply_net.EnsemblePerGpu."""
    import torch.distributed as dist
    from pod_compare_amd import apply_net, config
    cfgdir = os.path.join(ROOT, "pod_compare_amd", "configs")
    cfg = config.setup_config(os.path.join(cfgdir, "BDD-Detection/retinanet", spec["yaml"][0]), os.path.join(cfgdir, "Inference", spec["yaml"][1]))
    cfg.MODEL.WEIGHTS, cfg.OUTPUT_DIR = "", ""          # synthetic run: seeded random-init members
    cfg.MODEL.DEVICE = str(dev)
    runner = apply_net.EnsemblePerGpu(cfg, rank, world, FRAME_HW)
    n_img = max(1, args.images)
    frames = [synthetic.synthetic_frame(i, *FRAME_HW, device=dev) for i in range(n_img)]
    with torch.no_grad():
        runner.run(max(2 * world, args.warmup), lambda i: frames[i % n_img])
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids, recs, cnts = runner.run(args.steps, lambda i: frames[i % n_img])
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt, len(ids)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the hot path")
    # POD_BENCH_BACKEND=gloo + POD_BENCH_SHARE_GPU=1: functional check of the multi-rank code path on a box with a single
    # GPU (all ranks on cuda:0, collectives staged through the host).  The real path is nccl = RCCL over xGMI.
    backend = os.environ.get("POD_BENCH_BACKEND", "nccl")
    share = os.environ.get("POD_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the rank's host threads next to its GPU (enqueue thread, torch's intra-op threads of the CPU baseline): pod_compare_amd/hostbind.py;
    # with several ranks on a node only -- a single rank keeps the whole machine (its CPU-baseline leg uses 32 threads, AN:33-40)
    from pod_compare_amd import hostbind
    host_binding = hostbind.bind_rank_to_gpu_numa(local_rank, enable=None if world > 1 else False)
    dist = None
    # POD_BENCH_FORCE_DIST=1: take the multi-rank code path (process group on device_id, device check, flush all_gather, barriers)
    # even with one rank -- under `torch.distributed.run --nproc-per-node 1` this is RCCL's first contact on a one-GPU box
    multi = world > 1 or (os.environ.get("POD_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if multi:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    stage = (lambda t: t.cpu()) if backend != "nccl" else (lambda t: t)
    spec = CONFIGS[args.config]
    N = spec["runs"]
    from pod_compare_amd import wino
    wino.SPLIT_BF16 = bool(args.split_bf16)
    rank_devices = [local_rank]
    if multi:
        # one process per GPU: every rank must sit on a device of its own (PCI bus id, not just the ordinal: two ranks that both see
        # "cuda:0" through different HIP_VISIBLE_DEVICES are fine, two ranks on the same physical device are not)
        bus = torch.cuda.get_device_properties(dev).pci_bus_id if hasattr(torch.cuda.get_device_properties(dev), "pci_bus_id") else local_rank
        mine = stage(torch.tensor([local_rank, int(bus)], device=dev, dtype=torch.int64))
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_devices = [int(t[0]) for t in every]
        if not share and len({(int(t[0]), int(t[1])) for t in every}) != world:
            raise SystemExit("bench.py --gpus %d: ranks share a device %s; launch one rank per GPU" % (world, [tuple(int(v) for v in t) for t in every]))

    if args.ensemble_per_gpu:
        if args.config != "cfg5" or world < spec["members"]:
            raise SystemExit("--ensemble-per-gpu is BASELINE configs[4]: --config cfg5 and --gpus >= %d" % spec["members"])
        # (backend nccl = RCCL moves the packed device buffers directly; any other backend stages rows through pinned host
        #  memory: the functional check on a single-GPU box, POD_BENCH_BACKEND=gloo POD_BENCH_SHARE_GPU=1)
        dt, merged_here = run_ensemble_per_gpu(args, spec, world, rank, dev)
        t = stage(torch.tensor([dt], device=dev, dtype=torch.float64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:
            print(json.dumps({
                "metric": "images/sec (ensembles pre-NMS, one seed per GPU, 1280x720)", "value": args.steps / dt, "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                "higher_is_better": True, "scaling": "fixed: 5 member ranks, the other ranks only merge", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic (seeded 1280x720 uint8 frames; random-init members seeded 0,1000,..; detections of the members' own outputs)",
                "config": {"workload": spec["name"] + ", one seed per GPU", "parallelism": "5 member ranks + rotating merge rank, RCCL p2p",
                           "rccl_ranks": world, "collective_backend": backend, "ranks_share_one_gpu": bool(share),
                           "images_in_flight": 2, "conv_net_in_timed_region": True}}))
        dist.destroy_process_group()
        return

    # ---- model (random init, seed 0) and resident inputs ------------------------------------------------
    torch.manual_seed(0)
    model = modeling.ProbabilisticRetinaNet(
        dropout_rate=spec["dropout"], cls_var_loss="loss_attenuation" if spec["cls_var"] else "none", cls_var_num_samples=10,
        bbox_cov_loss="negative_log_likelihood" if spec["reg_var"] else "none").to(dev).eval()
    modeling.fold_frozen_bn(model)   # FrozenBN folded into the conv weights (inference-only algebra, same affine map)
    members = [model]
    for seed in (1000, 2000, 3000, 4000)[: spec.get("members", 1) - 1]:       # ENSEMBLES.RANDOM_SEED_NUMS
        torch.manual_seed(seed)
        mm = modeling.ProbabilisticRetinaNet(dropout_rate=0.0, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                             bbox_cov_loss="negative_log_likelihood").to(dev).eval()
        modeling.fold_frozen_bn(mm)
        members.append(mm)
    for mm in members:
        mm.enable_graphs(not args.no_graphs)
    net_hw = A.resize_shortest_edge(*FRAME_HW)                 # 750 x 1333
    padded = A.padded_size(*net_hw)                            # 768 x 1344
    n_img = max(1, args.images)
    frames = [synthetic.synthetic_frame(rank * 100003 + i, *FRAME_HW, device=dev) for i in range(n_img)]
    heads = [synthetic.planted_head_outputs(padded, N, seed=1000 + rank * 100003 + i, num_boxes=args.boxes, with_cls_var=spec["cls_var"],
                                            with_reg_var=spec["reg_var"], mode=args.synth, device=dev) for i in range(n_img)]
    params = hotpath.PathParams()
    D = 4 if spec["reg_var"] else 0
    # one workspace per stream: images are independent units (PI:86-111), so consecutive images go to different HIP streams
    # and the low-occupancy stretches of one image's backbone overlap the other image's head convs
    n_streams = args.streams if args.streams > 0 else (3 if spec.get("members", 1) > 1 else 2 if N > 1 else 4)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    hps = [hotpath.HotPath(heads[0].shapes, heads[0].anchors, params, n_runs=N, has_cls_var=spec["cls_var"], cov_dims=D, device=dev)
           for _ in range(n_streams)]
    hp = hps[0]
    R = hp.R
    mc = N > 1 and spec["dropout"] > 0.0

    sparse_state = {"on": bool(args.sparse_bbox), "heads": heads}
    if args.sparse_bbox and (len(members) > 1 or args.no_cnn):
        raise SystemExit("--sparse-bbox: single-model configurations with the conv net in the step")

    def step_sparse(i):
        """The step with the sparse bbox tower (pod_compare_amd/sparse.py): backbone + cls side (timed, output discarded as in the dense
        step), the candidates selected from the planted class tensors, the bbox side over the blocks that can reach one (timed, output
        discarded), the rest of the path on the planted tensors."""
        from pod_compare_amd import sparse
        s = i % n_streams
        hs = sparse_state["heads"]
        with torch.cuda.stream(streams[s]):
            h = hs[i % len(hs)]

            def hook(partial):
                hps[s].select(h.cls, h.cls_var)
                return sparse.LiveBlocks(hps[s])
            model(modeling.resize_test_image(frames[i % n_img]), num_mc_dropout_runs=N, skip_unused_last_run=params.merge_quirk, sparse_bbox=hook)
            return hps[s].finish(spec["mode"], h.cls, h.delta, h.cls_var, h.reg_var, net_hw, FRAME_HW)

    def step(i):
        if sparse_state["on"]:
            return step_sparse(i)
        s = i % n_streams
        with torch.cuda.stream(streams[s]):
            if not args.no_cnn:
                img = modeling.resize_test_image(frames[i % n_img])
                # conv net: run and timed; output discarded (see docstring).  The hot path runs with the reference's merge
                # quirk, which never reads the last run's cls / cls_var / reg_var, so the head does not compute them.
                if len(members) > 1:
                    for mm in members:                                 # PI:498-500: one full forward per ensemble member
                        mm(img)
                else:
                    model(img, num_mc_dropout_runs=N, skip_unused_last_run=params.merge_quirk)
            h = heads[i % n_img]
            # K1 .. K7 of the image, enqueued by one C call (pod_run_image); fresh Philox draws per image
            return hps[s].run(spec["mode"], h.cls, h.delta, h.cls_var, h.reg_var, image_size=net_hw, out_size=FRAME_HW)

    def barrier():
        if multi:
            dist.barrier()

    with torch.no_grad():
        # priming (setup, like building the model): MIOpen resolves its solvers per handle, i.e. per stream, on the first
        # images that stream sees; two images per stream keep that out of the W warm-up steps and of the timed region
        for i in range(2 * n_streams):
            step(i)
        torch.cuda.synchronize()
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        # host time to enqueue one image (model launches or graph replay + pod_run_image), measured on an EMPTY queue, one image per
        # stream: inside the timed region a GPU-bound configuration blocks the host on a full queue, which is waiting, not work
        th = time.perf_counter()
        for i in range(n_streams):
            step(i)
        host_enqueue_ms = 1e3 * (time.perf_counter() - th) / n_streams
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dets = [step(i) for i in range(args.steps)]
        host_loop_ms = 1e3 * (time.perf_counter() - t0) / args.steps         # (includes the time the host is blocked on a full queue)
        for st in streams[1:]:
            streams[0].wait_stream(st)          # the flush below reads every stream's detections
        flush_ms = 0.0
        # the path's only collective: gather the fixed-stride detection records of this flush
        if multi:
            torch.cuda.synchronize()
            t_images = time.perf_counter() - t0
            tf = time.perf_counter()
            rec = stage(torch.stack([d.records for d in dets]))
            cnt = stage(torch.stack([d.n_det for d in dets]))
            all_rec = [torch.empty_like(rec) for _ in range(world)]
            all_cnt = [torch.empty_like(cnt) for _ in range(world)]
            dist.all_gather(all_rec, rec)
            dist.all_gather(all_cnt, cnt)
            torch.cuda.synchronize()
            flush_ms = 1e3 * (time.perf_counter() - tf)
        torch.cuda.synchronize()
        if not multi:
            t_images = time.perf_counter() - t0
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    per_rank = [args.steps / t_images]
    if multi:
        t = stage(torch.tensor([dt, flush_ms, t_images, host_enqueue_ms], device=dev, dtype=torch.float64))
        tl = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(tl, t)
        dt = max(float(x[0]) for x in tl)
        flush_ms = max(float(x[1]) for x in tl)
        per_rank = [args.steps / float(x[2]) for x in tl]
        host_enqueue_ms = max(float(x[3]) for x in tl)
    n_det_mean = float(torch.stack([d.n_det for d in dets]).float().mean().item())
    second = None
    if world == 1 and not args.no_diagnostics and not args.no_cnn and len(members) == 1 and modeling.WINO_HEAD:
        # second leg, same steps, the OTHER convolution kernel (never `value`): pod_wino_conv3x3_split when the headline ran on the
        # fp32-MFMA kernel and vice versa
        wino.SPLIT_BF16 = not args.split_bf16
        k2 = max(10, min(args.steps, 60))
        with torch.no_grad():
            for i in range(2 * n_streams + 4):       # filter transforms of the other kernel, solver caches
                step(i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(k2):
                step(i)
            torch.cuda.synchronize()
            second = {"kernel": "pod_wino_conv3x3_split" if wino.SPLIT_BF16 else "pod_wino_conv3x3", "steps": k2,
                      "value": k2 / (time.perf_counter() - t2), "unit": "images/s"}
        wino.SPLIT_BF16 = bool(args.split_bf16)

    sparse_leg = None
    if world == 1 and not args.no_diagnostics and not args.no_cnn and len(members) == 1 and not args.sparse_bbox and args.split_bf16:
        # third leg (never `value`): the same step with the sparse bbox tower, on the planted heads and on the adversarial "worst" heads
        # (every level fills its top-k with candidates all over the map: every block live -- the sparse order of evaluation then only costs)
        k3 = max(10, min(args.steps, 60))
        sparse_leg = {"what": "the step with the bbox side of the head evaluated only over the blocks that can reach a candidate (PI:310-331 reads "
                              "box_delta / box_reg_var nowhere else); detections equal the dense tower's (tests/test_sparse_tower_gpu.py)", "steps": k3, "unit": "images/s"}
        sparse_state["on"] = True
        worst = [synthetic.planted_head_outputs(padded, N, seed=77, num_boxes=args.boxes, with_cls_var=spec["cls_var"], with_reg_var=spec["reg_var"], mode="worst", device=dev)]
        with torch.no_grad():
            for name, hs in (("planted", heads), ("worst", worst)):
                sparse_state["heads"] = hs
                for i in range(2 * n_streams + 4):
                    step(i)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for i in range(k3):
                    step(i)
                torch.cuda.synchronize()
                sparse_leg["value_" + name] = k3 / (time.perf_counter() - t3)
            from pod_compare_amd import sparse as _sp
            from pod_compare_amd.wino import block_table as _bt
            lv_shapes = [tuple(sh) for sh in heads[0].shapes]
            for name, hs in (("planted", heads), ("worst", worst)):
                hps[0].select(hs[0].cls, hs[0].cls_var)
                lb = _sp.LiveBlocks(hps[0])
                sparse_leg["live_blocks_" + name] = {"predictors": lb.fraction(_bt(lv_shapes, 1, dev), 0), "subnet_first_conv": lb.fraction(_bt(lv_shapes, 1, dev), 4)}
                hps[0]._dirty = True
                hps[0]._clean()
        del worst
        sparse_state["on"], sparse_state["heads"] = False, heads

    out = {
        "metric": "images/sec (BayesOD+MC-dropout, 1280x720)" if args.config == "cfg3" else "images/sec (%s, 1280x720)" % spec["mode"],
        "value": world * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if not args.split_bf16 else "f32 (convolutions: every fp32 product from 2-way f16 splits of the power-of-two-scaled fp32 operands, 3 partial products, fp32 accumulate; closer to fp64 than the fp32 MFMA: profiles/r05_f16_split_numerics.txt)",
        "data": "synthetic (seeded 1280x720 uint8 frames; random-init weights; planted-object head tensors, SURVEY 8d)",
        "config": {"workload": spec["name"], "frame": "1280x720 -> 750x1333 -> padded 768x1344", "anchors_R": R, "mc_runs": N,
                   "classes": params.num_classes, "synthetic_mode": args.synth, "planted_boxes": args.boxes, "conv_net_in_timed_region": not args.no_cnn,
                   "conv_net_output": "computed and timed, then discarded: the hot path consumes planted head tensors (random-init "
                                      "weights give no detections, SURVEY 8d)",
                   "skip_unused_last_run": bool(mc and params.merge_quirk),
                   "skip_unused_last_run_note": "the reference's merge (PI:216-222) never reads run N-1 of cls / cls_var / reg_var; "
                                                "the head does not compute those 3 of its 4N subnet evaluations",
                   "members_on_this_gpu": len(members),
                   "images_per_gpu_step": 1, "streams_per_gpu": n_streams, "parallelism": "image-sharded dp%d" % world,
                   "rccl_ranks": world, "collective_backend": backend if multi else None, "rank_devices": rank_devices, "host_binding": host_binding,
                   "ranks_share_one_gpu": bool(share and world > 1),
                   "conv3x3_kernel": "pod_wino_conv3x3_split (fp32 Winograd; every product from 2-way f16 splits of the scaled operands on the f16 matrix cores, 3 partial "
                                     "products, fp32 accumulate; per shape at least as close to fp64 as the fp32-MFMA kernel: tests/test_wino_conv_gpu.py); "
                                     "the fp32-MFMA kernel's figure: `fp32_mfma`" if args.split_bf16 else
                                     "pod_wino_conv3x3 (fp32 matrix instructions); the split kernel's figure: `split_f16`",
                   "rng": "in-kernel Philox4x32-10, fresh key per image",
                   "backbone": ("channels-last from the frame on: pod_stem7x7_split + pod_maxpool3x3s2_cl, pod_conv1x1_split, pod_wino_conv3x3_split (p6 / p7: PyTorch-ROCm)"
                                if (modeling.CL_BACKBONE and args.split_bf16) else "NCHW: 1x1 / strided convolutions on PyTorch-ROCm, 3x3 / stride-1 on the Winograd kernel"),
                   "model_forward": "HIP graph replay per (stream, shape)" if (not args.no_graphs and not args.no_cnn) else "eager launches from Python"},
        "per_rank_images_per_s": per_rank, "flush_ms": flush_ms if multi else None, "host_enqueue_ms_per_image": host_enqueue_ms,
        "host_loop_ms_per_image": host_loop_ms,
        "mean_detections": n_det_mean,
    }
    if second is not None:
        out["split_f16" if second["kernel"].endswith("_split") else "fp32_mfma"] = second
    if sparse_leg is not None:
        out["sparse_bbox_tower"] = sparse_leg
    out["config"]["bbox_tower"] = "sparse (--sparse-bbox)" if args.sparse_bbox else "dense (the reference's evaluation order; the sparse tower's figures: `sparse_bbox_tower`)"
    out["parity_bar"] = parity_bar()
    if rank == 0 and not args.no_diagnostics:
        diagnostics(args, spec, out, model, frames, heads, hps, params, net_hw, dev, R, N, D, mc)
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def parity_bar():
    """How the tests read north_star's "within 1e-4 on box means / covariances", and how much of that bar the last recorded GPU suite used."""
    rec = {"reading": "|hip - ref| <= 1e-4 * max(1, |ref|) per element (tests/helpers.py: assert_close) -- the builder's reading: fp32 moments of coordinates up to "
                      "~1 300 px; an ABSOLUTE 1e-4 there is below the fp32 spacing (1.2e-4); indices, classes and NMS keep lists: bit-exact"}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_parity_errors.json")), reverse=True):
        rows = json.load(open(path))
        worst = max(rows, key=lambda r: r["worst_fraction_of_bound"]) if rows else None
        if worst:
            rec.update(source=os.path.relpath(path, ROOT), worst_fraction_of_bar_used=worst["worst_fraction_of_bound"], worst_quantity=worst["quantity"],
                       worst_abs_error=max(r["max_abs"] for r in rows))
        break
    return rec


def diagnostics(args, spec, out, model, frames, heads, hps, params, net_hw, dev, R, N, D, mc):
    """Everything measured outside the timed region, on rank 0: the hot path alone, K1 against the HBM roofline, the conv
    net against the fp32 MFMA peak, NLL of the detections against the planted ground truth, worst-case inputs, and the
    CPU baseline (the oracle on the host cores)."""
    hp, n_img = hps[0], len(heads)
    run = lambda h, **kw: hp.run(spec["mode"], h.cls, h.delta, h.cls_var, h.reg_var, image_size=net_hw, out_size=FRAME_HW, **kw)

    def time_hot_path(hs, iters):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i in range(5):
            run(hs[i % len(hs)])
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(iters):
            ev[i][0].record()
            run(hs[i % len(hs)])
            ev[i][1].record()
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t) / iters
        return sum(a.elapsed_time(b) for a, b in ev) / iters, wall

    # ---- the hot path alone (no conv net, one stream): HIP events around each image -----------------------
    with torch.no_grad():
        hp_ms, hp_wall_ms = time_hot_path(heads, max(20, min(args.steps, 200)))
        out["hot_path_ms_per_image"], out["hot_path_wall_ms_per_image"] = hp_ms, hp_wall_ms
        if args.synth == "planted":
            # adversarial inputs: (almost) every anchor passes the threshold on every level, n = 4 594 candidates
            worst = synthetic.planted_head_outputs(A.padded_size(*net_hw), N, seed=4242, num_boxes=24, with_cls_var=spec["cls_var"],
                                                   with_reg_var=spec["reg_var"], mode="worst", device=dev)
            out["hot_path_worst_ms"], _ = time_hot_path([worst], 20)
            del worst

    # ---- K1 alone, HIP events on the launch stream, rotating over the distinct input sets ------------------
    k1_iters = max(20, min(args.steps, 200))
    lib, P = hp.lib, hotpath.hip.ptr
    lvs = [hp._levels(h.cls, h.delta, h.cls_var, h.reg_var, None) for h in heads]
    st = hotpath.hip.current_stream()
    prune = spec["cls_var"]   # native RNG + variance head: K1 runs in prune mode, exactly as in the timed steps
    # the product path merges box_delta / box_reg_var at the candidates (K2b); the dense variant also writes their merged
    # planes for every anchor, as PI:243-270 does (HotPath(dense_box_merge=True)).  Both are timed; `roofline` is the
    # product path's launch, `roofline_dense_merge` the reference-shaped dense merge of all 2K+4+D channels.
    dense_delta = torch.empty(R * 4, dtype=torch.float32, device=dev) if N > 1 else None
    dense_reg = torch.empty(R * D, dtype=torch.float32, device=dev) if N > 1 and D > 0 else None

    def time_k1(mean_delta, mean_reg_var, with_score=False, fused=None):
        def k1_call(j):
            if fused is not None:     # pod_merge_score_fused: the ONE launch pod_run_image enqueues (fused = "store the merged planes too")
                hotpath.hip.check(lib.pod_merge_score_fused(hp.cfg, lvs[j % n_img], P(hp.mean_cls) if fused else None,
                                                            P(hp.mean_cls_var) if fused and spec["cls_var"] else None, P(hp.cand_keys),
                                                            P(hp.cand_count), P(hp.probs_dense) if spec["cls_var"] else None, st), "pod_merge_score_fused")
                return
            hotpath.hip.check(lib.pod_mc_merge_score(hp.cfg, lvs[j % n_img], P(hp.mean_cls), P(hp.mean_cls_var), P(mean_delta),
                                                     P(mean_reg_var), P(hp.cand_keys), P(hp.cand_count),
                                                     P(hp.maybe_bits) if prune else None, st),
                              "pod_mc_merge_score")
            if with_score:     # merge AND score, the job SURVEY 8 a3 + a4 defines: K1b samples the anchors K1 flagged
                hotpath.hip.check(lib.pod_score_maybe(hp.cfg, lvs[j % n_img], P(hp.mean_cls), P(hp.mean_cls_var), P(hp.maybe_bits),
                                                      P(hp.cand_keys), P(hp.cand_count), P(hp.probs_dense), st), "pod_score_maybe")

        for j in range(3):
            lib.pod_reset_counters(P(hp.counters), 8, st)
            k1_call(j)
        torch.cuda.synchronize()
        # HIP events on the launch stream.  The host must stay AHEAD of the device (otherwise the start event fires
        # before the kernel has even been enqueued and the pair measures host launch latency): park the stream behind
        # a ~1 ms spin kernel, enqueue everything, synchronise once.  Each sample = one event pair around K1B
        # back-to-back launches (rotating over the distinct input sets), so the per-event overhead (~3 us on a 30 us
        # kernel when every launch is bracketed) is amortised and the figure is comparable with rocprofv3's average.
        K1B = 10
        n_batches = max(4, k1_iters // K1B)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_batches)]
        torch.cuda._sleep(3_000_000)
        for bidx in range(n_batches):
            lib.pod_reset_counters(P(hp.counters), 8, st)
            evs[bidx][0].record()
            for j in range(K1B):
                if not prune:
                    lib.pod_reset_counters(P(hp.counters), 8, st)   # dense-scoring mode appends candidates: keep the lists bounded
                k1_call(bidx * K1B + j)     # (with_score: K1b appends ~10 x 400 keys per batch into lists sized for every anchor)
            evs[bidx][1].record()
        torch.cuda.synchronize()
        lib.pod_reset_counters(P(hp.counters), int(hp.counters.numel()), st)
        if prune:
            hp.maybe_bits.zero_()
        ms = sorted(a.elapsed_time(b) / K1B for a, b in evs)
        return sum(ms) / len(ms), ms[0]

    K = params.num_classes
    k1_avg_ms, k1_min_ms = time_k1(hp.mean_delta, hp.mean_reg_var)
    k1_bytes = k1_algorithmic_bytes(R, K, D, N, spec["cls_var"], params.merge_quirk, dense_box=hp.dense_box_merge)
    kd_avg_ms, kd_min_ms = time_k1(dense_delta, dense_reg)
    ks_avg_ms, ks_min_ms = time_k1(hp.mean_delta, hp.mean_reg_var, with_score=True) if prune else (k1_avg_ms, k1_min_ms)
    kf_avg_ms, kf_min_ms = time_k1(None, None, fused=False)                                  # the product launch: merge + score, planes not stored
    kfp_avg_ms, kfp_min_ms = time_k1(None, None, fused=True) if N > 1 else (kf_avg_ms, kf_min_ms)   # ... with the merged class planes stored as well
    kf_bytes = k1_bytes - (4 * R * K * (2 if spec["cls_var"] else 1) if N > 1 else 0)      # the same reads, no merged planes written
    kd_bytes = k1_algorithmic_bytes(R, K, D, N, spec["cls_var"], params.merge_quirk, dense_box=True)
    # HBM traffic of K1 comes from separate rocprofv3 --pmc passes (a counter run cannot share a process with this timing
    # run); the committed summary of the latest pass is read back when it was taken on this very workload.
    traffic, traffic_src, traffic_dense, traffic_fused = args.k1_traffic_bytes, "--k1-traffic-bytes", None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_k1_traffic.json")), reverse=True):
        t = json.load(open(path))
        w = t.get("workload", {})
        if w.get("anchors_R") == R and w.get("mc_runs") == N and w.get("config") == args.config and w.get("synthetic_mode") == args.synth:
            key = "k1_dense_traffic_bytes" if hp.dense_box_merge else "k1_class_traffic_bytes"
            if traffic is None and t.get(key) is not None:
                traffic, traffic_src = t[key], os.path.relpath(path, ROOT) + ": " + t.get("source", "")
            traffic_dense = t.get("k1_dense_traffic_bytes")
            traffic_fused = t.get("k1_fused_traffic_bytes")
            break
    achieved = k1_bytes / (k1_avg_ms * 1e-3) / 1e9
    out["roofline"] = {"kernel": "pod_mc_merge_score (k1_prune_stream)" if prune else "pod_mc_merge_score (k1_mc_merge_score)",
                       "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                       "algorithmic_bytes": k1_bytes, "avg_launch_us": 1e3 * k1_avg_ms, "min_launch_us": 1e3 * k1_min_ms,
                       "channels_streamed": "2K class channels (box_delta / box_reg_var are merged at the candidates by K2b)"
                                            if not hp.dense_box_merge else "2K+4+D",
                       # merge AND score (SURVEY 8 a3 + a4) as the product path runs it since round 4: ONE streaming launch
                       # (pod_merge_score_fused); algorithmic bytes = the runs it must read (+ the merged planes when they are stored)
                       "merge_and_score": {"what": "pod_merge_score_fused: merge + score in one launch (what pod_run_image enqueues); the merged planes "
                                                   "are not stored (nothing downstream reads them)",
                                           "algorithmic_bytes": kf_bytes, "traffic": traffic_fused, "avg_us": 1e3 * kf_avg_ms, "min_us": 1e3 * kf_min_ms,
                                           "achieved": kf_bytes / (kf_avg_ms * 1e-3) / 1e9, "frac": kf_bytes / (kf_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "with_merged_planes_stored": {"algorithmic_bytes": k1_bytes, "avg_us": 1e3 * kfp_avg_ms,
                                                                         "frac": k1_bytes / (kfp_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                           "two_launch_form": {"what": "pod_mc_merge_score + pod_score_maybe back to back (rounds 1-3)",
                                                               "algorithmic_bytes": k1_bytes, "avg_us": 1e3 * ks_avg_ms, "min_us": 1e3 * ks_min_ms,
                                                               "frac": k1_bytes / (ks_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}}
    # the same kernel asked for the reference-shaped dense merge of every channel (PI:211-270), for comparison
    out["roofline_dense_merge"] = {"kernel": "pod_mc_merge_score, mean_delta / mean_reg_var requested", "bound": "hbm",
                                   "achieved": kd_bytes / (kd_avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": kd_bytes / (kd_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_dense,
                                   "algorithmic_bytes": kd_bytes, "avg_launch_us": 1e3 * kd_avg_ms, "min_launch_us": 1e3 * kd_min_ms,
                                   "survey_bytes_4RC(N+1)": 4 * R * (K * (2 if spec["cls_var"] else 1) + 4 + D) * (N + 1)}

    # ---- the kernel that owns the step: pod_wino_conv3x3 on the head trunk (HIP events per launch) ------------
    if not args.no_cnn and spec.get("members", 1) == 1 and modeling.WINO_HEAD:
        # `roofline` is the step's dominant kernel by time (pod_wino_conv3x3: ~90 % of the GPU time, MFMA-bound); the dominant kernel of the
        # post-processing chain K1..K7 (SURVEY 8d's merge + score, HBM-bound; `roofline` of round 1) is kept as `roofline_k1`
        out["roofline_k1"] = out["roofline"]
        out["roofline_k1"]["scope"] = "dominant kernel of the post-processing chain K1..K7 (SURVEY 8d's merge + score)"
        out["roofline"] = head_conv_roofline(model, net_hw, N, params.merge_quirk, dev)
        out["roofline"]["scope"] = "dominant kernel of the step by time (the head's convolutions); SURVEY 8(d)'s kernel (K1): roofline.k1"
        k1r = out["roofline_k1"]
        out["roofline"]["k1"] = {k: k1r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes",
                                                      "avg_launch_us", "merge_and_score")}
        # SURVEY 8(d)'s object in its two forms, side by side: GATE form = the reference-shaped dense merge of all 2K + 4 + D channels
        # (B_K1 = 4 R C (N + 1): what north_star's >= 0.60 is written for); PRODUCT form = what pod_run_image enqueues (class channels
        # only, merge + score fused, nothing stored)
        gate = out["roofline_dense_merge"]
        out["roofline"]["k1"]["gate_form"] = {"what": "pod_mc_merge_score with every channel merged densely (PI:211-270 as written; SURVEY 8(d)'s B_K1)",
                                              "algorithmic_bytes": gate["algorithmic_bytes"], "avg_launch_us": gate["avg_launch_us"], "frac": gate["frac"],
                                              "traffic": gate["traffic"]}
        out["roofline"]["k1"]["product_form"] = {"what": "pod_merge_score_fused as pod_run_image enqueues it (2K class channels, merge + score in one launch, no planes stored)",
                                                 "algorithmic_bytes": k1r["merge_and_score"]["algorithmic_bytes"], "avg_launch_us": k1r["merge_and_score"]["avg_us"],
                                                 "frac": k1r["merge_and_score"]["frac"], "traffic": k1r["merge_and_score"]["traffic"],
                                                 "note": "its streaming part alone runs at 0.62 (19.5 us, -DPOD_K1F_NOSCORE build, profiles/r05_k1f_variants.txt); the "
                                                         "remaining 3.5 us are one scoring round (10 Philox / Box-Muller samples per class), one barrier and the emission"}
        out["k1_hbm_frac"], out["k1_merge_and_score_hbm_frac"] = k1r["frac"], k1r["merge_and_score"]["frac"]
        out["roofline_head_conv"] = out["roofline"]
        other = "split_f16" if not args.split_bf16 else "fp32_mfma"
        if other in out:          # the same launch on the other kernel, for the second leg's record
            from pod_compare_amd import wino as _w
            _w.SPLIT_BF16 = not args.split_bf16
            r2 = head_conv_roofline(model, net_hw, N, params.merge_quirk, dev)
            _w.SPLIT_BF16 = bool(args.split_bf16)
            out[other]["head_conv_launch"] = {k: r2[k] for k in ("kernel", "peak", "achieved", "frac", "direct_equivalent_tflops", "avg_launch_us", "min_launch_us",
                                                                 "fp32_products_vs_fp32_mfma_peak")}

    # ---- the whole conv net of a step against the fp32 MFMA peak ---------------------------------------------
    if not args.no_cnn and spec.get("members", 1) == 1:
        census = conv_census(model, modeling.resize_test_image(frames[0]), N, params.merge_quirk, dev)
        gflop = sum(d["gflop"] for d in census.values())
        conv_ms = sum(d["ms"] for d in census.values())
        step_tf = gflop / out["ms_per_step"]     # GFLOP / ms = TFLOP/s, whole step (all streams overlapped, hot path included)
        out["conv_census"] = {"unit": "TFLOP/s (direct-convolution equivalent)", "dtype": "f32",
                                "gflop_per_image": gflop, "direct_equivalent_tflops": step_tf,
                                "direct_equivalent_over_fp32_mfma_peak": step_tf / FP32_MFMA_PEAK_TF,
                                "basis": "direct-convolution FLOPs of one image (2*N*Cout*Hout*Wout*Cin*kh*kw of every conv call) / ms_per_step: a THROUGHPUT "
                                         "figure, not a roofline fraction -- the Winograd kernels execute 24/72 of these multiply-adds and on the 16-bit "
                                         "matrix cores, so the ratio to the fp32 matrix peak exceeds 1 by construction (rounds 1-4 printed it as "
                                         "`conv_roofline.frac`); the roofline objects are `roofline`, `roofline_backbone`, `roofline_conv1x1`",
                                "conv_ms_per_image_one_stream": conv_ms,
                                "one_stream_direct_equivalent_tflops": gflop / conv_ms if conv_ms > 0 else None,
                                "by_kind": census}
        classes = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_conv_classes.json")), reverse=True):
            classes, classes_src = json.load(open(path)), os.path.relpath(path, ROOT)
            break

        def class_roofline(kind):
            """Per-class bounds of profiles/*_conv_classes.json (rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of tools/conv_classes.sh on this
            tree): bound = max(algorithmic bytes / 8 TB/s, executed f16 FLOP / 2.5 PFLOP/s) per class; achieved / bound over the image's calls."""
            if not classes:
                return {"traffic": None}
            sel = [c for c in classes if c["kind"] == kind]
            t_us = sum(c["us_per_launch"] * c["calls_per_image"] for c in sel)
            b_us = sum(c["bound_us"] * c["calls_per_image"] for c in sel)
            worst = min(sel, key=lambda c: c["frac_of_bound"])
            best = max(sel, key=lambda c: c["frac_of_bound"])
            return {"traffic": sum(c["traffic_bytes"] * c["calls_per_image"] for c in sel),
                    "algorithmic_bytes": sum(c["bytes"] * c["calls_per_image"] for c in sel),
                    "per_class_bound": {"source": classes_src, "classes": len(sel), "sum_of_bounds_ms_per_image": b_us / 1e3, "profiled_ms_per_image": t_us / 1e3,
                                        "achieved_over_bound": b_us / t_us, "best_class": [best["class"], best["frac_of_bound"]],
                                        "worst_class": [worst["class"], worst["frac_of_bound"]],
                                        "reading": "the large maps (res2) stream at 0.5-0.6 of HBM; from res3 down a launch is 60-1000 one-wavefront tiles each "
                                                   "walking its whole K chain (0.3-0.4 us per 16 channels): latency of a lone wavefront, not a roofline"}}

        bb = census.get("3x3_backbone_fpn_winograd_hip")
        if bb:      # the bottlenecks' and the FPN's 3x3 / stride-1 convolutions on pod_wino_conv3x3: one NCHW feature map per launch
            out["roofline_backbone"] = {"kernel": "pod_wino_conv3x3 on the backbone's and the FPN's 3x3 / stride-1 convolutions (%d launches per image, "
                                                  "one feature map each, batch 1: 24-1008 workgroups; small maps cut over their input channels: pod_wino_conv3x3_split_partial + pod_wino_reduce)" % bb["calls"],
                                        "bound": "mfma", "unit": "TFLOP/s", "peak": BF16_MFMA_PEAK_TF if args.split_bf16 else FP32_MFMA_PEAK_TF,
                                        "achieved": bb["mfma_tflops_executed"] * (SPLIT_PRODUCTS if args.split_bf16 else 1.0),
                                        "frac": bb["mfma_tflops_executed"] * (SPLIT_PRODUCTS if args.split_bf16 else 1.0) / (BF16_MFMA_PEAK_TF if args.split_bf16 else FP32_MFMA_PEAK_TF),
                                        "fp32_products_vs_fp32_mfma_peak": bb["mfma_tflops_executed"] / FP32_MFMA_PEAK_TF,     # every fp32 product counted once
                                        "direct_equivalent_tflops": bb["tflops"], "gflop_direct": bb["gflop"], "ms_per_image": bb["ms"],
                                        "note": "HIP events around each launch on one stream (launch gaps included); MIOpen's Winograd on the same "
                                                "convolutions ran at 82 TFLOP/s direct-equivalent in round 2"}
            out["roofline_backbone"].update(class_roofline("3x3"))

        c1 = census.get("1x1_hip")
        if c1:      # the bottlenecks' 1x1 convolutions, shortcuts and FPN laterals on pod_conv1x1_split (channels-last GEMM, 3 f16 partial products)
            out["roofline_conv1x1"] = {"kernel": "pod_conv1x1_split on the backbone's 1x1 convolutions and the FPN laterals (%d calls per image, batch 1; "
                                                 "small maps cut over their input channels: + pod_reduce_partials)" % c1["calls"],
                                       "bound": "mfma", "unit": "TFLOP/s", "peak": BF16_MFMA_PEAK_TF, "achieved": c1["tflops"] * SPLIT_PRODUCTS,
                                       "frac": c1["tflops"] * SPLIT_PRODUCTS / BF16_MFMA_PEAK_TF, "fp32_products_vs_fp32_mfma_peak": c1["tflops"] / FP32_MFMA_PEAK_TF,
                                       "gflop": c1["gflop"], "ms_per_image": c1["ms"],
                                       "note": "HIP events around each call on one stream (launch gaps and reduce launches included); the res2 calls are HBM-bound "
                                               "(per_class_bound)"}
            out["roofline_conv1x1"].update(class_roofline("1x1"))

    # ---- NLL of the detections against the planted ground truth (the "NLL parity" half of the metric) ---------
    if spec["reg_var"] or N > 1:
        from pod_compare_amd import evaluation_utils as ev
        n_nll = min(2, n_img)
        with torch.no_grad():
            nat = [run(heads[i], draw_id=i) for i in range(n_nll)]
            rep = [run(heads[i], eps_fn=synthetic.SeededNormals(90000 + i)) for i in range(n_nll)]
        gts = [heads[i] for i in range(n_nll)]
        nll_nat = ev.score_against_planted(nat, gts, net_hw, FRAME_HW)
        nll_rep = ev.score_against_planted(rep, gts, net_hw, FRAME_HW)
        out["nll"] = {"rule": "mean over true positives of -log N(gt; mean, cov + 1e-2 I) (scoring_rules.py:68-74), "
                              "matching evaluation_utils.py:191-367 (IoU >= 0.7), planted boxes as ground truth",
                      "images": n_nll, "hip_native_rng": nll_nat, "hip_eps_replay": nll_rep}

    # ---- CPU baseline: the oracle (port of the reference's CPU path) on this host's cores, rank 0, N=1 ----------
    if out["n_gpus"] == 1 and not args.no_cpu_baseline:
        from oracle import pod_oracle as po
        cores = os.cpu_count() or 1
        threads = min(cores, 32)   # the reference pins torch.set_num_threads(32), apply_net.py:33-40
        torch.set_num_threads(threads)
        op = po.PathParams()
        n_cpu, t_cpu = 0, 0.0
        cpu_sets = []
        for h in heads:                                   # host copies in the reference's (1, HWA, C) layout, made once
            hc = h.to("cpu")
            cpu_sets.append([synthetic.to_reference_layout(hc, r) for r in range(N)])
        for i in range(args.cpu_images):
            runs = cpu_sets[i % n_img]
            t1 = time.perf_counter()
            po.predict(spec["mode"], op, net_hw, FRAME_HW, outputs=runs[0] if N == 1 else None,
                       run_outputs=runs if N > 1 else None)
            t_cpu += time.perf_counter() - t1
            n_cpu += 1
            if t_cpu > 12.0:
                break
        out["cpu_baseline"] = {"value": n_cpu / t_cpu, "unit": "images/s", "cores": threads, "kind": "port",
                               "host_cpus": cores,
                               "sample": "%d images, post-processing only (head tensors given; conv net excluded), torch CPU "
                                         "oracle/pod_oracle.py, same planted tensors" % n_cpu,
                               "gpu_hot_path_images_per_s": 1e3 / hp_wall_ms}
        if "nll" in out:
            # the same eps-replay images through the CPU oracle and the oracle's scoring rule: the other backend's NLL
            vals = []
            for i in range(out["nll"]["images"]):
                ref = po.predict(spec["mode"], op, net_hw, FRAME_HW, outputs=cpu_sets[i][0] if N == 1 else None,
                                 run_outputs=cpu_sets[i] if N > 1 else None, eps_fn=synthetic.SeededNormals(90000 + i))
                vals.append((ref.pred_boxes, ref.pred_cls_probs, ref.pred_boxes_covariance))
            cpu_nll = po.score_against_planted(vals, [heads[i].to("cpu") for i in range(out["nll"]["images"])], net_hw, FRAME_HW)
            out["nll"]["cpu_oracle_eps_replay"] = cpu_nll
            a, b = out["nll"]["hip_eps_replay"]["nll"], cpu_nll["nll"]
            out["nll"]["abs_delta_hip_vs_cpu"] = abs(a - b) if a is not None and b is not None else None


if __name__ == "__main__":
    main()
