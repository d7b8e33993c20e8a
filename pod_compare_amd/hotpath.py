"""Host side of the MI355X hot path: owns the HBM workspace of one image geometry and enqueues
the kernel chain K1..K7 (include/pod_mi355x.h) on the current HIP stream.

Nothing here computes on the CPU and nothing synchronises in native-RNG mode: counts stay in
device words, all buffers are sized for the worst case once (288 GB of HBM: the workspace of the
BASELINE geometry is ~25 MB), so the whole chain is hipGraph-capturable (`HotPath.capture`).
The eps-replay parity mode needs one host sync to learn the candidate count n before the
(1000, n, 4) normal tensor can be drawn -- exactly where the reference draws it
(probabilistic_inference.py:351-356).
"""
import functools
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import hip

MODES = ("standard_nms", "mc_dropout_ensembles", "anchor_statistics", "ensembles", "bayes_od")


@dataclass
class PathParams:
    """Model attributes / config keys the path reads (defaults of core/setup.py:90-133
    and detectron2's RetinaNet defaults)."""
    num_classes: int = 7
    num_anchors: int = 9
    topk_candidates: int = 1000          # model.test_topk_candidates (PI:300)
    score_thresh: float = 0.05           # model.test_score_thresh (PI:304)
    nms_thresh: float = 0.5              # model.test_nms_thresh
    max_detections: int = 100            # model.max_detections_per_image
    cls_var_num_samples: int = 10        # model.cls_var_num_samples (PI:294)
    prop_num_samples: int = 1000         # hard-coded at PI:355
    affinity_thresh: float = 0.9         # PROBABILISTIC_INFERENCE.AFFINITY_THRESHOLD
    merge_quirk: bool = True             # PI:216-222; False = true mean over runs
    box_weights: Tuple[float, float, float, float] = (1.0, 1.0, 1.0, 1.0)
    philox_seed: int = 0x5EED


class DeviceDetections:
    """Fixed-capacity detection buffers in HBM + device count (no host sync until `.count()`).

    One allocation per image; the field tensors are views made on first access (the launch path only needs the
    pointers, so a throughput loop that reads `records` / `n_det` never pays for the other views)."""
    __slots__ = ("image_size", "buf", "K", "md", "_n", "_views")

    def __init__(self, image_size: Tuple[int, int], num_classes: int, max_det: int, device):
        self.image_size, self.K, self.md, self._n, self._views = image_size, num_classes, max_det, None, {}
        self.buf = torch.empty(self.layout(num_classes, max_det)["end"][0], dtype=torch.float32, device=device)

    @staticmethod
    @functools.lru_cache(maxsize=None)
    def layout(K: int, md: int):
        """name -> (offset in floats, shape, elements); 16-byte aligned segments."""
        out, off = {}, 0
        for name, shape in (("boxes", (md, 4)), ("cov", (md, 4, 4)), ("scores", (md,)), ("classes", (md,)), ("probs", (md, K)),
                            ("records", (md, 6 + K + 16)), ("n_det", ())):
            n = 1
            for d in shape:
                n *= d
            out[name] = (off, shape, n)
            off += (n + 3) // 4 * 4
        out["end"] = (off, (), 0)
        return out

    def ptr(self, name: str) -> int:
        return self.buf.data_ptr() + 4 * self.layout(self.K, self.md)[name][0]

    def _view(self, name: str) -> torch.Tensor:
        v = self._views.get(name)
        if v is None:
            off, shape, n = self.layout(self.K, self.md)[name]
            v = self.buf[off:off + n]
            if name in ("classes", "n_det"):
                v = v.view(torch.int32)
            v = self._views[name] = v.view(shape)
        return v

    boxes = property(lambda self: self._view("boxes"))        # (max_det, 4) fp32 XYXY in output pixels
    cov = property(lambda self: self._view("cov"))            # (max_det, 4, 4)
    scores = property(lambda self: self._view("scores"))      # (max_det,)
    classes = property(lambda self: self._view("classes"))    # (max_det,) int32
    probs = property(lambda self: self._view("probs"))        # (max_det, K)
    records = property(lambda self: self._view("records"))    # (max_det, 6 + K + 16) fixed-stride JSON payload (XYWH box, T cov T^T)
    n_det = property(lambda self: self._view("n_det"))        # () int32 on device

    def count(self) -> int:
        if self._n is None:
            self._n = int(self.n_det.item())   # the one device->host sync of an image
        return self._n


class HotPath:
    """Workspace + launcher for one (level geometry, head configuration)."""

    def __init__(self, shapes: Sequence[Tuple[int, int]], anchors: Sequence[torch.Tensor], params: PathParams,
                 n_runs: int = 1, has_cls_var: bool = False, cov_dims: int = 0, device="cuda", dense_box_merge: bool = False):
        """dense_box_merge: also have K1 write the merged box_delta / box_reg_var planes (PI:243-270) for every anchor.
        Nothing downstream reads them -- K2b evaluates the same merge, in the same order, at the <= L*topk candidates
        (k2_topk_gather.hip) -- so the product path leaves it off and K1 streams the 2K class channels only
        (108 MB instead of 170 MB per image at BASELINE size).  On = the full dense merge of the reference."""
        self.lib = hip.load()
        self.dense_box_merge = bool(dense_box_merge)
        self.p = params
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.shapes = [tuple(s) for s in shapes]
        self.L = len(self.shapes)
        self.n_runs, self.has_cls_var, self.cov_dims = int(n_runs), bool(has_cls_var), int(cov_dims)
        A, K = params.num_anchors, params.num_classes
        if self.L > hip.POD_MAX_LEVELS or K > hip.POD_MAX_CLASSES - 1 or self.n_runs > hip.POD_MAX_RUNS:
            raise hip.PodError("unsupported geometry: levels={} (max {}), classes={} (max {}: the kernels put an anchor's classes on the "
                               "lanes of one 16-lane row; BDD has 7), runs={} (max {})".format(
                                   self.L, hip.POD_MAX_LEVELS, K, hip.POD_MAX_CLASSES - 1, self.n_runs, hip.POD_MAX_RUNS))
        self.level_R = [h * w * A for h, w in self.shapes]
        self.anchor_base = [0]
        for r in self.level_R[:-1]:
            self.anchor_base.append(self.anchor_base[-1] + r)
        self.R = sum(self.level_R)
        self.n_cap = self.L * params.topk_candidates
        if params.topk_candidates > hip.POD_MAX_TOPK or self.n_cap > hip.POD_MAX_CANDIDATES:
            raise hip.PodError("topk_candidates={} x {} levels exceeds the kernel capacity".format(params.topk_candidates, self.L))
        dev = self.device
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        self.anchors = torch.cat([a.to(dev, torch.float32) for a in anchors]).contiguous()
        assert self.anchors.shape == (self.R, 4)
        D = self.cov_dims
        # K1 outputs
        merged = self.n_runs > 1
        # merged planes + the K1 -> K1b bitmap: only the two-launch form (pod_mc_merge_score + pod_score_maybe; eps replay, dense-merge
        # gate) reads or writes them -- pod_run_image's fused launch stores no planes -- so they are allocated on first use
        self._planes = None         # see _merged_planes(); read through the properties mean_cls / mean_cls_var / mean_delta / mean_reg_var / maybe_bits
        self.cand_keys = torch.empty(self.R, dtype=torch.int64, device=dev)
        self.counters = torch.zeros(2 * hip.POD_MAX_LEVELS, dtype=torch.int32, device=dev)   # [0:L] cand_count, [L:2L] K2's tickets
        self.cand_count = self.counters[: self.L]
        # K2 outputs
        self.sel_keys = torch.empty(self.L * params.topk_candidates, dtype=torch.int64, device=dev)
        self.sel_count = i32(self.L)
        self.cat_keys = torch.empty(self.L * params.topk_candidates, dtype=torch.int64, device=dev)   # level-concatenated selection
        self.cat_level = i32(self.L * params.topk_candidates)
        # class probabilities of the anchors K1b emits, reused by the gather kernel (native draws + variance head)
        self.probs_dense = f32(self.R * K) if has_cls_var else None
        n = self.n_cap
        self.n_total = i32(1)
        self.cand_anchor_idx, self.cand_level, self.cand_class = i32(n), i32(n), i32(n)
        self.cand_score, self.cand_probs = f32(n), f32(n, K)
        self.cand_delta, self.cand_anchor = f32(n, 4), f32(n, 4)
        self.cand_reg_var = f32(n, max(D, 1))
        self.cand_run_delta = f32(n, self.n_runs, 4) if merged else None
        # K3 outputs
        self.boxes, self.cov = f32(n, 4), f32(n, 4, 4)
        # K4
        self.keep, self.n_keep = i32(hip.POD_MAX_DETECTIONS), i32(1)
        self.nms_scratch = torch.zeros(self.lib.pod_nms_scratch_bytes(n), dtype=torch.uint8, device=dev)     # zeroed once (generation word)
        # K5/K6 outputs
        md = hip.POD_MAX_DETECTIONS
        self.m_boxes, self.m_cov, self.m_scores = f32(md, 4), f32(md, 4, 4), f32(md)
        self.m_classes, self.m_probs = i32(md), f32(md, K)
        self.cfg = self._make_cfg()
        self._levels_t = hip.PodLevel * self.L
        ws = hip.PodWorkspace()
        for name, t in (("anchors", self.anchors), ("mean_cls", None), ("mean_cls_var", None),
                        ("mean_delta", None), ("mean_reg_var", None), ("cand_keys", self.cand_keys),
                        ("cand_count", self.cand_count), ("maybe_bits", None), ("sel_keys", self.sel_keys),
                        ("sel_count", self.sel_count), ("cat_keys", self.cat_keys), ("cat_level", self.cat_level),
                        ("probs_dense", self.probs_dense), ("n_total", self.n_total), ("cand_anchor_idx", self.cand_anchor_idx),
                        ("cand_level", self.cand_level), ("cand_class", self.cand_class), ("cand_score", self.cand_score),
                        ("cand_probs", self.cand_probs), ("cand_delta", self.cand_delta), ("cand_reg_var", self.cand_reg_var),
                        ("cand_anchor", self.cand_anchor), ("cand_run_delta", self.cand_run_delta), ("boxes", self.boxes),
                        ("cov", self.cov), ("keep", self.keep), ("n_keep", self.n_keep), ("nms_scratch", self.nms_scratch),
                        ("m_boxes", self.m_boxes), ("m_cov", self.m_cov), ("m_scores", self.m_scores),
                        ("m_classes", self.m_classes), ("m_probs", self.m_probs)):
            setattr(ws, name, hip.ptr(t))
        ws.n_capacity = self.n_cap
        self.ws = ws
        self._draws = 0          # native-RNG calls served so far (default Philox key stream, see _begin_draw)
        self._dirty = False      # an enqueue failed half way: counters / bitmap may be non-zero, reset before the next image

    def _merged_planes(self) -> dict:
        """Allocates (once) what only the two-launch merge / the eps-replay path touches."""
        if self._planes is None:
            K, D, dev = self.p.num_classes, self.cov_dims, self.device
            f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
            merged = self.n_runs > 1
            pl = {"mean_cls": f32(self.R * K) if merged else None,
                  "mean_cls_var": f32(self.R * K) if merged and self.has_cls_var else None,
                  "mean_delta": f32(self.R * 4) if merged and self.dense_box_merge else None,
                  "mean_reg_var": f32(self.R * D) if merged and D > 0 and self.dense_box_merge else None, "maybe_bits": None}
            if self.has_cls_var:
                n_words = sum(self.p.num_anchors * K * ((h * w + 63) // 64) for h, w in self.shapes)   # == pod_maybe_words(): a word per (plane, 64 cells)
                pl["maybe_bits"] = torch.zeros(n_words, dtype=torch.int64, device=dev)
            for name, t in pl.items():
                setattr(self.ws, name, hip.ptr(t))
            self._planes = pl
        return self._planes

    mean_cls = property(lambda self: self._merged_planes()["mean_cls"])
    mean_cls_var = property(lambda self: self._merged_planes()["mean_cls_var"])
    mean_delta = property(lambda self: self._merged_planes()["mean_delta"])
    mean_reg_var = property(lambda self: self._merged_planes()["mean_reg_var"])
    maybe_bits = property(lambda self: self._merged_planes()["maybe_bits"])

    # ------------------------------------------------------------------------------------------
    def set_anchors(self, anchors: Sequence[torch.Tensor]) -> None:
        """Replaces the workspace's copy of the anchors (same geometry) in place, on the current stream."""
        new = torch.cat([a.to(self.anchors.device, torch.float32) for a in anchors])
        assert new.shape == self.anchors.shape, (new.shape, self.anchors.shape)
        self.anchors.copy_(new)

    def _make_cfg(self) -> hip.PodConfig:
        p = self.p
        c = hip.PodConfig()
        c.n_levels, c.n_runs, c.num_anchors, c.num_classes = self.L, self.n_runs, p.num_anchors, p.num_classes
        c.cov_dims, c.has_cls_var, c.merge_quirk = self.cov_dims, int(self.has_cls_var), int(p.merge_quirk)
        c.cls_samples, c.prop_samples, c.topk, c.max_detections = p.cls_var_num_samples, p.prop_num_samples, p.topk_candidates, p.max_detections
        c.score_thresh, c.nms_thresh, c.affinity_thresh = p.score_thresh, p.nms_thresh, p.affinity_thresh
        for i in range(4):
            c.box_weights[i] = p.box_weights[i]
        c.philox_seed = p.philox_seed
        return c

    def _begin_draw(self, draw_id: Optional[int]) -> None:
        """Philox key of the next image's in-kernel draws: (seed, draw id).  The reference draws FRESH normals on every
        call (PI:291-294, PI:351-356); with a constant key every image -- and every member of a post-NMS ensemble --
        would see the same eps at the same (level, anchor, class, sample).  draw_id=None takes the next value of this
        workspace's own counter; pass an explicit id (e.g. the image id) for reproducible draws.  K1, K1b and K2b/K3 of
        one image read the key from the same PodConfig, so they still re-derive identical draws."""
        if draw_id is None:
            draw_id = self._draws
            self._draws += 1
        seed = int(self.p.philox_seed)
        lo = (seed ^ (seed >> 32)) & 0xFFFFFFFF
        self.cfg.philox_seed = lo | ((int(draw_id) & 0xFFFFFFFF) << 32)

    def _clean(self) -> None:
        """Workspace invariants (cand_count / tickets / maybe_bits all zero between images) are restored by the kernels
        themselves (K2 consumes the counters, K1b the bitmap).  If an enqueue raised half way they are re-established
        here, before the next image appends at a stale count."""
        if self._dirty:
            hip.check(self.lib.pod_reset_counters(hip.ptr(self.counters), int(self.counters.numel()), hip.current_stream()),
                      "pod_reset_counters")
            if self._planes is not None and self._planes["maybe_bits"] is not None:
                self._planes["maybe_bits"].zero_()
            self._dirty = False

    def _run_strided(self, name, l, t, c):
        """A level tensor (n_runs, A*c, H, W): every run a contiguous NCHW slab; the run stride is free (batched MC
        runs: A*c*H*W; ensemble members gathered into packed per-member buffers: the packed size)."""
        h, w = self.shapes[l]
        shape = (self.n_runs, self.p.num_anchors * c, h, w)
        ok = t.dtype == torch.float32 and tuple(t.shape) == shape and tuple(t.stride()[1:]) == (h * w, w, 1) and t.device == self.device
        if not ok:
            raise hip.PodError("{}[{}]: expected fp32 {} with contiguous runs on {}, got {} {} strides {} on {}".format(
                name, l, shape, self.device, t.dtype, tuple(t.shape), tuple(t.stride()), t.device))
        return t.data_ptr(), (t.stride(0) if self.n_runs > 1 else shape[1] * h * w)

    def _levels(self, cls, delta, cls_var, reg_var, eps_cls, cls_only: bool = False):
        """cls_only: the SELECT part of the path alone (pod_run_image_part, parts = 1): delta / reg_var do not exist yet."""
        A, K, D = self.p.num_anchors, self.p.num_classes, self.cov_dims
        arr = self._levels_t()
        for l, (h, w) in enumerate(self.shapes):
            lv = arr[l]
            lv.cls, lv.run_stride_cls = self._run_strided("cls", l, cls[l], K)
            lv.delta, lv.run_stride_delta = (None, 0) if cls_only else self._run_strided("delta", l, delta[l], 4)
            lv.cls_var = lv.reg_var = lv.eps_cls = None
            lv.run_stride_reg = 0
            if self.has_cls_var:
                lv.cls_var, rs = self._run_strided("cls_var", l, cls_var[l], K)
                if rs != lv.run_stride_cls:
                    raise hip.PodError("cls_var[{}] must share the run stride of cls".format(l))
            if D > 0 and not cls_only:
                lv.reg_var, lv.run_stride_reg = self._run_strided("reg_var", l, reg_var[l], D)
            if eps_cls is not None:
                e = eps_cls[l]
                assert tuple(e.shape) == (self.p.cls_var_num_samples, h * w * A, K) and e.is_contiguous() and e.device == self.device
                lv.eps_cls = e.data_ptr()
            lv.H, lv.W, lv.anchor_base = h, w, self.anchor_base[l]
        self._keep_inputs = (cls, delta, cls_var, reg_var)
        return arr

    # ------------------------------------------------------------------------------------------
    def candidates(self, cls, delta, cls_var=None, reg_var=None, eps_cls=None, write_merged: bool = True,
                   draw_id: Optional[int] = None, fused: bool = False):
        """K1 + K2 + K2b: dense tensors -> level-concatenated candidate arrays (device-resident).
        fused (native draws only): merge + score by pod_merge_score_fused, the one launch pod_run_image uses, instead of
        pod_mc_merge_score + pod_score_maybe."""
        self._clean()
        self._begin_draw(draw_id)
        self._dirty = True
        lv = self._candidates(cls, delta, cls_var, reg_var, eps_cls, write_merged, fused)
        self._dirty = False
        return lv

    def _candidates(self, cls, delta, cls_var, reg_var, eps_cls, write_merged, fused=False):
        lib, cfg, st = self.lib, self.cfg, hip.current_stream()
        lv = self._levels(cls, delta, cls_var, reg_var, eps_cls)
        self._lv_keepalive = (lv, eps_cls)
        P = hip.ptr
        # cand_count is zero here: allocated zeroed, and the gather kernel consumes (re-zeroes) it every image
        # prune mode: native RNG with a variance head -> dense pass flags, K1b samples (see k1_mc_merge_score.hip)
        prune = self.has_cls_var and eps_cls is None
        wm = (write_merged or prune) and self.n_runs > 1
        if fused:
            assert eps_cls is None, "pod_merge_score_fused draws its own normals"
            hip.check(lib.pod_merge_score_fused(cfg, lv, P(self.mean_cls) if write_merged and self.n_runs > 1 else None,
                                                P(self.mean_cls_var) if write_merged and self.n_runs > 1 and self.has_cls_var else None,
                                                P(self.cand_keys), P(self.cand_count), P(self.probs_dense) if prune else None, st),
                      "pod_merge_score_fused")
        else:
            self._merge_score_two_launches(lib, cfg, lv, wm, prune, st)
        hip.check(lib.pod_level_topk(cfg, lv, P(self.cand_keys), P(self.cand_count), P(self.sel_keys), P(self.sel_count),
                                     P(self.cat_keys), P(self.cat_level), P(self.n_total), st), "pod_level_topk")
        hip.check(lib.pod_gather_candidates(cfg, lv, P(self.anchors), P(self.cat_keys), P(self.cat_level), P(self.n_total),
                                            P(self.cand_count), P(self.probs_dense) if prune else None,
                                            P(self.cand_anchor_idx), P(self.cand_level), P(self.cand_score), P(self.cand_class),
                                            P(self.cand_probs), P(self.cand_delta), P(self.cand_reg_var) if self.cov_dims else None,
                                            P(self.cand_anchor), P(self.cand_run_delta), st),
                  "pod_gather_candidates")
        return lv

    def _merge_score_two_launches(self, lib, cfg, lv, wm, prune, st):
        P = hip.ptr
        hip.check(lib.pod_mc_merge_score(cfg, lv, P(self.mean_cls) if wm else None, P(self.mean_cls_var) if wm else None,
                                         P(self.mean_delta) if wm and self.dense_box_merge else None,
                                         P(self.mean_reg_var) if wm and self.dense_box_merge and self.cov_dims else None,
                                         P(self.cand_keys), P(self.cand_count), P(self.maybe_bits) if prune else None, st),
                  "pod_mc_merge_score")
        if prune:
            hip.check(lib.pod_score_maybe(cfg, lv, P(self.mean_cls), P(self.mean_cls_var), P(self.maybe_bits),
                                          P(self.cand_keys), P(self.cand_count), P(self.probs_dense), st), "pod_score_maybe")

    # -- test support: the native-RNG draws of one draw id, in the reference's tensor layouts ----------------------
    def dump_cls_normals(self, draw_id: int) -> List[torch.Tensor]:
        """Per level, the (cls_samples, H*W*A, K) normals K1b / K2b use for `draw_id` (PI:291-294's rsample)."""
        self._begin_draw(int(draw_id))
        A, K, S = self.p.num_anchors, self.p.num_classes, self.p.cls_var_num_samples
        lv = self._levels_t()
        for l, (h, w) in enumerate(self.shapes):
            lv[l].H, lv[l].W, lv[l].anchor_base = h, w, self.anchor_base[l]
        out = []
        for l, (h, w) in enumerate(self.shapes):
            t = torch.empty((S, h * w * A, K), dtype=torch.float32, device=self.device)
            hip.check(self.lib.pod_dump_cls_normals(self.cfg, lv, l, hip.ptr(t), hip.current_stream()), "pod_dump_cls_normals")
            out.append(t)
        return out

    def dump_box_normals(self, draw_id: int, global_anchor_ids: torch.Tensor) -> torch.Tensor:
        """(prop_samples, n, 4) normals K3 uses for `draw_id` at the given anchors (PI:351-356's rsample);
        global id = anchor_base[level] + index inside the level."""
        self._begin_draw(int(draw_id))
        g = global_anchor_ids.to(self.device, torch.int32).contiguous()
        t = torch.empty((self.p.prop_num_samples, int(g.numel()), 4), dtype=torch.float32, device=self.device)
        hip.check(self.lib.pod_dump_box_normals(self.cfg, hip.ptr(g), int(g.numel()), hip.ptr(t), hip.current_stream()), "pod_dump_box_normals")
        return t

    def decode(self, lv, eps_prop: Optional[torch.Tensor] = None):
        """K3: candidate boxes + covariances."""
        P = hip.ptr
        n_replay = 0
        if eps_prop is not None:
            assert eps_prop.dim() == 3 and eps_prop.shape[0] == self.p.prop_num_samples and eps_prop.shape[2] == 4
            assert eps_prop.is_contiguous() and eps_prop.device == self.device
            n_replay = int(eps_prop.shape[1])
            self._eps_keepalive = eps_prop
        hip.check(self.lib.pod_decode_cov(self.cfg, lv, P(self.n_total), self.n_cap, P(self.cand_delta),
                                          P(self.cand_reg_var) if self.cov_dims else None, P(self.cand_anchor),
                                          P(self.cand_run_delta), P(self.cand_anchor_idx), P(self.cand_level),
                                          P(eps_prop), n_replay, P(self.boxes), P(self.cov), hip.current_stream()),
                  "pod_decode_cov")

    @property
    def has_covariance(self) -> bool:
        """False reproduces the reference's `all_predicted_boxes_covariance = []` (PI:381)."""
        return self.cov_dims > 0 or self.n_runs > 1

    def nms(self):
        """K4 on the candidate list: keep[:max_detections] (PI:554-560, IU:31-36, IU:83-89)."""
        P = hip.ptr
        hip.check(self.lib.pod_nms_cluster(self.cfg, P(self.n_total), self.n_cap, P(self.boxes), P(self.cand_score),
                                           P(self.cand_class), P(self.keep), P(self.n_keep), P(self.nms_scratch),
                                           hip.current_stream()), "pod_nms_cluster")

    def new_detections(self, out_size) -> DeviceDetections:
        return DeviceDetections((int(out_size[0]), int(out_size[1])), self.p.num_classes, hip.POD_MAX_DETECTIONS, self.device)

    def finalize(self, keep, n_rows, boxes, cov, scores, classes, probs, image_size, out_size) -> DeviceDetections:
        """K7: gather through `keep` (or identity), rescale to the output resolution, records (IU:42-53, :374-425, :428-502)."""
        P = hip.ptr
        out = self.new_detections(out_size)
        sx, sy = out_size[1] / image_size[1], out_size[0] / image_size[0]   # IU:394-396
        hip.check(self.lib.pod_finalize(self.cfg, P(keep), P(n_rows), P(boxes), P(cov), P(scores), P(classes), P(probs), sx, sy,
                                        float(out_size[0]), float(out_size[1]), out.ptr("boxes"), out.ptr("cov"), out.ptr("scores"),
                                        out.ptr("classes"), out.ptr("probs"), out.ptr("records"), out.ptr("n_det"),
                                        hip.current_stream()),
                  "pod_finalize")
        return out

    def postprocess(self, mode: str, image_size, out_size, box_merge_mode: str = "bayesian_inference",
                    cls_merge_mode: str = "max_score") -> DeviceDetections:
        """K4 (+K5/K6) + K7."""
        if mode not in MODES:
            raise ValueError("Invalid inference mode {}.".format(mode))   # PI:100-103
        lib, cfg, st, P = self.lib, self.cfg, hip.current_stream(), hip.ptr
        K, md = self.p.num_classes, hip.POD_MAX_DETECTIONS
        dev = self.device
        self.nms()
        cov_in = self.cov if self.has_covariance else None
        if mode == "bayes_od":
            if cov_in is None:
                raise hip.PodError("bayes_od needs box covariances (a reg_var head or MC runs)")
            bm = {"bayesian_inference": 0, "covariance_intersection": 1}[box_merge_mode]
            cm = {"max_score": 0, "bayesian_inference": 1}[cls_merge_mode]
            hip.check(lib.pod_bayes_fuse(cfg, P(self.n_total), P(self.keep), P(self.n_keep), P(self.boxes), P(self.cov),
                                         P(self.cand_score), P(self.cand_class), P(self.cand_probs), bm, cm, P(self.m_boxes),
                                         P(self.m_cov), P(self.m_scores), P(self.m_classes), P(self.m_probs), st), "pod_bayes_fuse")
            src = (None, self.m_boxes, self.m_cov, self.m_scores, self.m_classes, self.m_probs)
        elif mode == "anchor_statistics":
            hip.check(lib.pod_anchor_stats_merge(cfg, P(self.n_total), P(self.keep), P(self.n_keep), P(self.boxes), P(cov_in),
                                                 P(self.cand_class), P(self.cand_probs), P(self.m_boxes), P(self.m_cov),
                                                 P(self.m_scores), P(self.m_classes), P(self.m_probs), st), "pod_anchor_stats_merge")
            src = (None, self.m_boxes, self.m_cov, self.m_scores, self.m_classes, self.m_probs)
        else:   # standard NMS (also the pre-NMS MC-dropout / ensemble modes): gather through keep
            src = (self.keep, self.boxes, cov_in, self.cand_score, self.cand_class, self.cand_probs)
        keep, b, c, s, cl, pr = src
        return self.finalize(keep, self.n_keep, b, c, s, cl, pr, image_size, out_size)

    # ------------------------------------------------------------------------------------------
    _MODE_ID = {"standard_nms": hip.POD_MODE_STANDARD_NMS, "mc_dropout_ensembles": hip.POD_MODE_STANDARD_NMS,
                "ensembles": hip.POD_MODE_STANDARD_NMS, "bayes_od": hip.POD_MODE_BAYES_OD,
                "anchor_statistics": hip.POD_MODE_ANCHOR_STATISTICS}

    def run_image(self, mode: str, cls, delta, cls_var, reg_var, image_size, out_size,
                  box_merge_mode: str = "bayesian_inference", cls_merge_mode: str = "max_score",
                  draw_id: Optional[int] = None) -> DeviceDetections:
        """pod_run_image: K1 .. K7 of one image enqueued by one C call (native Philox draws)."""
        if mode not in MODES:
            raise ValueError("Invalid inference mode {}.".format(mode))   # PI:100-103
        if mode == "bayes_od" and not self.has_covariance:
            raise hip.PodError("bayes_od needs box covariances (a reg_var head or MC runs)")
        self._clean()
        self._begin_draw(draw_id)
        lv = self._levels(cls, delta, cls_var, reg_var, None)
        self._lv_keepalive = (lv, None)
        out = self.new_detections(out_size)
        d = hip.PodDetections(out.ptr("boxes"), out.ptr("cov"), out.ptr("scores"), out.ptr("classes"), out.ptr("probs"),
                              out.ptr("records"), out.ptr("n_det"))
        bm = {"bayesian_inference": 0, "covariance_intersection": 1}[box_merge_mode]
        cm = {"max_score": 0, "bayesian_inference": 1}[cls_merge_mode]
        self._dirty = True       # a failure between K1 and K2 leaves counters / bitmap non-zero
        hip.check(self.lib.pod_run_image(self.cfg, lv, self.ws, self._MODE_ID[mode], bm, cm, int(image_size[0]), int(image_size[1]),
                                         int(out_size[0]), int(out_size[1]), d, hip.current_stream()), "pod_run_image")
        self._dirty = False
        return out

    # -- the path in two parts, for the sparse bbox tower between them (pod_compare_amd/sparse.py) -------------------------------------
    def select(self, cls, cls_var=None, draw_id: Optional[int] = None) -> None:
        """PI:211-308 on the class side alone (pod_run_image_part, parts = 1): merge + score + per-level top-k.  Afterwards the image's
        candidates stand in cat_keys / cat_level / n_total on the device; box_delta / box_reg_var have not been looked at."""
        self._clean()
        self._begin_draw(draw_id)
        lv = self._levels(cls, None, cls_var, None, None, cls_only=True)
        self._lv_keepalive = (lv, None)
        self._dirty = True
        hip.check(self.lib.pod_run_image_part(self.cfg, lv, self.ws, hip.POD_MODE_STANDARD_NMS, 0, 0, 1, 1, 1, 1, None, 1, hip.current_stream()),
                  "pod_run_image_part(select)")

    def finish(self, mode: str, cls, delta, cls_var, reg_var, image_size, out_size, box_merge_mode: str = "bayesian_inference",
               cls_merge_mode: str = "max_score") -> DeviceDetections:
        """PI:310-636 for the candidates `select` left (parts = 2): gather + decode + NMS / fusion + finalize.  delta / reg_var need to be
        defined at the candidates' cells only."""
        if mode not in MODES:
            raise ValueError("Invalid inference mode {}.".format(mode))   # PI:100-103
        if mode == "bayes_od" and not self.has_covariance:
            raise hip.PodError("bayes_od needs box covariances (a reg_var head or MC runs)")
        lv = self._levels(cls, delta, cls_var, reg_var, None)
        self._lv_keepalive = (lv, None)
        out = self.new_detections(out_size)
        d = hip.PodDetections(out.ptr("boxes"), out.ptr("cov"), out.ptr("scores"), out.ptr("classes"), out.ptr("probs"),
                              out.ptr("records"), out.ptr("n_det"))
        bm = {"bayesian_inference": 0, "covariance_intersection": 1}[box_merge_mode]
        cm = {"max_score": 0, "bayesian_inference": 1}[cls_merge_mode]
        hip.check(self.lib.pod_run_image_part(self.cfg, lv, self.ws, self._MODE_ID[mode], bm, cm, int(image_size[0]), int(image_size[1]),
                                              int(out_size[0]), int(out_size[1]), d, 2, hip.current_stream()), "pod_run_image_part(finish)")
        self._dirty = False
        return out

    def run(self, mode: str, cls, delta, cls_var=None, reg_var=None, *, image_size, out_size,
            eps_fn: Optional[Callable] = None, box_merge_mode: str = "bayesian_inference",
            cls_merge_mode: str = "max_score", write_merged: bool = True, one_call: bool = True,
            draw_id: Optional[int] = None) -> DeviceDetections:
        """predictor(input_im) minus the conv net: dense head tensors -> detections.

        eps_fn=None  : native mode, in-kernel Philox4x32-10, fully asynchronous; with one_call the whole launch
          sequence is enqueued by ONE C call (pod_run_image) instead of one ctypes call per kernel.
        eps_fn=callable(shape)->CPU tensor : eps-replay parity mode; draws are requested in the
          reference's order (one (S_cls, R_l, K) tensor per level, then one (1000, n, 4)).
        draw_id (native mode): Philox key of this call's draws, see _begin_draw; None = fresh draws on every call."""
        if eps_fn is None and one_call and write_merged:
            return self.run_image(mode, cls, delta, cls_var, reg_var, image_size, out_size, box_merge_mode, cls_merge_mode, draw_id)
        eps_cls = eps_prop = None
        if eps_fn is not None and self.has_cls_var:
            A, K = self.p.num_anchors, self.p.num_classes
            eps_cls = [eps_fn((self.p.cls_var_num_samples, h * w * A, K)).to(self.device).contiguous() for h, w in self.shapes]
        lv = self.candidates(cls, delta, cls_var, reg_var, eps_cls, write_merged, draw_id)
        if eps_fn is not None and self.cov_dims > 0:
            n = int(self.n_total.item())
            if n > 0:
                eps_prop = eps_fn((self.p.prop_num_samples, n, 4)).to(self.device).contiguous()
        self.decode(lv, eps_prop)
        return self.postprocess(mode, image_size, out_size, box_merge_mode, cls_merge_mode)


class PostNmsEnsemble:
    """Post-NMS merge of ensemble members / MC-dropout runs: PI:444-481, PI:506-534 ->
    general_black_box_ensembles_post_processing IU:165-289.  Every member goes through the single-run path
    (K1..K4 with N = 1) and its kept rows are appended on the device; the cluster sweep, the per-cluster moments and
    the second NMS then run once."""

    def __init__(self, hp: HotPath, n_members: int):
        assert hp.n_runs == 1, "members are processed one run at a time"
        self.hp, self.n_members = hp, int(n_members)
        self.cap = self.n_members * hp.p.max_detections
        if self.cap > hip.POD_MAX_CANDIDATES:
            raise hip.PodError("too many member detections for the post-NMS merge: {}".format(self.cap))
        dev, K, cap = hp.device, hp.p.num_classes, self.cap
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        self.m_boxes, self.m_cov, self.m_classes, self.m_probs = f32(cap, 4), f32(cap, 4, 4), i32(cap), f32(cap, K)
        self.total = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seeds, self.n_seeds = i32(cap), i32(1)
        self.c_boxes, self.c_cov, self.c_scores, self.c_classes, self.c_probs = f32(cap, 4), f32(cap, 4, 4), f32(cap), i32(cap), f32(cap, K)
        self.keep, self.n_keep = i32(hip.POD_MAX_DETECTIONS), i32(1)
        self.scratch = torch.zeros(hp.lib.pod_nms_scratch_bytes(cap), dtype=torch.uint8, device=dev)

    def run(self, members, *, image_size, out_size, eps_fn: Optional[Callable] = None,
            draw_id: Optional[int] = None) -> DeviceDetections:
        """members: iterable of (cls, delta, cls_var, reg_var) per-level tensor lists with N = 1.
        draw_id (native mode): member j draws with Philox key (seed, draw_id * n_members + j): independent normals per
        member, as the reference's per-call sampling (PI:291-294, 351-356); None = the workspace's running counter."""
        hp, P, st = self.hp, hip.ptr, hip.current_stream()
        self.total.zero_()
        for j, (cls, delta, cls_var, reg_var) in enumerate(members):
            eps_cls = eps_prop = None
            if eps_fn is not None and hp.has_cls_var:
                A, K = hp.p.num_anchors, hp.p.num_classes
                eps_cls = [eps_fn((hp.p.cls_var_num_samples, h * w * A, K)).to(hp.device).contiguous() for h, w in hp.shapes]
            lv = hp.candidates(cls, delta, cls_var, reg_var, eps_cls,
                               draw_id=None if draw_id is None else int(draw_id) * self.n_members + j)
            if eps_fn is not None and hp.cov_dims > 0:
                n = int(hp.n_total.item())
                if n > 0:
                    eps_prop = eps_fn((hp.p.prop_num_samples, n, 4)).to(hp.device).contiguous()
            hp.decode(lv, eps_prop)
            hp.nms()
            hip.check(hp.lib.pod_ensemble_append(hp.cfg, P(hp.keep), P(hp.n_keep), P(hp.boxes), P(hp.cov) if hp.has_covariance else None,
                                                 P(hp.cand_class), P(hp.cand_probs), self.cap, P(self.m_boxes), P(self.m_cov),
                                                 P(self.m_classes), P(self.m_probs), P(self.total), st), "pod_ensemble_append")
        hip.check(hp.lib.pod_ensemble_merge(hp.cfg, P(self.total), self.cap, P(self.m_boxes), P(self.m_cov), P(self.m_classes),
                                            P(self.m_probs), P(self.seeds), P(self.n_seeds), P(self.c_boxes), P(self.c_cov),
                                            P(self.c_scores), P(self.c_classes), P(self.c_probs), st), "pod_ensemble_merge")
        hip.check(hp.lib.pod_nms_cluster(hp.cfg, P(self.n_seeds), self.cap, P(self.c_boxes), P(self.c_scores), P(self.c_classes),
                                         P(self.keep), P(self.n_keep), P(self.scratch), st), "pod_nms_cluster")      # IU:269-274
        return hp.finalize(self.keep, self.n_keep, self.c_boxes, self.c_cov, self.c_scores, self.c_classes, self.c_probs,
                           image_size, out_size)
