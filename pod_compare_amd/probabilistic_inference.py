"""The plugin surface of the reference, on the MI355X hot path.

Mirrors /root/reference/src/probabilistic_inference/probabilistic_inference.py (PI):

    build_predictor(cfg)                       PI:20-33
    ProbabilisticPredictor.__call__(input_im)  PI:86-111   mode dispatch + probabilistic_detector_postprocess
    RetinaNetProbabilisticPredictor
        .retinanet_probabilistic_inference     PI:178-388  -> K1, K1b, K2, K2b, K3
        .post_processing_standard_nms          PI:390-407  -> K4, K7
        .post_processing_anchor_statistics     PI:409-428  -> K4, K6, K7
        .post_processing_mc_dropout_ensembles  PI:430-481  (pre_nms -> standard NMS on merged runs)
        .post_processing_ensembles             PI:483-534  (pre_nms: stack member outputs -> K1 with N = members)
        .post_processing_bayes_od              PI:536-636  -> K4, K5, K7

`input_im` is the reference's: a list with ONE dict {'image': (3,H,W) BGR tensor, 'height', 'width',
'image_id'} (batch 1, apply_net.py:35).  The return value is an `Instances` with the reference's
fields.  All arithmetic after the conv net runs in the HIP library (pod_compare_amd.hip); there is no
CPU fallback.  Differences from the reference, all documented in DESIGN.md:
  * MC dropout is a flag on the model, not `model.train()`;
  * zero candidates give an empty `Instances` for every model type (the reference raises for reg-var
    models); a degenerate BayesOD cluster falls back to its centre;
  * post-NMS ensemble merges (PI:444-481, PI:506-534) run every member through the N = 1 path, then the HIP
    restatement of general_black_box_ensembles_post_processing (IU:165-289).
"""
import os
from abc import ABC, abstractmethod
from typing import Callable, List, Optional

import torch

from . import checkpoint, hotpath, modeling
from .structures import Boxes, Instances
from .synthetic import HeadOutputs


def build_model(cfg, save_dir: Optional[str] = "", load_weights: bool = True) -> modeling.ProbabilisticRetinaNet:
    """detectron2 `build_model(cfg)` for META_ARCHITECTURE == ProbabilisticRetinaNet (PR:25-65), followed by the
    reference's `DetectionCheckpointer(model, save_dir).resume_or_load(cfg.MODEL.WEIGHTS, resume=True)` (PI:59-84):
    `<save_dir>/last_checkpoint` wins over cfg.MODEL.WEIGHTS, an empty path keeps the random initialisation, a path
    that cannot be read raises (checkpoint.CheckpointError) -- never a silent random-init run.  save_dir "" means
    cfg.OUTPUT_DIR, None means "no directory" (only cfg.MODEL.WEIGHTS is considered).  FrozenBN is folded into the conv
    weights AFTER loading."""
    pm = cfg.MODEL.PROBABILISTIC_MODELING
    model = modeling.ProbabilisticRetinaNet(
        num_classes=cfg.MODEL.RETINANET.NUM_CLASSES, dropout_rate=pm.DROPOUT_RATE, cls_var_loss=pm.CLS_VAR_LOSS.NAME,
        cls_var_num_samples=pm.CLS_VAR_LOSS.NUM_SAMPLES, bbox_cov_loss=pm.BBOX_COV_LOSS.NAME,
        bbox_cov_type=pm.BBOX_COV_LOSS.COVARIANCE_TYPE, test_score_thresh=cfg.MODEL.RETINANET.SCORE_THRESH_TEST,
        test_topk_candidates=cfg.MODEL.RETINANET.TOPK_CANDIDATES_TEST, test_nms_thresh=cfg.MODEL.RETINANET.NMS_THRESH_TEST,
        max_detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE, min_size_test=cfg.INPUT.MIN_SIZE_TEST,
        max_size_test=cfg.INPUT.MAX_SIZE_TEST)
    model.loaded_from = ""
    if load_weights:
        if save_dir == "":
            save_dir = cfg.get("OUTPUT_DIR", None)
        model.loaded_from = checkpoint.load_model_weights(model, save_dir, cfg.MODEL.get("WEIGHTS", ""))
    model = model.to(torch.device(cfg.MODEL.DEVICE)).eval()
    modeling.fold_frozen_bn(model)      # inference-only algebra: the same affine map, one biased conv per (conv, FrozenBN) pair
    return model


def model_test_attributes(cfg):
    """The model attributes the predictor reads (test_topk_candidates PI:300, test_score_thresh PI:304,
    test_nms_thresh, max_detections_per_image PI:407, cls_var_num_samples PI:294) without building a model: what a
    merge-only rank of the one-seed-per-GPU ensemble needs."""
    from types import SimpleNamespace
    return SimpleNamespace(test_topk_candidates=cfg.MODEL.RETINANET.TOPK_CANDIDATES_TEST,
                           test_score_thresh=cfg.MODEL.RETINANET.SCORE_THRESH_TEST,
                           test_nms_thresh=cfg.MODEL.RETINANET.NMS_THRESH_TEST,
                           max_detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                           cls_var_num_samples=cfg.MODEL.PROBABILISTIC_MODELING.CLS_VAR_LOSS.NUM_SAMPLES)


def ensemble_member_dir(cfg, random_seed) -> str:
    """PI:66-71: `<parent of OUTPUT_DIR>/random_seed_<s>`."""
    return os.path.join(os.path.split(cfg.OUTPUT_DIR)[0], "random_seed_" + str(random_seed))


def build_predictor(cfg, model=None, model_list=None):
    """PI:20-33.  `model` / `model_list` let a caller inject built (or fake) models, e.g. ensemble members
    whose weights were loaded elsewhere (PI:59-77 loads them from sibling `random_seed_<s>` directories)."""
    if cfg.MODEL.META_ARCHITECTURE == "ProbabilisticRetinaNet":
        return RetinaNetProbabilisticPredictor(cfg, model=model, model_list=model_list)
    raise ValueError("Invalid meta-architecture {}.".format(cfg.MODEL.META_ARCHITECTURE))


class ProbabilisticPredictor(ABC):
    """PI:36-166."""

    def __init__(self, cfg, model=None, model_list=None):
        self.cfg = cfg.clone()
        pi = self.cfg.PROBABILISTIC_INFERENCE
        self.inference_mode = pi.INFERENCE_MODE
        # PI:58-84: in ensembles mode only the members are loaded, self.model stays as built
        self.model = model if model is not None else build_model(self.cfg, load_weights=self.inference_mode != "ensembles")
        self.model_list = list(model_list) if model_list is not None else []
        self.mc_dropout_enabled = pi.MC_DROPOUT.ENABLE
        self.num_mc_dropout_runs = pi.MC_DROPOUT.NUM_RUNS
        if self.inference_mode == "ensembles" and not self.model_list:
            # PI:59-77: one model per RANDOM_SEED_NUMS entry, each loaded from its sibling `random_seed_<s>` directory.
            # A member without any checkpoint (no last_checkpoint there, empty MODEL.WEIGHTS) is a random-init model
            # seeded with its number (synthetic runs).
            state = torch.random.get_rng_state()
            for seed in pi.ENSEMBLES.RANDOM_SEED_NUMS:
                torch.manual_seed(int(seed))
                self.model_list.append(build_model(self.cfg, save_dir=ensemble_member_dir(self.cfg, seed)))
            torch.random.set_rng_state(state)

    def __call__(self, input_im):
        if self.inference_mode == "standard_nms":
            return self.post_processing_standard_nms(input_im)
        elif self.inference_mode == "mc_dropout_ensembles":
            return self.post_processing_mc_dropout_ensembles(input_im)
        elif self.inference_mode == "anchor_statistics":
            return self.post_processing_anchor_statistics(input_im)
        elif self.inference_mode == "ensembles":
            return self.post_processing_ensembles(input_im, self.model_list)
        elif self.inference_mode == "bayes_od":
            return self.post_processing_bayes_od(input_im)
        raise ValueError("Invalid inference mode {}.".format(self.inference_mode))

    @abstractmethod
    def post_processing_standard_nms(self, input_im):
        pass

    @abstractmethod
    def post_processing_anchor_statistics(self, input_im):
        pass

    @abstractmethod
    def post_processing_mc_dropout_ensembles(self, input_im):
        pass

    @abstractmethod
    def post_processing_ensembles(self, input_im, model_list):
        pass

    @abstractmethod
    def post_processing_bayes_od(self, input_im):
        pass


class RetinaNetProbabilisticPredictor(ProbabilisticPredictor):
    """PI:169-636 on the HIP hot path."""

    def __init__(self, cfg, model=None, model_list=None, eps_fn: Optional[Callable] = None, merge_quirk: bool = True):
        super().__init__(cfg, model=model, model_list=model_list)
        self.eps_fn = eps_fn            # None = in-kernel Philox; callable(shape) = eps-replay parity mode
        self.merge_quirk = merge_quirk  # PI:216-222 behaviour; False = true mean over runs
        self.box_reg_weights = tuple(self.cfg.MODEL.RPN.BBOX_REG_WEIGHTS)   # PI:175-176 reads the RPN key
        self._paths = {}
        self.last_path: Optional[hotpath.HotPath] = None
        self.last_detections: Optional[hotpath.DeviceDetections] = None   # device-resident records of the last image
        # True: predictor(input_im) returns the fixed-capacity DeviceDetections (no host sync at all: the count stays on
        # the GPU) instead of an `Instances`; the image-sharded driver uses it so the host can run ahead of the device
        self.return_device = False
        # Sparse bbox tower (round 5, pod_compare_amd/sparse.py): evaluate the cls side of the head first, select the candidates
        # (PI:283-308), then run bbox_subnet / bbox_pred / bbox_cov only over the blocks that can reach a candidate (PI:310-331 reads
        # nothing else).  Own model, native draws, single-model modes; POD_SPARSE_BBOX=1 switches it on for a process.
        self.sparse_bbox_tower = os.environ.get("POD_SPARSE_BBOX", "0") == "1"

    # -- helpers -----------------------------------------------------------------------------------
    def _path_for(self, ho: HeadOutputs, cov_dims: Optional[int] = None) -> hotpath.HotPath:
        if cov_dims is None:
            cov_dims = 0 if ho.reg_var is None else ho.reg_var[0].shape[1] // ho.num_anchors
        dev = ho.cls[0].device
        # one workspace per (geometry, stream): a driver may keep several images in flight on different HIP streams
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = (tuple(ho.shapes), ho.num_runs, ho.cls_var is not None, cov_dims, str(dev), stream)
        if key not in self._paths:
            m = self.model
            params = hotpath.PathParams(
                num_classes=ho.num_classes, num_anchors=ho.num_anchors, topk_candidates=m.test_topk_candidates,
                score_thresh=m.test_score_thresh, nms_thresh=m.test_nms_thresh, max_detections=m.max_detections_per_image,
                cls_var_num_samples=m.cls_var_num_samples, affinity_thresh=self.cfg.PROBABILISTIC_INFERENCE.AFFINITY_THRESHOLD,
                merge_quirk=self.merge_quirk, box_weights=self.box_reg_weights, philox_seed=int(self.cfg.get("SEED", 0)) + 0x5EED)
            self._paths[key] = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=ho.num_runs,
                                               has_cls_var=ho.cls_var is not None, cov_dims=cov_dims, device=ho.cls[0].device)
        return self._paths[key]

    def _head_outputs(self, input_im, outputs=None, ensemble_inference=False, outputs_list=None, need_all_runs=False) -> HeadOutputs:
        """PI:199-273: which raw outputs feed the path (MC-dropout runs, ensemble members or a single pass)."""
        if outputs is not None:
            return outputs
        if ensemble_inference:
            return stack_members(outputs_list)
        image = input_im[0]["image"]
        own = isinstance(self.model, modeling.ProbabilisticRetinaNet)
        if self.mc_dropout_enabled and self.num_mc_dropout_runs > 1:
            if own:
                # the quirky merge (PI:216-222) never reads the last run's cls / cls_var / reg_var: do not compute them
                return self.model(image, num_mc_dropout_runs=self.num_mc_dropout_runs, mc_dropout=True,
                                  skip_unused_last_run=self.merge_quirk and not need_all_runs)
            return self.model(image, num_mc_dropout_runs=self.num_mc_dropout_runs)      # PI:203-206
        if own:
            # PI:53-56 puts the model in train() whenever MC_DROPOUT.ENABLE is set: a single pass (NUM_RUNS = 1) then
            # still runs with dropout active
            return self.model(image, mc_dropout=bool(self.mc_dropout_enabled))
        return self.model(image)                                                          # PI:273

    @staticmethod
    def _draw_id(input_im) -> Optional[int]:
        """Philox key of the image's in-kernel draws: its image_id when it is an integer (reproducible per image,
        independent across images), else the workspace's running counter."""
        i = input_im[0].get("image_id", None)
        return int(i) if isinstance(i, int) and not isinstance(i, bool) else None

    def _sizes(self, input_im, ho: HeadOutputs):
        image_size = tuple(input_im[0]["image"].shape[1:])                                 # IU:39-41, PI:604-606
        out = (input_im[0].get("height", image_size[0]), input_im[0].get("width", image_size[1]))   # PI:106-107
        return image_size, out

    def _sparse_ok(self) -> bool:
        m = self.model
        return (self.sparse_bbox_tower and self.eps_fn is None and isinstance(m, modeling.ProbabilisticRetinaNet) and m.head.takes_wino_path()
                and input_is_cuda(m))

    def _run_sparse(self, mode, input_im) -> Instances:
        """`_run` with the model's bbox side evaluated between the two parts of the path (hotpath.select / finish)."""
        from . import sparse
        m, image, state = self.model, input_im[0]["image"], {}
        cov_dims = m.bbox_cov_dims if m.compute_bbox_cov else 0

        def hook(partial: HeadOutputs):
            hp = self._path_for(partial, cov_dims=cov_dims)
            hp.select(partial.cls, partial.cls_var, draw_id=self._draw_id(input_im))
            state["hp"] = hp
            return sparse.LiveBlocks(hp)

        mc = self.mc_dropout_enabled and self.num_mc_dropout_runs > 1
        ho = m(image, num_mc_dropout_runs=self.num_mc_dropout_runs if mc else -1, mc_dropout=bool(self.mc_dropout_enabled),
               skip_unused_last_run=mc and self.merge_quirk, sparse_bbox=hook)
        hp = self.last_path = state["hp"]
        image_size, out = self._sizes(input_im, ho)
        bo = self.cfg.PROBABILISTIC_INFERENCE.BAYES_OD
        det = hp.finish(mode, ho.cls, ho.delta, ho.cls_var, ho.reg_var, image_size, out, bo.BOX_MERGE_MODE, bo.CLS_MERGE_MODE)
        self.last_detections = det
        return det if self.return_device else detections_to_instances(det)

    def _run(self, mode, input_im, ho: HeadOutputs) -> Instances:
        hp = self._path_for(ho)
        self.last_path = hp
        image_size, out = self._sizes(input_im, ho)
        bo = self.cfg.PROBABILISTIC_INFERENCE.BAYES_OD
        if not ho.last_run_valid and not self.merge_quirk:
            raise hotpath.hip.PodError("the last MC run's cls / cls_var / reg_var were not computed (skip_unused_last_run); "
                                       "only the reference's quirky merge (PI:216-222) may consume these outputs")
        det = hp.run(mode, ho.cls, ho.delta, ho.cls_var, ho.reg_var, image_size=image_size, out_size=out, eps_fn=self.eps_fn,
                     box_merge_mode=bo.BOX_MERGE_MODE, cls_merge_mode=bo.CLS_MERGE_MODE, draw_id=self._draw_id(input_im))
        self.last_detections = det
        return det if self.return_device else detections_to_instances(det)

    def _run_sparse_ensemble(self, input_im, models) -> Instances:
        """PI:495-505 (pre-NMS ensembles) with the sparse bbox tower: every member's trunk + cls side first (PI:498-500's forwards, cut in two),
        the candidates selected from the members' merged class outputs, then every member's bbox side over the SAME live blocks."""
        from . import sparse
        image = input_im[0]["image"]
        sts = [m.cls_part(image) for m in models]
        first = models[0]
        cov_dims = first.bbox_cov_dims if first.compute_bbox_cov else 0
        partial = stack_members([m.partial_outputs(st) for m, st in zip(models, sts)])
        hp = self.last_path = self._path_for(partial, cov_dims=cov_dims)
        hp.select(partial.cls, partial.cls_var, draw_id=self._draw_id(input_im))
        live = sparse.LiveBlocks(hp)
        ho = stack_members([m.bbox_part(st, live) for m, st in zip(models, sts)])
        image_size, out = self._sizes(input_im, ho)
        det = hp.finish("standard_nms", ho.cls, ho.delta, ho.cls_var, ho.reg_var, image_size, out)
        self.last_detections = det
        return det if self.return_device else detections_to_instances(det)

    def _run_post_nms(self, input_im, members: List[HeadOutputs]) -> Instances:
        """Per-member standard NMS, then general_black_box_ensembles_post_processing (IU:165-289)."""
        hp = self._path_for(members[0])
        self.last_path = hp
        key = ("post_nms", id(hp), len(members))
        if key not in self._paths:
            self._paths[key] = hotpath.PostNmsEnsemble(hp, len(members))
        image_size, out = self._sizes(input_im, members[0])
        det = self._paths[key].run([(m.cls, m.delta, m.cls_var, m.reg_var) for m in members], image_size=image_size, out_size=out,
                                   eps_fn=self.eps_fn, draw_id=self._draw_id(input_im))
        self.last_detections = det
        return det if self.return_device else detections_to_instances(det)

    # -- reference surface ---------------------------------------------------------------------------
    def retinanet_probabilistic_inference(self, input_im, outputs=None, ensemble_inference=False, outputs_list=None):
        """PI:178-388.  Returns (boxes (n,4), covariances (n,4,4) or [], scores (n,), class ids (n,) int64,
        prob vectors (n,K)) as device tensors; n needs one host sync."""
        ho = self._head_outputs(input_im, outputs, ensemble_inference, outputs_list)
        hp = self._path_for(ho)
        self.last_path = hp
        eps_cls = eps_prop = None
        if self.eps_fn is not None and hp.has_cls_var:
            A, K = hp.p.num_anchors, hp.p.num_classes
            eps_cls = [self.eps_fn((hp.p.cls_var_num_samples, h * w * A, K)).to(hp.device).contiguous() for h, w in hp.shapes]
        lv = hp.candidates(ho.cls, ho.delta, ho.cls_var, ho.reg_var, eps_cls)
        n = int(hp.n_total.item())
        if self.eps_fn is not None and hp.cov_dims > 0 and n > 0:
            eps_prop = self.eps_fn((hp.p.prop_num_samples, n, 4)).to(hp.device).contiguous()
        hp.decode(lv, eps_prop)
        cov = hp.cov[:n] if hp.has_covariance else []
        return hp.boxes[:n], cov, hp.cand_score[:n], hp.cand_class[:n].long(), hp.cand_probs[:n]

    def post_processing_standard_nms(self, input_im):
        if self._sparse_ok():
            return self._run_sparse("standard_nms", input_im)
        return self._run("standard_nms", input_im, self._head_outputs(input_im))

    def post_processing_anchor_statistics(self, input_im):
        if self._sparse_ok():
            return self._run_sparse("anchor_statistics", input_im)
        return self._run("anchor_statistics", input_im, self._head_outputs(input_im))

    def post_processing_mc_dropout_ensembles(self, input_im):
        if self.cfg.PROBABILISTIC_INFERENCE.ENSEMBLES_DROPOUT.BOX_MERGE_MODE == "pre_nms":     # PI:442-443
            if self._sparse_ok():
                return self._run_sparse("standard_nms", input_im)
            return self._run("standard_nms", input_im, self._head_outputs(input_im))
        ho = self._head_outputs(input_im, need_all_runs=True)                                  # PI:445-451: N runs, merged post-NMS
        return self._run_post_nms(input_im, [run_slice(ho, r) for r in range(ho.num_runs)])

    def post_processing_ensembles(self, input_im, model_dict):
        if self.cfg.PROBABILISTIC_INFERENCE.ENSEMBLES.BOX_MERGE_MODE == "pre_nms":             # PI:495-505
            if (self.sparse_bbox_tower and self.eps_fn is None and all(isinstance(m, modeling.ProbabilisticRetinaNet) and m.head.takes_wino_path()
                                                                        and input_is_cuda(m) for m in model_dict)):
                return self._run_sparse_ensemble(input_im, model_dict)
            members = [m(input_im[0]["image"]) for m in model_dict]
            return self._run("standard_nms", input_im, stack_members(members))
        return self._run_post_nms(input_im, [m(input_im[0]["image"]) for m in model_dict])     # PI:506-534

    def post_processing_bayes_od(self, input_im):
        if self._sparse_ok():
            return self._run_sparse("bayes_od", input_im)
        return self._run("bayes_od", input_im, self._head_outputs(input_im))


def input_is_cuda(model) -> bool:
    return model.device.type == "cuda"


def run_slice(ho: HeadOutputs, run: int) -> HeadOutputs:
    """Run `run` of a batched HeadOutputs as an N = 1 HeadOutputs (views, no copy: runs are contiguous slabs)."""
    if not ho.last_run_valid and run == ho.num_runs - 1:
        raise hotpath.hip.PodError("run {} was skipped by skip_unused_last_run: request all runs from the model".format(run))
    sel = lambda lst: None if lst is None else [t[run:run + 1] for t in lst]
    return HeadOutputs(sel(ho.cls), sel(ho.delta), sel(ho.cls_var), sel(ho.reg_var), ho.anchors, ho.shapes, ho.num_anchors,
                       ho.num_classes, ho.image_size)


def stack_members(members: List[HeadOutputs]) -> HeadOutputs:
    """Ensemble members (each N = 1) -> one HeadOutputs with runs = members: the (members, A*C, H, W) layout K1
    streams.  On 8 GPUs (BASELINE config 5) the same tensor is the destination of the RCCL gather."""
    first = members[0]
    cat = lambda name: None if getattr(first, name) is None else \
        [torch.cat([getattr(m, name)[l] for m in members], 0).contiguous() for l in range(len(first.cls))]
    return HeadOutputs(cat("cls"), cat("delta"), cat("cls_var"), cat("reg_var"), first.anchors, first.shapes,
                       first.num_anchors, first.num_classes, first.image_size)


def detections_to_instances(det: hotpath.DeviceDetections) -> Instances:
    """Fixed-capacity device buffers -> `Instances` (one host sync for the count; int64 classes as the reference)."""
    m = det.count()
    res = Instances(det.image_size)
    res.pred_boxes = Boxes(det.boxes[:m])
    res.scores = det.scores[:m]
    res.pred_classes = det.classes[:m].long()
    res.pred_cls_probs = det.probs[:m]
    res.pred_boxes_covariance = det.cov[:m]
    return res
