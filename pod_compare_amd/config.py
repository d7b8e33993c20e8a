"""Config loading with the reference's key names (core/setup.py:79-212), without yacs/detectron2.

`get_cfg()` returns the defaults the path reads (detectron2 RetinaNet defaults + add_probabilistic_config,
core/setup.py:90-133); `merge_from_file` follows `_BASE_` chains like detectron2's CfgNode and loads the
reference's YAMLs unmodified.  Base-RetinaNet.yaml:8 carries a `!!python/object/apply:eval` tag: it is
never evaluated -- the node is replaced by the literal anchor sizes it would produce.
"""
import copy
import os
from typing import Any, Dict

import yaml

from .anchors import ANCHOR_SIZES


class CfgNode(dict):
    """dict with attribute access (enough of yacs.CfgNode for the predictor surface)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        return None

    def defrost(self):
        return None

    def merge_from_dict(self, other: Dict[str, Any]):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_dict(v)
            else:
                self[k] = _wrap(v)

    def merge_from_file(self, path: str):
        self.merge_from_dict(load_yaml_with_base(path))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = yaml.safe_load(value) if isinstance(value, str) else value


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        return CfgNode({k: _wrap(x) for k, x in v.items()})
    return v


class _SafeLoaderNoEval(yaml.SafeLoader):
    pass


def _anchor_sizes_literal(loader, suffix, node):   # the only python/* tag in the reference's YAMLs
    return [list(s) for s in ANCHOR_SIZES]


_SafeLoaderNoEval.add_multi_constructor("tag:yaml.org,2002:python/", _anchor_sizes_literal)


def load_yaml_with_base(path: str) -> Dict[str, Any]:
    with open(path, "r") as f:
        cfg = yaml.load(f, Loader=_SafeLoaderNoEval) or {}
    base = cfg.pop("_BASE_", None)
    if base is None:
        return cfg
    if not os.path.isabs(base) and not base.startswith("~"):
        base = os.path.join(os.path.dirname(path), base)
    merged = load_yaml_with_base(base)

    def deep(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                deep(a[k], v)
            else:
                a[k] = v
        return a

    return deep(merged, cfg)


def get_cfg() -> CfgNode:
    """Defaults of the keys the inference path reads."""
    return _wrap({
        "VERSION": 2,
        "OUTPUT_DIR": "./output",
        "SEED": 0,
        "DATASETS": {"TRAIN": (), "TEST": ()},
        "DATALOADER": {"NUM_WORKERS": 4},
        "SOLVER": {"IMS_PER_BATCH": 1, "STEPS": (60000, 80000)},
        "INPUT": {"MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR"},
        "MODEL": {
            "META_ARCHITECTURE": "ProbabilisticRetinaNet",
            "DEVICE": "cuda",
            "WEIGHTS": "",
            "PIXEL_MEAN": [103.530, 116.280, 123.675],
            "PIXEL_STD": [1.0, 1.0, 1.0],
            "BACKBONE": {"NAME": "build_retinanet_resnet_fpn_backbone"},
            "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res3", "res4", "res5"]},
            "FPN": {"IN_FEATURES": ["res3", "res4", "res5"]},
            "ANCHOR_GENERATOR": {"SIZES": [list(s) for s in ANCHOR_SIZES], "ASPECT_RATIOS": [[0.5, 1.0, 2.0]]},
            "RPN": {"BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0)},
            "RETINANET": {"NUM_CLASSES": 80, "NUM_CONVS": 4, "PRIOR_PROB": 0.01, "SCORE_THRESH_TEST": 0.05,
                          "TOPK_CANDIDATES_TEST": 1000, "NMS_THRESH_TEST": 0.5, "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0),
                          "IN_FEATURES": ["p3", "p4", "p5", "p6", "p7"]},
            # add_probabilistic_config, core/setup.py:90-107
            "PROBABILISTIC_MODELING": {"MC_DROPOUT": {}, "ANNEALING_STEP": 0, "DROPOUT_RATE": 0.0,
                                       "CLS_VAR_LOSS": {"NAME": "none", "NUM_SAMPLES": 3},
                                       "BBOX_COV_LOSS": {"NAME": "none", "COVARIANCE_TYPE": "diagonal", "NUM_SAMPLES": 1000}},
        },
        "TEST": {"DETECTIONS_PER_IMAGE": 100},
        # core/setup.py:109-133
        "PROBABILISTIC_INFERENCE": {"INFERENCE_MODE": "standard_nms", "AFFINITY_THRESHOLD": 0.7,
                                    "MC_DROPOUT": {"ENABLE": False, "NUM_RUNS": 1},
                                    "BAYES_OD": {"BOX_MERGE_MODE": "bayesian_inference", "CLS_MERGE_MODE": "bayesian_inference",
                                                 "DIRCH_PRIOR": "uniform"},
                                    "ENSEMBLES_DROPOUT": {"BOX_MERGE_MODE": "pre_nms"},
                                    "ENSEMBLES": {"BOX_MERGE_MODE": "pre_nms", "RANDOM_SEED_NUMS": [0, 1000, 2000, 3000, 4000]}},
    })


def setup_config(config_file: str, inference_config: str = "", random_seed: int = 0, opts=(), data_dir: str = "",
                 is_testing: bool = False) -> CfgNode:
    """core/setup.py:136-212 minus dataset registration / logging: model YAML, then inference YAML.

    data_dir: the reference's `core.data_dir()`; when given, OUTPUT_DIR becomes
    `<data_dir>/<dataset>/<model family>/<config name>/random_seed_<seed>` (CS:170-176) -- the directory whose
    `last_checkpoint` the predictor loads (PI:78-84) -- and, with is_testing, a missing directory raises
    NotADirectoryError (CS:178-182)."""
    cfg = get_cfg()
    cfg.merge_from_file(config_file)
    if inference_config:
        cfg.merge_from_file(inference_config)
    if opts:
        cfg.merge_from_list(list(opts))
    if data_dir:
        model_dir = os.path.dirname(os.path.abspath(config_file))
        model_name = os.path.basename(model_dir)
        dataset_name = os.path.basename(os.path.dirname(model_dir))
        cfg.OUTPUT_DIR = os.path.join(data_dir, dataset_name, model_name, os.path.basename(config_file)[:-5],
                                      "random_seed_" + str(random_seed))
        if is_testing and not os.path.isdir(cfg.OUTPUT_DIR):
            raise NotADirectoryError("Checkpoint directory {} does not exist.".format(cfg.OUTPUT_DIR))
    cfg.SEED = random_seed
    return cfg
