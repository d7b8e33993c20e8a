"""Weight loading (PI:59-84): detectron2-named checkpoints -> pod_compare_amd.modeling, on the CPU.

The expected detectron2 key names are enumerated HERE, independently of pod_compare_amd.checkpoint's map: detectron2's
ResNet (`stem.conv1`, `res<s>.<b>.{shortcut,conv1,conv2,conv3}` each with a `.norm` FrozenBatchNorm2d), FPN
(`fpn_lateral<l>`, `fpn_output<l>`, `top_block.p6/p7`) and the reference's head (nn.Sequential subnets whose convs sit at
0,3,6,9 with Dropout entries and 0,2,4,6 without, PR:403-427)."""
import os
import pickle

import numpy as np
import pytest
import torch

from pod_compare_amd import checkpoint, config, modeling
from pod_compare_amd import probabilistic_inference as pinf

CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pod_compare_amd", "configs")
M_YAML = os.path.join(CFG, "BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml")
I_YAML = os.path.join(CFG, "Inference/bayes_od_mc_dropout.yaml")


def expected_detectron2_keys(dropout: bool, cls_var=True, bbox_cov=True):
    keys = []
    bn = ("weight", "bias", "running_mean", "running_var")

    def conv_bn(name):
        keys.append(name + ".weight")
        keys.extend(name + ".norm." + f for f in bn)

    conv_bn("backbone.bottom_up.stem.conv1")
    for stage, blocks in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
        for b in range(blocks):
            if b == 0:
                conv_bn("backbone.bottom_up.%s.%d.shortcut" % (stage, b))
            for c in ("conv1", "conv2", "conv3"):
                conv_bn("backbone.bottom_up.%s.%d.%s" % (stage, b, c))
    for name in ["backbone.fpn_lateral%d" % l for l in (3, 4, 5)] + ["backbone.fpn_output%d" % l for l in (3, 4, 5)] + \
            ["backbone.top_block.p6", "backbone.top_block.p7"]:
        keys.extend((name + ".weight", name + ".bias"))
    step = 3 if dropout else 2
    for sub in ("cls_subnet", "bbox_subnet"):
        for j in range(4):
            keys.extend(("head.%s.%d.weight" % (sub, j * step), "head.%s.%d.bias" % (sub, j * step)))
    for name in ["cls_score", "bbox_pred"] + (["cls_var"] if cls_var else []) + (["bbox_cov"] if bbox_cov else []):
        keys.extend(("head.%s.weight" % name, "head.%s.bias" % name))
    return keys


def small_model(seed, dropout=0.2):
    torch.manual_seed(seed)
    return modeling.ProbabilisticRetinaNet(dropout_rate=dropout, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                           bbox_cov_loss="negative_log_likelihood").eval()


def randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, modeling.FrozenBatchNorm2d):
            n = m.weight.numel()
            m.weight.copy_(0.5 + torch.rand(n, generator=g))
            m.bias.copy_(0.2 * torch.randn(n, generator=g))
            m.running_mean.copy_(0.3 * torch.randn(n, generator=g))
            m.running_var.copy_(0.5 + torch.rand(n, generator=g))


@pytest.mark.parametrize("dropout_entries", [True, False])
def test_exported_names_are_detectron2s(dropout_entries):
    model = small_model(1)
    sd = checkpoint.to_detectron2_state_dict(model, with_dropout_entries=dropout_entries)
    assert sorted(sd) == sorted(expected_detectron2_keys(dropout_entries))
    assert tuple(sd["backbone.bottom_up.res3.0.shortcut.weight"].shape) == (512, 256, 1, 1)
    assert tuple(sd["backbone.fpn_lateral5.weight"].shape) == (256, 2048, 1, 1)
    assert tuple(sd["backbone.top_block.p6.weight"].shape) == (256, 2048, 3, 3)
    assert tuple(sd["head.cls_var.weight"].shape) == (63, 256, 3, 3) and tuple(sd["head.bbox_cov.weight"].shape) == (36, 256, 3, 3)


@pytest.mark.parametrize("dropout_entries", [True, False])
def test_round_trip_with_non_identity_bn_statistics(tmp_path, dropout_entries):
    """A detectron2-named file (written here under hand-enumerated names) loads into a fresh model; after BN folding the
    network output equals the source model's unfolded forward (which also gives fold_frozen_bn a non-trivial test)."""
    src = small_model(3)
    randomise_bn(src, 4)
    own = src.state_dict()
    kmap = checkpoint.detectron2_to_local_keys(src, 3 if dropout_entries else 2)
    sd = {k: own[kmap[k]].clone() for k in expected_detectron2_keys(dropout_entries)}       # every expected name must be mapped
    sd["anchor_generator.cell_anchors.0"] = torch.zeros(9, 4)                                # ignored buffers
    sd["pixel_mean"] = torch.zeros(3, 1, 1)
    path = str(tmp_path / "model_final.pth")
    torch.save({"model": sd, "iteration": 89999}, path)

    dst = small_model(99)
    missing, unexpected = checkpoint.load_detectron2_state_dict(dst, checkpoint.read_checkpoint_file(path), strict=True)
    assert missing == [] and unexpected == []
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b), ka
    image = torch.randint(0, 256, (3, 64, 96), generator=torch.Generator().manual_seed(5), dtype=torch.uint8)
    ref = src(image)                                   # unfolded: conv -> FrozenBatchNorm2d
    assert modeling.fold_frozen_bn(dst) == 53          # 1 stem + 16 blocks x 3 + 4 shortcuts
    got = dst(image)
    for a, b in zip(ref.cls + ref.delta + ref.cls_var + ref.reg_var, got.cls + got.delta + got.cls_var + got.reg_var):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()), float((a - b).abs().max())     # fp32 re-association only
    with pytest.raises(checkpoint.CheckpointError):
        checkpoint.load_detectron2_state_dict(dst, sd)          # loading after folding must fail loudly


def test_shape_mismatch_and_missing_file_raise(tmp_path):
    model = small_model(1)
    sd = checkpoint.to_detectron2_state_dict(model)
    sd["head.cls_score.weight"] = torch.zeros(720, 256, 3, 3)      # an 80-class COCO head
    with pytest.raises(checkpoint.CheckpointError, match="shape"):
        checkpoint.load_detectron2_state_dict(small_model(2), sd)
    with pytest.raises(checkpoint.CheckpointError, match="does not exist"):
        checkpoint.read_checkpoint_file(str(tmp_path / "nope.pth"))
    with pytest.raises(checkpoint.CheckpointError, match="remote"):
        checkpoint.read_checkpoint_file("detectron2://ImageNetPretrained/MSRA/R-50.pkl")


def test_caffe2_backbone_pkl(tmp_path):
    """ImageNet-pretrained MSRA R-50 files: Caffe2 names, affine-only BN (no statistics), numpy arrays, no head."""
    src = small_model(7)
    randomise_bn(src, 8)
    own = src.state_dict()
    blobs = {"conv1_w": own["bottom_up.stem.0.weight"].numpy(), "res_conv1_bn_s": own["bottom_up.stem.1.weight"].numpy(),
             "res_conv1_bn_b": own["bottom_up.stem.1.bias"].numpy(), "fc1000_w": np.zeros((1000, 2048), np.float32)}
    c2 = {"conv1": "branch2a", "conv2": "branch2b", "conv3": "branch2c", "shortcut": "branch1"}
    for stage, blocks in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
        for b in range(blocks):
            for c, name in c2.items():
                if c == "shortcut" and b > 0:
                    continue
                base = "bottom_up.%s.%d.%s" % (stage, b, c)
                blobs["%s_%d_%s_w" % (stage, b, name)] = own[base + ".0.weight"].numpy()
                blobs["%s_%d_%s_bn_s" % (stage, b, name)] = own[base + ".1.weight"].numpy()
                blobs["%s_%d_%s_bn_b" % (stage, b, name)] = own[base + ".1.bias"].numpy()
    path = str(tmp_path / "R-50.pkl")
    with open(path, "wb") as f:
        pickle.dump({"model": blobs, "__author__": "Caffe2"}, f)
    dst = small_model(9)
    missing, unexpected = checkpoint.load_detectron2_state_dict(dst, checkpoint.read_checkpoint_file(path))
    assert unexpected == [] and all(k.startswith(("fpn.", "head.")) for k in missing)
    got = dst.state_dict()
    assert torch.equal(got["bottom_up.res4.5.conv2.0.weight"], own["bottom_up.res4.5.conv2.0.weight"])
    assert torch.equal(got["bottom_up.res2.0.shortcut.1.bias"], own["bottom_up.res2.0.shortcut.1.bias"])
    bn = dst.bottom_up.res3[1].conv1[1]
    assert float(bn.running_mean.abs().max()) == 0.0 and torch.allclose(bn.running_var + bn.eps, torch.ones_like(bn.running_var))


def _cfg(tmp_path, mode_yaml=I_YAML, model_yaml=M_YAML):
    cfg = config.setup_config(model_yaml, mode_yaml, random_seed=0, data_dir=str(tmp_path))
    cfg.MODEL.DEVICE = "cpu"
    return cfg


def test_predictor_loads_last_checkpoint_of_output_dir(tmp_path):
    """PI:78-84: `<OUTPUT_DIR>/last_checkpoint` names the file; MODEL.WEIGHTS (the ImageNet URL of the BDD yamls) is then
    never touched.  CS:170-182: OUTPUT_DIR layout and the missing-directory error."""
    with pytest.raises(NotADirectoryError):
        config.setup_config(M_YAML, I_YAML, random_seed=0, data_dir=str(tmp_path), is_testing=True)
    cfg = _cfg(tmp_path)
    assert cfg.OUTPUT_DIR == os.path.join(str(tmp_path), "BDD-Detection", "retinanet", "retinanet_R_50_FPN_1x_reg_cls_var_dropout", "random_seed_0")
    os.makedirs(cfg.OUTPUT_DIR)
    # no last_checkpoint: the reference falls back to MODEL.WEIGHTS = detectron2://... which cannot be fetched -> loud failure
    with pytest.raises(checkpoint.CheckpointError, match="remote"):
        pinf.build_predictor(cfg)
    src = small_model(11)
    randomise_bn(src, 12)
    torch.save({"model": checkpoint.to_detectron2_state_dict(src)}, os.path.join(cfg.OUTPUT_DIR, "model_final.pth"))
    with open(os.path.join(cfg.OUTPUT_DIR, "last_checkpoint"), "w") as f:
        f.write("model_final.pth")
    pred = pinf.build_predictor(cfg)
    assert pred.model.loaded_from == os.path.join(cfg.OUTPUT_DIR, "model_final.pth")
    modeling.fold_frozen_bn(src)
    assert torch.equal(pred.model.head.cls_var.weight, src.head.cls_var.weight)
    assert torch.allclose(pred.model.bottom_up.res5[2].conv3[0].bias, src.bottom_up.res5[2].conv3[0].bias)
    assert float(pred.model.bottom_up.res5[2].conv3[0].bias.abs().max()) > 0.0          # non-identity statistics were folded


def test_ensemble_members_load_from_sibling_seed_directories(tmp_path):
    """PI:59-77: member s comes from `<parent of OUTPUT_DIR>/random_seed_<s>`; self.model itself is not loaded."""
    cfg = _cfg(tmp_path, os.path.join(CFG, "Inference/ensembles_pre_nms.yaml"),
               os.path.join(CFG, "BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var.yaml"))
    seeds = [0, 1000]
    cfg.PROBABILISTIC_INFERENCE.ENSEMBLES.RANDOM_SEED_NUMS = seeds
    srcs = {}
    for s in seeds:
        d = pinf.ensemble_member_dir(cfg, s)
        assert d == os.path.join(os.path.dirname(cfg.OUTPUT_DIR), "random_seed_%d" % s)
        os.makedirs(d)
        srcs[s] = small_model(100 + s, dropout=0.0)
        torch.save({"model": checkpoint.to_detectron2_state_dict(srcs[s])}, os.path.join(d, "model_0089999.pth"))
        with open(os.path.join(d, "last_checkpoint"), "w") as f:
            f.write("model_0089999.pth")
    pred = pinf.build_predictor(cfg)
    assert pred.model.loaded_from == "" and len(pred.model_list) == 2
    for s, m in zip(seeds, pred.model_list):
        assert m.loaded_from.endswith("random_seed_%d/model_0089999.pth" % s)
        assert torch.equal(m.head.bbox_pred.weight, srcs[s].head.bbox_pred.weight)
    assert not torch.equal(pred.model_list[0].head.bbox_pred.weight, pred.model_list[1].head.bbox_pred.weight)


def test_explicit_random_init(tmp_path):
    cfg = _cfg(tmp_path)
    cfg.MODEL.WEIGHTS, cfg.OUTPUT_DIR = "", ""
    pred = pinf.build_predictor(cfg)
    assert pred.model.loaded_from == ""
