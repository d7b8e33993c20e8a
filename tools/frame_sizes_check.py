"""GPU box: the whole predictor (random-init cfg3 model, class scores lifted) on frames of unusual sizes -- K11 canvases, the hot path's level geometry
and the output rescale must cope with any H x W; prints detections per frame and whether every box / covariance is finite."""
import sys, os
sys.path.insert(0, ".")
import torch
from pod_compare_amd import config, probabilistic_inference as pinf
CFG = "pod_compare_amd/configs"
cfg = config.setup_config(os.path.join(CFG, "BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var_dropout.yaml"), os.path.join(CFG, "Inference/bayes_od_mc_dropout.yaml"))
cfg.MODEL.WEIGHTS, cfg.OUTPUT_DIR, cfg.MODEL.DEVICE = "", "", "cuda:0"
torch.manual_seed(0)
pred = pinf.build_predictor(cfg)
with torch.no_grad():
    pred.model.head.cls_score.bias.add_(3.0)      # random-init scores sit at the 0.01 prior: lift them so that the post-processing has work
g = torch.Generator().manual_seed(1)
from pod_compare_amd import modeling
for hw in [(1080, 1920), (240, 320), (333, 1777), (1536, 2048), (97, 131)]:
    frame = torch.randint(0, 255, (3,) + hw, dtype=torch.uint8, generator=g).cuda()
    image = modeling.resize_test_image(frame, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    inst = pred([{"image": image, "height": hw[0], "width": hw[1], "image_id": 7}])
    n = len(inst.pred_boxes.tensor)
    ok = (bool(torch.isfinite(inst.pred_boxes.tensor).all()) and bool(torch.isfinite(inst.pred_boxes_covariance).all())) if n > 0 else True
    again = pred([{"image": image, "height": hw[0], "width": hw[1], "image_id": 7}])
    same = n == len(again.pred_boxes.tensor) and (n == 0 or bool(torch.equal(inst.scores, again.scores)))
    print(hw, "->", tuple(image.shape[1:]), "detections", n, "finite", ok, "same on a second call (same image_id, dropout seeds differ per call: scores may differ)", same)
