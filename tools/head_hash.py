"""GPU box: checksum of the model's raw head outputs (the NCHW planes the predictor launches of pod_wino_conv3x3_split store) on a fixed frame, eval mode and
an MC-dropout forward, to compare two builds of the library bit for bit:   POD_MI355X_LIB=<lib> python tools/head_hash.py"""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from pod_compare_amd import modeling  # noqa: E402

torch.manual_seed(0)
m = modeling.ProbabilisticRetinaNet(dropout_rate=0.1, cls_var_loss="loss_attenuation", bbox_cov_loss="negative_log_likelihood").cuda().eval()
modeling.fold_frozen_bn(m)
f = torch.randint(0, 256, (3, 300, 420), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for kw in (dict(), dict(num_mc_dropout_runs=3, skip_unused_last_run=True)):
    ho = m(f, **kw)
    h = hashlib.sha256()
    for ts in (ho.cls, ho.delta, ho.cls_var, ho.reg_var):
        for t in ts:
            h.update(t.cpu().numpy().tobytes())
    print(sorted(kw), h.hexdigest()[:20], float(ho.cls[0].abs().max()))
