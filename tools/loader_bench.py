"""GPU box: apply_net on a COCO-format list of real files -- what the host-side loader threads are worth.
    python tools/loader_bench.py <n_images> <out.txt>
Writes n JPEG frames of 1280x720 (smooth synthetic content, quality 90) under /tmp, then runs pod_compare_amd.apply_net on them with
--loader-workers 0 / 4 / 8 / 16 for a single-model config and the MC-dropout config and collects the `inference loop:` lines."""
import json
import os
import subprocess
import sys

import numpy as np
from PIL import Image

n, out = int(sys.argv[1]), sys.argv[2]
root = "/tmp/pod_loader_bench"
os.makedirs(root, exist_ok=True)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:720, 0:1280].astype(np.float32)
images = []
for k in range(n):
    f = rng.uniform(0.002, 0.02, size=6)
    img = np.stack([127 + 120 * np.sin(f[2 * c] * xx + k) * np.cos(f[2 * c + 1] * yy) for c in range(3)], axis=-1)
    img += rng.normal(0, 4, size=img.shape)
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, "%05d.jpg" % k), quality=90)
    images.append({"id": k + 1, "file_name": "%05d.jpg" % k, "height": 720, "width": 1280})
json.dump({"images": images}, open(os.path.join(root, "set.json"), "w"))
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = {"cfg2 (reg_cls_var + bayes_od)": ["--config-file", "pod_compare_amd/configs/BDD-Detection/retinanet/retinanet_R_50_FPN_1x_reg_cls_var.yaml",
                                          "--inference-config", "pod_compare_amd/configs/Inference/bayes_od.yaml"],
        "cfg3 (MC dropout, N = 10)": []}
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
with open(out, "w") as fp:
    for name, extra in cfgs.items():
        for workers in (0, 4, 8, 16):
            cmd = [sys.executable, "-m", "pod_compare_amd.apply_net", "--coco-json", os.path.join(root, "set.json"), "--image-root", root, "--random-init",
                   "--output", "/tmp/pod_loader_bench/out.json", "--loader-workers", str(workers)] + extra
            r = subprocess.run(cmd, cwd=here, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1200)
            line = [l for l in r.stdout.splitlines() if l.startswith("inference loop")]
            msg = "%s, --loader-workers %d: %s" % (name, workers, line[-1] if line else "FAILED rc=%d\n%s" % (r.returncode, r.stdout[-2000:]))
            print(msg)
            fp.write(msg + "\n")
