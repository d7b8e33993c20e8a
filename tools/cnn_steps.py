import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import modeling, synthetic
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = modeling.ProbabilisticRetinaNet(dropout_rate=0.2, cls_var_loss="loss_attenuation", cls_var_num_samples=10, bbox_cov_loss="negative_log_likelihood").to(dev).eval()
modeling.fold_frozen_bn(m)
img = modeling.resize_test_image(synthetic.synthetic_frame(0, device=dev))
with torch.no_grad():
    for _ in range(3): m(img, num_mc_dropout_runs=10, skip_unused_last_run=True)
    torch.cuda.synchronize(); time.sleep(0.5)
    t = time.perf_counter()
    for _ in range(5): m(img, num_mc_dropout_runs=10, skip_unused_last_run=True)
    torch.cuda.synchronize()
    print("steady ms/img", (time.perf_counter() - t) / 5 * 1e3)
