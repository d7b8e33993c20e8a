import sys, time, torch
sys.path.insert(0, ".")
from pod_compare_amd import modeling, synthetic
torch.manual_seed(0)
model = modeling.ProbabilisticRetinaNet(dropout_rate=0.1, cls_var_loss="loss_attenuation", cls_var_num_samples=10, bbox_cov_loss="negative_log_likelihood").cuda().eval()
modeling.fold_frozen_bn(model)
for q in model.parameters(): q.requires_grad_(False)
img = modeling.resize_test_image(synthetic.synthetic_frame(0, 720, 1280, device="cuda"))
with torch.no_grad():
    for _ in range(5): model(img, num_mc_dropout_runs=10, skip_unused_last_run=True)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n): model(img, num_mc_dropout_runs=10, skip_unused_last_run=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host enqueue %.2f ms/image, with device %.2f ms/image" % (1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
