"""Times the conv net alone (ResNet-50-FPN + probabilistic head, N MC runs) under different PyTorch-ROCm settings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import modeling, synthetic
N = int(os.environ.get("RUNS", "10"))
dev = torch.device("cuda", 0)
frame = synthetic.synthetic_frame(0, device=dev)
def run(tag, benchmark, channels_last, steps=8):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(0)
    m = modeling.ProbabilisticRetinaNet(dropout_rate=0.2, cls_var_loss="loss_attenuation", cls_var_num_samples=10,
                                        bbox_cov_loss="negative_log_likelihood").to(dev).eval()
    modeling.fold_frozen_bn(m)
    if channels_last:
        m = m.to(memory_format=torch.channels_last)
    img = modeling.resize_test_image(frame)
    with torch.no_grad():
        for _ in range(3):
            ho = m(img, num_mc_dropout_runs=N, skip_unused_last_run=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            ho = m(img, num_mc_dropout_runs=N, skip_unused_last_run=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("%-40s %7.2f ms/image   out contiguous=%s" % (tag, dt * 1e3, ho.cls[0].is_contiguous()), flush=True)
want = os.environ.get("VARIANTS", "default,cudnn.benchmark,channels_last,benchmark+channels_last").split(",")
for tag, b, cl in (("default", False, False), ("cudnn.benchmark", True, False), ("channels_last", False, True), ("benchmark+channels_last", True, True)):
    if tag not in want:
        continue
    try:
        run(tag, b, cl)
    except Exception as e:
        print(tag, "FAILED", repr(e)[:200])
