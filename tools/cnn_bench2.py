import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pod_compare_amd import modeling, synthetic
dev = torch.device("cuda", 0)
img = modeling.resize_test_image(synthetic.synthetic_frame(0, device=dev))
def run(tag, fold, fused, graph=False, steps=8):
    torch.manual_seed(0)
    m = modeling.ProbabilisticRetinaNet(dropout_rate=0.2, cls_var_loss="loss_attenuation", cls_var_num_samples=10, bbox_cov_loss="negative_log_likelihood").to(dev).eval()
    if fold: print("folded", modeling.fold_frozen_bn(m), end="  ")
    m.head.fused_relu_dropout = fused
    f = lambda: m(img, num_mc_dropout_runs=10, skip_unused_last_run=True)
    with torch.no_grad():
        for _ in range(3): ho = f()
        torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                f(); torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    ho = f()
            torch.cuda.synchronize()
            f = g.replay
            f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps): f()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    # dropout sanity: fraction of exact zeros after relu+dropout is ~ 0.5 + 0.5*0.2
    print("%-34s %7.2f ms/image  cls mean %.4f std %.4f" % (tag, dt * 1e3, float(ho.cls[0].mean()), float(ho.cls[0].std())), flush=True)
run("baseline", False, False)
run("fold BN", True, False)
run("fold BN + fused relu/dropout", True, True)
try:
    run("fold + fused + CUDA graph", True, True, graph=True)
except Exception as e:
    print("graph FAILED", repr(e)[:300])
