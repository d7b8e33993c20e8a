"""GPU box: where a wavefront of pod_conv1x1_split spends its time (build: POD_BUILD_TAG=c1t POD_EXTRA_DEFINES=-DPOD_C1_TRACE python -m
pod_compare_amd.build; run: POD_MI355X_LIB=pod_compare_amd/lib/c1t/libpod_mi355x.so python tools/conv1x1_trace.py)."""
import ctypes
import sys
import torch
sys.path.insert(0, ".")
from pod_compare_amd import hip  # noqa: E402
from pod_compare_amd.conv1x1 import Conv1x1  # noqa: E402
lib = ctypes.CDLL(hip.library_path())
for name, cin, cout, h, w, s, res in (("res4 conv3", 256, 1024, 48, 84, 1, True), ("res2 conv3", 64, 256, 192, 336, 1, True), ("res3 conv3", 128, 512, 96, 168, 1, True),
                                      ("res5 conv3", 512, 2048, 24, 42, 1, True), ("res4 conv1", 1024, 256, 48, 84, 1, False), ("fpn lateral5", 2048, 256, 24, 42, 1, False),
                                      ("res2 conv1", 256, 64, 192, 336, 1, False)):
    torch.manual_seed(1)
    wt = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    x = torch.randn(h * w, cin, device="cuda").relu()
    conv = Conv1x1(wt, b, s)
    ho, wo = conv.out_hw(h, w)
    r = torch.randn(ho * wo, cout, device="cuda") if res else None
    print("%s (%d -> %d, %d pixels, splits %d): the last of 200 back-to-back launches" % (name, cin, cout, ho * wo, conv.splits_for(ho * wo)), file=sys.stderr, flush=True)
    for _ in range(200):
        conv(x, h, w, relu=True, residual=r)
    torch.cuda.synchronize()
    lib.pod_c1_trace_dump()
