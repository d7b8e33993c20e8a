"""GPU box diagnostics: per-workgroup phase time stamps (s_memtime) of pod_wino_conv3x3 on the launch bench.py times.
    POD_TRACE=1 POD_BUILD_TAG=trace python -m pod_compare_amd.build          (here, before gpurun)
    POD_MI355X_LIB=pod_compare_amd/lib/trace/libpod_mi355x.so python tools/wino_trace.py [copies] [p3|bench]
Prints the median / p10 / p90 shader cycles of: descriptor -> first loads landed | prologue transform | K loop | accumulator
dump | store pass, the rounds a CU worked (workgroups per CU), and the launch's span in cycles."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pod_compare_amd import hip  # noqa: E402
from pod_compare_amd.wino import WinoConv, block_table  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 19
mode = sys.argv[2] if len(sys.argv) > 2 else "bench"
levels = [(90, 160)] if mode == "p3" else [(96, 168), (48, 84), (24, 42), (12, 21), (6, 11)]
dev = torch.device("cuda")
torch.manual_seed(0)
conv = WinoConv(torch.randn(256, 256, 3, 3, device=dev) * 0.03, torch.randn(256, device=dev))
tab = block_table(levels, copies, dev)
src = torch.randn(tab.pod_pixels, 256, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    conv(src, dst, tab, relu=True, dropout_p=0.1, seed=1)
torch.cuda.synchronize()
n_wg = tab.shape[0] * 4
lib = ctypes.CDLL(hip.library_path())
host = np.zeros((min(n_wg, 8192), 16), dtype=np.int64)
import os  # noqa: E402
form8 = os.environ.get("POD_WINO_FORM") == "8"      # the eight-wavefront form keeps its own stamps (k16_wino_conv_split8.hip)
dump = (lib.pod_wino_trace_dump_split8 if form8 else lib.pod_wino_trace_dump_split) if conv.split else lib.pod_wino_trace_dump
dump.argtypes = [ctypes.c_void_p, ctypes.c_int32]
assert dump(host.ctypes.data, host.shape[0]) == 0
t = host[:, :6]
live = t[:, 5] > 0
t = t[live]
names = ["first loads landed", "prologue transform", "K loop", "accumulator dump", "store pass"]
print("%d workgroups traced of %d; shader cycles per phase (median, p10, p90):" % (t.shape[0], n_wg))
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("  %-20s %8.0f %8.0f %8.0f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
if host[live, 8].any():
    e = host[live]
    for nm, i0, i1 in (("  record arrived", 0, 8), ("  mini + filter loads issued", 8, 9), ("  stage offsets", 9, 10), ("  16 stage pieces issued", 10, 11), ("  wait + barrier", 11, 1)):
        d = e[:, i1] - e[:, i0]
        print("  %-28s %8.0f %8.0f %8.0f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
if host[live, 13].any():       # constant-rate 100 MHz stamps at both ends of a workgroup: the shader clock the kernel really ran at
    e = host[live]
    wall_ns = (e[:, 13] - e[:, 12]) * 10.0
    cyc = e[:, 5] - e[:, 0]
    print("  workgroup wall time %.1f us median; shader clock = cycles / wall = %.3f GHz" % (np.median(wall_ns) / 1e3, np.median(cyc / wall_ns)))
    print("  launch span by the 100 MHz clock: %.3f ms" % ((e[:, 13].max() - e[:, 12].min()) * 1e-5))
tot = t[:, 5] - t[:, 0]
print("  %-20s %8.0f %8.0f %8.0f" % ("workgroup total", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
span = t[:, 5].max() - t[:, 0].min()
print("launch span %d cycles; sum of workgroup cycles / 256 CUs = %.0f (%.3f of the span)" % (span, tot.sum() / 256.0, tot.sum() / 256.0 / span))
hw = host[live, 6]
cu = ((hw >> 32) & 0xF) * 1024 + ((hw >> 8) & 0xF) + 16 * ((hw >> 12) & 0x7) + 128 * ((hw >> 16) & 0x3)   # xcc, cu_id, sh_id, se_id
u, c = np.unique(cu, return_counts=True)
print("CUs seen %d; workgroups per CU min %d median %d max %d" % (len(u), c.min(), int(np.median(c)), c.max()))
# gaps between consecutive workgroups on the same CU
gaps = []
for k in u:
    rows = t[cu == k]
    rows = rows[np.argsort(rows[:, 0])]
    gaps.extend((rows[1:, 0] - rows[:-1, 5]).tolist())
gaps = np.array(gaps)
print("gap between a workgroup's last stamp and the next one's first on the same CU: median %.0f p90 %.0f" % (np.median(gaps), np.percentile(gaps, 90)))
