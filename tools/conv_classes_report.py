"""Attributes the dispatches of a rocprofv3 run of tools/conv_classes.py to the convolution classes (by order) and writes the per-class
roofline table.   python tools/conv_classes_report.py <plan.json> <kernel_trace.csv> <pmc FETCH csv> <pmc WRITE csv> <out.json> <out.md>"""
import csv
import json
import sys

plan = json.load(open(sys.argv[1]))
NAMES = ("k_conv1x1_split", "k_conv1x1_reduce", "k_wino_conv3x3_split", "k_wino_reduce")


def conv_rows(path, value):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        if any(n in name for n in NAMES):
            rows.append((int(r.get("Dispatch_Id") or r.get("Dispatch Id") or len(rows)), value(r)))
    rows.sort()
    return [v for _, v in rows]


dur = conv_rows(sys.argv[2], lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)          # us


def pmc(path):
    return conv_rows(path, lambda r: float(r["Counter_Value"]))


fetch, write = pmc(sys.argv[3]), pmc(sys.argv[4])
need = sum(p["launches"] * p["kernels_per_launch"] for p in plan)
assert len(dur) == need and len(fetch) == need and len(write) == need, (len(dur), len(fetch), len(write), need)
HBM, F16 = 8.0e12, 2.5e15
out, at = [], 0
for p in plan:
    m = p["launches"] * p["kernels_per_launch"]
    d, f, w = dur[at:at + m], fetch[at:at + m], write[at:at + m]
    at += m
    keep = slice(2 * p["kernels_per_launch"], None)                    # drop the first two launches (cold caches)
    per = lambda xs: sum(xs[keep]) / (p["launches"] - 2)
    us = per(d)
    traffic = 2 * per(f) * 1024 + per(w) * 1024                        # FETCH_SIZE x 2 (gfx950, 16-byte loads), KB units; WRITE_SIZE as reported
    bound_us = max(p["bytes"] / HBM, p["executed_f16_flop"] / F16) * 1e6
    out.append(dict(p, us_per_launch=us, traffic_bytes=traffic, traffic_over_algorithmic=traffic / p["bytes"], bound_us=bound_us,
                    bound="hbm" if p["bytes"] / HBM >= p["executed_f16_flop"] / F16 else "mfma", frac_of_bound=bound_us / us,
                    hbm_gbps=p["bytes"] / us * 1e-3, f16_tflops_executed=p["executed_f16_flop"] / us * 1e-6))
json.dump(out, open(sys.argv[5], "w"), indent=1)
with open(sys.argv[6], "w") as f:
    f.write("| class | calls / image | us / launch | algorithmic MB | HBM-side traffic MB (x algorithmic) | executed f16 GFLOP | bound | bound us | achieved / bound | us / image |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for o in out:
        f.write("| %s %s | %d | %.1f | %.2f | %.2f (%.2fx) | %.2f | %s | %.2f | %.2f | %.1f |\n" % (
            o["kind"], o["class"], o["calls_per_image"], o["us_per_launch"], o["bytes"] / 1e6, o["traffic_bytes"] / 1e6, o["traffic_over_algorithmic"],
            o["executed_f16_flop"] / 1e9, o["bound"], o["bound_us"], o["frac_of_bound"], o["us_per_launch"] * o["calls_per_image"]))
    for kind in ("1x1", "3x3"):
        sel = [o for o in out if o["kind"] == kind]
        f.write("\n%s: %.3f ms / image, sum of per-class bounds %.3f ms (%.2f of the time)\n" % (
            kind, sum(o["us_per_launch"] * o["calls_per_image"] for o in sel) / 1e3, sum(o["bound_us"] * o["calls_per_image"] for o in sel) / 1e3,
            sum(o["bound_us"] * o["calls_per_image"] for o in sel) / sum(o["us_per_launch"] * o["calls_per_image"] for o in sel)))
print(open(sys.argv[6]).read())
