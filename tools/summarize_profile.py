"""Condenses a tools/profile_round.sh output directory into profiles/<tag>_* (tracked)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = "gpurun_out/prof_%s" % tag
dst = "profiles"
os.makedirs(dst, exist_ok=True)
lines = ["# rocprofv3 summary, round tag %s" % tag, ""]


def short(n):
    n = n.replace("void ", "")
    return (n[:110] + "...") if len(n) > 113 else n


def table(name, title, only_pod=False, top=24):
    f = os.path.join(src, name + "_kernel_stats.csv")
    if not os.path.exists(f):
        return
    rows = [r for r in csv.DictReader(open(f)) if not only_pod or "pod::" in r["Name"]]
    lines.extend(["## " + title, "", "| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"])
    for r in rows[:top]:
        lines.append("| %s | %s | %.2f | %.2f | %.2f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                     float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    lines.append("")
    with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)), "w") as fo:
        w = csv.writer(fo)
        w.writerow(rows[0].keys())
        for r in rows[:40]:
            w.writerow([short(v) if k == "Name" else v for k, v in r.items()])


table("k1_class", "K1 alone, product configuration (2K class channels): `rocprofv3 --kernel-trace --stats -- python tools/k1_only.py 120`", True)
table("k1_dense", "K1 alone, dense merge of all 2K+4+D channels: `K1_DENSE=1 rocprofv3 --kernel-trace --stats -- python tools/k1_only.py 120`", True)
table("k1_fused", "K1f alone (merge + score in one launch, the product form since round 4): `K1_FUSED=1 rocprofv3 --kernel-trace --stats -- python tools/k1_only.py 120`", True)
ev = os.path.join(src, "k1_events.txt")
if os.path.exists(ev):
    lines += ["K1 alone, HIP events per launch (`python tools/k1_only.py 120`, class channels then `K1_DENSE=1`):", "", "```"] + \
             [l.rstrip() for l in open(ev) if "events" in l or "counts" in l] + ["```", ""]


def k1_counters(prefix):
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(src, "pmc_%s%s" % (prefix, c), "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if any(t in r["Kernel_Name"] for t in ("k1f_merge_score", "k1_prune_stream", "k1_mc_merge_score")):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pm[k] = (sum(v) / len(v), min(v), max(v), len(v))
    return pm


traffic = {"workload": {"anchors_R": 193374, "mc_runs": 10, "config": "cfg3", "synthetic_mode": "planted"},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/k1_only.py 12; FETCH_SIZE x2 (gfx950)"}
for prefix, key, what in (("", "k1_class_traffic_bytes", "product path: K1 streams the 2K class channels"),
                          ("dense_", "k1_dense_traffic_bytes", "K1_DENSE=1: dense merge of all 2K+4+D channels"),
                          ("fused_", "k1_fused_traffic_bytes", "K1_FUSED=1: pod_merge_score_fused (k1f_merge_score), merge + score in one launch, planes not "
                                                               "stored; 4-byte loads per lane: the x2 correction of FETCH_SIZE is calibrated for 16-byte "
                                                               "loads only, so the figure is an upper bound here")):
    pm = k1_counters(prefix)
    if not pm:
        continue
    lines += ["## PMC counters of K1 / K1f, %s (separate `rocprofv3 --pmc` passes, `python tools/k1_only.py 12`)" % what, "",
              "| counter | mean per launch | min | max | launches |", "|---|---|---|---|---|"]
    for k, (m, lo, hi, n) in sorted(pm.items()):
        lines.append("| %s | %.6g | %.6g | %.6g | %d |" % (k, m, lo, hi, n))
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        fetch_b, write_b = pm["FETCH_SIZE"][0] * 1024, pm["WRITE_SIZE"][0] * 1024
        lines += ["", "HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 128-B requests at 64 B on",
                  "gfx950 for wide coalesced reads: x2; WRITE_SIZE taken as reported, KB units):",
                  "", "    reads  = 2 * FETCH_SIZE * 1024 = %.1f MB" % (2 * fetch_b / 1e6),
                  "    writes =     WRITE_SIZE * 1024 = %.1f MB" % (write_b / 1e6),
                  "    traffic = %.0f bytes" % (2 * fetch_b + write_b), ""]
        traffic[key] = 2 * fetch_b + write_b
        traffic[key.replace("traffic_bytes", "fetch_bytes_corrected")] = 2 * fetch_b
        traffic[key.replace("traffic_bytes", "write_bytes")] = write_b
if len(traffic) > 2:
    json.dump(traffic, open(os.path.join(dst, "%s_k1_traffic.json" % tag), "w"))

for synth in ("planted", "worst"):
    table("hot_" + synth, "hot path alone, one stream, %s inputs: `rocprofv3 --kernel-trace --stats -- python bench.py --no-cnn --streams 1 "
          "--steps 100 --warmup 10 --synth %s --no-cpu-baseline --no-diagnostics`" % (synth, synth), True)
    tl = os.path.join(src, "hot_%s_timeline.txt" % synth)
    if os.path.exists(tl):
        body = [l.rstrip() for l in open(tl)]
        starts = [i for i, l in enumerate(body) if l.startswith("image")]
        if starts:
            # the trace's time stamps jitter by a microsecond or two between kernels on some boxes: of the images printed,
            # show the one whose kernels line up best (smallest sum of |gap|)
            blocks = [body[a:b] for a, b in zip(starts, starts[1:] + [len(body)])]
            def jitter(block):
                return sum(abs(float(l.split("gap")[1].split()[0])) for l in block if " gap " in l)
            best = min(blocks, key=jitter)
            lines += ["one image of that trace (start offset / duration / gap to the previous kernel, us; `tools/trace_timeline.py`):", "", "```"] + \
                     best + ["```", ""]
            open(os.path.join(dst, "%s_hot_%s_timeline.txt" % (tag, synth)), "w").write("\n".join(best) + "\n")
ss = os.path.join(src, "steady_state.txt")
if os.path.exists(ss):
    txt = [l.rstrip() for l in open(ss) if "amdgpu" not in l]
    lines += ["## steady state per image, conv net + hot path, one stream (`bench.py --steps 10 --warmup 3 --streams 1`, `tools/steady_state.py`)", ""] + txt + [""]
    open(os.path.join(dst, "%s_steady_state.txt" % tag), "w").write("\n".join(txt) + "\n")
bench = []
for j in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    txt = [l for l in open(j) if l.startswith("{")]
    if txt:
        bench.append(json.dumps({"file": os.path.basename(j), "line": json.loads(txt[-1])}))
if bench:
    open(os.path.join(dst, "%s_bench_lines.jsonl" % tag), "w").write("\n".join(bench) + "\n")
    lines += ["## bench lines (`profiles/%s_bench_lines.jsonl`)" % tag, "",
              "| file | n_gpus | value | unit | ms/step | hot path ms | worst ms | K1 frac | conv net TFLOP/s (direct-equivalent) |", "|---|---|---|---|---|---|---|---|---|"]
    for b in bench:
        d = json.loads(b)
        l = d["line"]
        f = lambda x: "" if x is None else ("%.4g" % x)
        lines.append("| %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (d["file"], l.get("n_gpus"), f(l.get("value")), l.get("unit"), f(l.get("ms_per_step")),
                     f(l.get("hot_path_ms_per_image")), f(l.get("hot_path_worst_ms")), f((l.get("roofline_k1") or l.get("roofline") or {}).get("frac")),
                     f((l.get("conv_census") or l.get("conv_roofline") or {}).get("direct_equivalent_tflops"))))
    lines.append("")
# ---- pod_wino_conv3x3 (tools/profile_wino.sh <tag>w) and pod_wino_conv3x3_split (POD_WINO_SPLIT=1 tools/profile_wino.sh <tag>ws)
import re
for suffix, split in (("w", False), ("ws", True)):
    wsrc = "gpurun_out/%s%s" % (tag, suffix)
    if not os.path.exists(os.path.join(wsrc, "wino_stats.csv")):
        continue
    name = "pod_wino_conv3x3_split (fp32 products from 2-way f16 splits of the scaled operands, f16 matrix cores)" if split else "pod_wino_conv3x3 (fp32 matrix cores)"
    lines += ["## %s, the launch bench.py times (`tools/profile_wino.sh`: 19 runs x 5 FPN levels of the 768x1344 frame, C = K = 256)" % name, "",
              "`%srocprofv3 --kernel-trace --stats -- python tools/wino_only.py 20 19 bench`:" % ("POD_WINO_SPLIT=1 " if split else ""), "",
              "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
    rows = [r for r in csv.DictReader(open(os.path.join(wsrc, "wino_stats.csv"))) if "k_wino_conv3x3" in r["Name"]]
    for r in rows[:1]:
        lines.append("| %s | %s | %.1f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    ev_txt = open(os.path.join(wsrc, "wino_events.txt")).read().strip()
    lines += ["", "HIP events, same script without the profiler: `%s`" % ev_txt, ""]
    pm = {}
    for l in open(os.path.join(wsrc, "wino_pmc.txt")):
        k, v, n = l.split()
        pm[k] = float(v)
    lines += ["PMC counters per launch (separate `rocprofv3 --pmc` passes):", "", "| counter | mean per launch |", "|---|---|"]
    lines += ["| %s | %.6g |" % (k, v) for k, v in sorted(pm.items())]
    avg_ns = float(rows[0]["AverageNs"])
    m = re.search(r"\((\d+) tiles, (\d+) with block padding\)", ev_txt)
    tiles, padded = (int(m.group(1)), int(m.group(2))) if m else (51243, 51968)                      # 2 x 4 output tiles of the bench launch
    fetch, write = 2 * pm.get("FETCH_SIZE", 0) * 1024, pm.get("WRITE_SIZE", 0) * 1024
    mfma_busy = pm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (pm.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024)
    products, peak, unit = (3, 2500.0, "f16") if split else (1, 157.3, "fp32")
    flop = 2 * 24 * tiles * 65536 * products
    lines += ["", "Derived: executed MFMA FLOPs of the real 2x4 tiles %s2*24*%d*256*256 = %.1f GFLOP -> %.1f TFLOP/s = %.2f of the %.1f TFLOP/s %s MFMA peak;" % (
                  "3 partial products x " if split else "", tiles, flop / 1e9, flop / avg_ns / 1e3, flop / avg_ns / 1e3 / peak, peak, unit),
              "matrix pipe busy SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = %.2f (it also works on the %d padding tiles of partial 16x16 blocks);" % (mfma_busy, padded - tiles),
              "LDS bank conflicts SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.2f;" % (pm.get("SQ_LDS_BANK_CONFLICT", 0) / max(pm.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
              "direct-convolution rate 2*9*pixels*256*256 / t = %.1f TFLOP/s.  HBM-side traffic (FETCH_SIZE x 2 on gfx950, KB units) %.2f GB read + %.2f GB written" % (
                  2 * 9 * 19 * 21486 * 65536 / avg_ns / 1e3, fetch / 1e9, write / 1e9),
              "per launch = %.2f TB/s: the activations (0.41 GB) are read once per PAIR of 64-channel filter slices (an XCD holds two of the four slices in" % ((fetch + write) / avg_ns / 1e3),
              "its L2 and takes a block with both, back to back) plus the 18x18 / 16x16 halo; far below the HBM roof.", ""]
    json.dump({"kernel": "pod_wino_conv3x3_split" if split else "pod_wino_conv3x3", "levels": [[96, 168], [48, 84], [24, 42], [12, 21], [6, 11]], "copies": 19,
               "traffic_bytes": fetch + write, "fetch_bytes_corrected": fetch,
               "write_bytes": write, "avg_launch_ns_rocprof": avg_ns, "mfma_busy_frac": mfma_busy,
               "source": "tools/profile_wino.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/wino_only.py 2 19 bench; FETCH_SIZE x2 (gfx950)"},
              open(os.path.join(dst, "%s_wino%s_traffic.json" % (tag, "_split" if split else "")), "w"))
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:80]))
