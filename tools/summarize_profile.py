"""Condenses a tools/profile_round.sh output directory into profiles/<tag>_*.md|csv (tracked)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s" % tag
dst = "profiles"
os.makedirs(dst, exist_ok=True)
lines = ["# rocprofv3 summary, round tag %s" % tag, ""]

def short(n):
    n = n.replace("void ", "")
    return (n[:110] + "...") if len(n) > 113 else n

for name, title in (("hotpath/hp", "`rocprofv3 --kernel-trace --stats -- python bench.py --no-cnn --steps 40 --no-cpu-baseline`"),
                    ("full/full", "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline`")):
    f = glob.glob(os.path.join(src, name + "*kernel_stats.csv"))
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    lines += ["## " + title, "", "| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:22]:
        lines.append("| %s | %s | %.2f | %.2f | %.2f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                     float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    if any("naive_conv" in r["Name"] for r in rows[:22]):
        lines += ["", "(`naive_conv_*` and most of the first rows are MIOpen's find pass on the first images -- every solver is tried once per",
                  "conv shape and stream, including the naive reference kernel -- not the steady state; with 3 streams the durations of",
                  "overlapping kernels also stretch.  The steady-state table below is the per-kernel breakdown.)"]
    lines.append("")
    out = os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name.split("/")[0]))
    with open(out, "w") as fo:
        w = csv.writer(fo)
        w.writerow(rows[0].keys())
        for r in rows[:40]:
            w.writerow([short(v) if k == "Name" else v for k, v in r.items()])
ss = os.path.join(src, "steady_state.txt")
if os.path.exists(ss):
    lines += ["## steady state per image, one stream (`bench.py --steps 10 --warmup 3 --streams 1`, `tools/steady_state.py`)", ""]
    lines += [l.rstrip() for l in open(ss) if "amdgpu" not in l] + [""]
for j in ("bench_hotpath.json", "bench_full.json", "bench_full_streams1.json"):
    p = os.path.join(src, j)
    if os.path.exists(p):
        txt = [l for l in open(p) if l.startswith("{")]
        if txt:
            lines += ["## bench line (%s)" % j, "", "```json", txt[-1].strip(), "```", ""]
def k1_counters(prefix):
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        f = glob.glob(os.path.join(src, "pmc_%s%s" % (prefix, c), "*counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "k1_prune_stream" in r["Kernel_Name"] or "k1_mc_merge_score" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pm[k] = (sum(v) / len(v), min(v), max(v), len(v))
    return pm

traffic = {"workload": {"anchors_R": 193374, "mc_runs": 10, "config": "cfg3", "synthetic_mode": "planted"},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/k1_only.py 12; FETCH_SIZE x2 (gfx950)"}
for prefix, key, what in (("", "k1_class_traffic_bytes", "product path: K1 streams the 2K class channels"),
                          ("dense_", "k1_dense_traffic_bytes", "K1_DENSE=1: dense merge of all 2K+4+D channels")):
    pm = k1_counters(prefix)
    if not pm:
        continue
    lines += ["## PMC counters of K1 (`k1_prune_stream`), %s (separate `rocprofv3 --pmc` passes, `python tools/k1_only.py 12`)" % what, "",
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
ev = os.path.join(src, "k1_events.txt")
if os.path.exists(ev):
    lines += ["## K1 alone, HIP events (`python tools/k1_only.py 60`)", "", "```"] + [l.rstrip() for l in open(ev) if "events" in l or "counts" in l] + ["```", ""]
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
