"""Per-image kernel time in the steady state of `bench.py` (conv net + hot path) from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o full -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline
    python tools/steady_state.py DIR/full_kernel_trace.csv [images]
Images are delimited by k7_finalize; the last `images` of them are aggregated (MIOpen's find pass is long over by then)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
idx = [i for i, r in enumerate(rows) if "k7_finalize" in r["Kernel_Name"]]
a, b = idx[-n - 1], idx[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a + 1:b + 1]:
    k = r["Kernel_Name"].replace("void ", "")[:96]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 / n
tot = sum(v[1] for v in agg.values()) / n
print("steady state, last %d images: %.2f ms per image wall, %.2f ms of kernel time" % (n, span / 1e3, tot / 1e3))
print("| ms/image | calls/image | avg us | kernel |\n|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("| %.3f | %d | %.1f | %s |" % (v[1] / 1e3 / n, v[0] // n, v[1] / v[0], k))
