"""Per-image kernel time in the steady state of `bench.py` (conv net + hot path) from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o full -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline
    python tools/steady_state.py DIR/full_kernel_trace.csv [images] [first]
Images are delimited by K1 (pod_mc_merge_score's kernel: the first hot-path kernel of an image, right after its conv net).  bench.py runs 2 x streams priming images, W warm-up images,
K timed images, then hot-path-only loops: `first` (default 2*3 + 3 + 1; 2*1 + 3 + 1 = 6 for --streams 1) skips to the timed region and `images` of them are
aggregated (with several streams the kernels of neighbouring images interleave; the aggregate is what matters)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
first = int(sys.argv[3]) if len(sys.argv) > 3 else 10
idx = [i for i, r in enumerate(rows) if any(t in r["Kernel_Name"] for t in ("k1f_merge_score", "k1_prune_stream", "k1_mc_merge_score"))]
# K1 of image j comes AFTER image j's conv net: the span between K1 of image first-1 and K1 of image first-1+n covers n
# hot-path tails + n conv nets
a, b = idx[first - 1], idx[first - 1 + n]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a + 1:b + 1]:
    k = r["Kernel_Name"].replace("void ", "")[:96]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 / n
tot = sum(v[1] for v in agg.values()) / n
print("steady state, %d images of the timed region: %.2f ms per image wall, %.2f ms of kernel time" % (n, span / 1e3, tot / 1e3))
print("| ms/image | calls/image | avg us | kernel |\n|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("| %.3f | %d | %.1f | %s |" % (v[1] / 1e3 / n, v[0] // n, v[1] / v[0], k))
