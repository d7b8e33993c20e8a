"""Prints the per-image kernel timeline of the hot path from a rocprofv3 --kernel-trace csv:
    python tools/trace_timeline.py gpurun_out/<dir>/*_kernel_trace.csv
For each kernel of the last full image: start offset, duration, gap to the previous kernel's end (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# an image starts at k_reset_counters / the K1 launch; take the last 3 complete images
starts = [i for i, n in enumerate(names) if any(t in n for t in ("k1f_merge_score", "k1_prune_stream", "k1_mc_merge_score"))]
if len(starts) < 4:
    sys.exit("not enough images in the trace")
for img in range(len(starts) - 4, len(starts) - 1):
    a, b = starts[img], starts[img + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = None
    print("image", img)
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        print("  %8.2f  dur %7.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:70]))
        prev_end = e
    print("  image span %.2f us" % ((prev_end - t0) / 1e3))
