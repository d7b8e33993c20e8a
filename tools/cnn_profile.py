"""Aggregates a rocprofv3 kernel trace of cnn-only steps: per-kernel time per image in the steady state
(skips the first `skip` seconds where MIOpen's find phase runs)."""
import csv, sys, collections
path, n_img = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
t = sorted(int(r["Start_Timestamp"]) for r in rows)
t_end = max(int(r["End_Timestamp"]) for r in rows)
# steady state = last n_img "marker" intervals: markers are the copy of the uint8 frame? use time: last X ms given by argv[3]
win = float(sys.argv[3]) * 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r["Start_Timestamp"]) >= t_end - win:
        a = agg[r["Kernel_Name"][:90]]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("window %.1f ms, kernel time %.1f ms, per image %.2f ms" % (win / 1e6, tot / 1e3, tot / 1e3 / n_img))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%8.2f ms/img %6d calls/img  avg %8.1f us  %s" % (v[1] / 1e3 / n_img, v[0] // n_img, v[1] / v[0], k))
