"""rocprofv3 kernel trace (csv) of a ONE-STREAM bench run -> the pod_wino_conv3x3_split launches of the last images grouped by grid size:
calls per image, average duration, rounds of 256 workgroups (the kernel runs one workgroup per CU, so a launch costs whole rounds).
python tools/k12_launches.py <kernel_trace.csv> [images]"""
import collections
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 6
marks = [i for i, r in enumerate(rows) if "k1f_merge_score" in r["Kernel_Name"]]          # one per image: the first hot-path launch
rows = rows[marks[-n_img - 1] + 1:marks[-1] + 1]
agg = collections.OrderedDict()
for r in rows:
    if "k_wino_conv3x3" not in r["Kernel_Name"]:
        continue
    key = (int(r["Grid_Size_X"]) // 256, int(r.get("Grid_Size_Y", 1) or 1))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print("pod_wino_conv3x3_split, %d images of the steady state: %.1f launches, %.3f ms per image" % (n_img, sum(a[0] for a in agg.values()) / n_img, tot / n_img / 1e3))
for (gx, gy), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  workgroups %6d x %d (%5.2f rounds of 256): %4.1f calls/image  avg %8.1f us  %6.3f ms/image (%4.1f %%)" % (
        gx, gy, gx * gy / 256.0, n / n_img, us / n, us / n_img / 1e3, 100 * us / tot))
