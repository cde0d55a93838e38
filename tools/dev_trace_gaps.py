# reads a rocprofv3 kernel_trace.csv and prints, for the last step, per-launch durations and gaps of the two kernels
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ms = [i for i, e in enumerate(ev) if "k_knn_bvh_post" in e[2]]
last = ms[-50:]
out = []
for i in last:
    s, e, n = ev[i]
    nxt = ev[i + 1] if i + 1 < len(ev) else None
    prev = ev[i - 1]
    out.append((round((e - s) / 1000, 1), round((s - prev[1]) / 1000, 1), round((nxt[1] - nxt[0]) / 1000, 1) if nxt else None, round((nxt[0] - e) / 1000, 1) if nxt else None))
print("match_us, gap_before_us, next_kernel_us, gap_after_us")
for o in out: print(o)
