"""Dev tool: reads a rocprofv3 kernel_trace.csv of a batch run and says how busy the device was: the span of the trace, the time at
least one kernel ran (union), the sum of kernel times (-> average concurrency), the same per queue, and the kernels by total time.
usage: python tools/dev_trace_busy.py <rocprofv3 output dir> [skip_fraction]   (skip_fraction: leading part of the span to ignore, warm-up)"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
t0 = rows[0][0]
def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)
span = t1 - t0
busy = union([(r[0], r[1]) for r in rows]); ksum = sum(r[1] - r[0] for r in rows)
print("span %.3f ms, device busy %.3f ms (%.0f %%), sum of kernel times %.3f ms (concurrency while busy %.2f), %d launches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, ksum / 1e6, ksum / busy, len(rows)))
byq = collections.defaultdict(list)
for r in rows: byq[r[2]].append((r[0], r[1]))
for q, iv in sorted(byq.items()):
    print("  queue %s: %d launches, busy %.3f ms (%.0f %% of the span), mean gap between consecutive launches %.1f us" % (q, len(iv), union(iv) / 1e6, 100.0 * union(iv) / span, (span - union(iv)) / max(len(iv), 1) / 1e3))
byk = collections.defaultdict(lambda: [0, 0])
for r in rows: byk[r[3][:90]][0] += r[1] - r[0]; byk[r[3][:90]][1] += 1
for k, (t, n) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %8.3f ms %6d x %7.1f us  %s" % (t / 1e6, n, t / n / 1e3, k))
