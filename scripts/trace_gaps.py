"""Idle time between kernels on the GPU timeline from a rocprofv3 --kernel-trace CSV (last N ms of the run)."""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# take the last third of the trace (steady-state steps)
t_end = rows[-1][1]
t_begin = rows[0][0]
cut = t_begin + (t_end - t_begin) * 2 // 3
rows = [r for r in rows if r[0] >= cut]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if s1 > e0:
        gaps.append((s1 - e0, n0.split("(")[0][:50], n1.split("(")[0][:50]))
print("kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
import collections
by = collections.Counter()
cnt = collections.Counter()
for g, a, b in gaps:
    by[b] += g
    cnt[b] += 1
print("idle before kernel (top):")
for k, v in by.most_common(12):
    print("  %-52s %7.3f ms over %4d gaps (%.1f us avg)" % (k, v / 1e6, cnt[k], v / cnt[k] / 1e3))
