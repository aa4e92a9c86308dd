"""Idle time between kernels on the GPU timeline from a rocprofv3 --kernel-trace CSV (last N ms of the run)."""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps start with two augment_params launches; analyse timed steps only: skip the first SKIP steps (warm-up,
# autotuning) and stop before the instrumented, per-kernel-synchronised steps bench.py appends (STEPS steps analysed)
SKIP = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
starts = [s for s, e, n in rows if n.startswith("augment_params_kernel")][0::2]
lo = starts[SKIP]
hi = starts[SKIP + STEPS] if len(starts) > SKIP + STEPS else rows[-1][1]
rows = [r for r in rows if lo <= r[0] < hi]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if s1 > e0:
        gaps.append((s1 - e0, n0.split("(")[0][:50], n1.split("(")[0][:50]))
print("kernels %d over %d steps  span %.2f ms/step  busy %.2f ms/step  idle %.2f ms/step (%.1f %%)" % (len(rows), STEPS, span / 1e6 / STEPS, busy / 1e6 / STEPS, (span - busy) / 1e6 / STEPS, 100.0 * (span - busy) / span))
import collections
by = collections.Counter()
cnt = collections.Counter()
for g, a, b in gaps:
    by[b] += g
    cnt[b] += 1
print("idle before kernel (top):")
for k, v in by.most_common(12):
    print("  %-52s %7.3f ms over %4d gaps (%.1f us avg)" % (k, v / 1e6, cnt[k], v / cnt[k] / 1e3))
tot = collections.Counter()
num = collections.Counter()
for s_, e_, n_ in rows:
    k = n_.split("(")[0]
    k = k.replace("void ", "")[:60]
    tot[k] += e_ - s_
    num[k] += 1
print("busy time by kernel (per step):")
for k, v in tot.most_common(40):
    print("  %-62s %8.3f ms  %5.1f launches" % (k, v / 1e6 / STEPS, num[k] / STEPS))
