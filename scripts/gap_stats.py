#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: total, histogram, and which kernel precedes / follows
the long gaps (host-bound stretches show up as many gaps of tens of microseconds after short kernels)."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last job of the run: split at the largest gaps (job boundaries: arena release / synchronize)
n = len(rows)
print("kernels", n, "span %.3f s" % ((rows[-1][1] - rows[0][0]) / 1e9))
third = rows[n * 2 // 3:] if n > 3000 else rows  # roughly the last (timed) job
busy = sum(e - s for s, e, _ in third)
gaps = []
for (s0, e0, k0), (s1, e1, k1) in zip(third, third[1:]):
    gaps.append((max(0, s1 - e0), k0, k1, e0 - s0))
tot = sum(g[0] for g in gaps)
print("last third: kernels %d busy %.3f s gaps %.3f s (%.1f %% of span)" % (len(third), busy / 1e9, tot / 1e9, 100.0 * tot / (busy + tot)))
edges = [0, 1000, 2000, 3000, 5000, 8000, 12000, 20000, 50000, 200000, 10 ** 12]
for lo, hi in zip(edges, edges[1:]):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print("gap %7.1f-%9.1f us: %6d gaps, %8.3f ms" % (lo / 1e3, hi / 1e3, len(sel), sum(sel) / 1e6))
short = lambda k: k.split("(")[0][:60]
by_prev = collections.defaultdict(lambda: [0, 0])
for g, k0, k1, d0 in gaps:
    by_prev[short(k1)][0] += 1
    by_prev[short(k1)][1] += g
print("gap time by FOLLOWING kernel:")
for k, (c, t) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %8.3f ms %6d x %6.2f us  %s" % (t / 1e6, c, t / c / 1e3, k))
# gaps as a function of the preceding kernel's duration
for lo, hi in [(0, 5000), (5000, 10000), (10000, 20000), (20000, 50000), (50000, 10 ** 12)]:
    sel = [g[0] for g in gaps if lo <= g[3] < hi]
    if sel:
        print("after kernels of %5.0f-%8.0f us: %6d gaps, mean %.2f us" % (lo / 1e3, hi / 1e3, len(sel), sum(sel) / len(sel) / 1e3))
