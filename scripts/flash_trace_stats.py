#!/usr/bin/env python3
"""Per-launch durations of the judged flash kernel from a rocprofv3 --kernel-trace CSV of a bench run: distribution per launch size
(8-frame inversion launches vs 16-frame edit launches), position inside the UNet forward, and what ran right before it."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
fl = [(i, e - s, g) for i, (s, e, k, g) in enumerate(rows) if "attn_flash_kernelILi40" in k]
by_grid = collections.defaultdict(list)
for i, d, g in fl:
    by_grid[g].append((d, i))
for g, v in sorted(by_grid.items()):
    ds = sorted(d for d, _ in v)
    n = len(ds)
    print(f"grid {g}: {n} launches  min {ds[0]/1e3:.1f}  p10 {ds[n//10]/1e3:.1f}  median {ds[n//2]/1e3:.1f}  p90 {ds[n*9//10]/1e3:.1f}  max {ds[-1]/1e3:.1f} us  mean {sum(ds)/n/1e3:.1f}")
    # which of the 5 flash layers of a forward (position modulo 5 in launch order) is slow?
    pos = collections.defaultdict(list)
    for j, (d, i) in enumerate(sorted(v, key=lambda x: x[1])):
        pos[j % 5].append(d)
    print("   by layer position in the forward (down0.0 down0.1 up3.0 up3.1 up3.2): " + "  ".join(f"{sum(p)/len(p)/1e3:.1f}" for _, p in sorted(pos.items())))
    prev = collections.defaultdict(list)
    for d, i in v:
        prev[rows[i - 1][2].split("(")[0][:50]].append(d)
    for k, p in sorted(prev.items(), key=lambda x: -len(x[1]))[:4]:
        print(f"   after {k}: {len(p)} launches, mean {sum(p)/len(p)/1e3:.1f} us")
