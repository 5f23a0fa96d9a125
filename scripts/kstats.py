#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_stats.csv by kernel family: python scripts/kstats.py <csv> [jobs]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
jobs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6:.1f} ms over {jobs:g} jobs = {tot/1e6/jobs:.1f} ms/job")
grp = {}
for r in rows:
    n = r['Name']
    m = re.search(r'igemm_kernel<([^>]*)>', n)
    if m:
        a = [x.strip() for x in m.group(1).split(',')]
        key = f"igemm {['gemm','conv3x3','tconv','conv3x3'][int(a[6])]}{' geglu' if a[7]=='true' else ''} tile {a[0]}{a[1]}{a[2]}{a[3]}{' K-groups' if len(a) > 9 and a[9].isdigit() and int(a[9]) & 32 else ''}"
    elif 'lora_pair_kernel' in n: key = 'lora_pair_kernel (both temporal LoRA convolutions)'
    elif 'Cijk' in n: key = 'hipBLASLt Cijk'
    elif 'at::native' in n: key = 'torch ' + re.sub(r'.*native::(\(anonymous namespace\)::)?', '', n)[:40]
    else: key = re.sub(r'^_Z\d+', '', n)[:58]
    g = grp.setdefault(key, [0, 0]); g[0] += int(r['TotalDurationNs']); g[1] += int(r['Calls'])
fam = {}
for k, (t, c) in grp.items():
    f = k.split(' tile')[0] if k.startswith('igemm') else None
    if f: fam[f] = fam.get(f, 0) + t
for f, t in sorted(fam.items(), key=lambda x: -x[1]): print(f"   family {f:22s} {t/1e6/jobs:8.1f} ms/job {100*t/tot:5.1f}%")
for k, (t, c) in sorted(grp.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 32]:
    print(f"{t/1e6/jobs:8.1f} ms/job {100*t/tot:5.1f}% calls/job {c/jobs:7.0f} avg {t/c/1e3:8.1f} us  {k}")
