#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math",
                      "-fno-finite-math-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        if cur: print(cur)
        cur = {"name": subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()[:90]}
        continue
    m = re.search(r"(TotalSGPRs|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m: cur[m.group(1).split(" [")[0]] = int(m.group(2))
if cur: print(cur)
