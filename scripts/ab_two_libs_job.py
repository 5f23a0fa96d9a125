#!/usr/bin/env python3
"""Same-process A/B of whole bench jobs between TWO builds of the kernel library (same ABI, same packed formats) loaded side by side: the library the
host code calls is switched between jobs, interleaved.   usage: ab_two_libs_job.py <libA.so> <libB.so> [rounds] [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import _native as N
libs = [(os.path.basename(p), N._open(os.path.abspath(p))) for p in sys.argv[1:3]]
N._lib, N._lib_path = libs[0][1], os.path.abspath(sys.argv[1])
import bench  # noqa: E402

rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
z0 = torch.randn(1, 4, frames, 64, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
times = {name: [] for name, _ in libs}
for rnd in range(rounds + 1):
    for name, L in libs:
        N._lib = L
        torch.cuda.synchronize()
        t0 = time.time()
        bench.run_job(pipe, z0, 50, dev)
        torch.cuda.synchronize()
        if rnd > 0:
            times[name].append(time.time() - t0)
for name, _ in libs:
    print(f"{name}: " + " ".join(f"{t:.3f}" for t in times[name]) + f"  s/job (min {min(times[name]):.3f})")
