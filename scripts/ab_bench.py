#!/usr/bin/env python3
"""Same-box A/B of a module-level switch: runs bench.run_job with the switch on / off, interleaved, and prints job times.
    python scripts/ab_bench.py fatezero_amd.video_diffusion.models.attention LN_FUSION"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    mod, name = importlib.import_module(sys.argv[1]), sys.argv[2]
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev)
    z0 = torch.randn(1, 4, 8, 64, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
    times = {True: [], False: []}
    for rnd in range(3):
        for val in (True, False):
            setattr(mod, name, val)
            torch.cuda.synchronize()
            t0 = time.time()
            bench.run_job(pipe, z0, 50, dev)
            torch.cuda.synchronize()
            if rnd > 0:
                times[val].append(time.time() - t0)
    for val in (True, False):
        print(f"{name}={val}: " + " ".join(f"{t:.3f}" for t in times[val]) + f"  s/job (min {min(times[val]):.3f})")


main()
