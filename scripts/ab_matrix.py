#!/usr/bin/env python3
"""Same-process interleaved A/B over the combinations of several module-level switches (scripts/ab_bench.py for more than one):
    python scripts/ab_matrix.py fatezero_amd.issue:ENABLED fatezero_amd.video_diffusion.models.unet_3d_condition:TIME_EMBED_CACHE
runs bench.run_job for every True / False combination, `--rounds` times round-robin (first round dropped), and prints s/job per combination."""
import importlib
import itertools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    specs = [a for a in sys.argv[1:] if ":" in a]
    rounds = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--rounds=")), 3))
    sw = [(importlib.import_module(m), n) for m, n in (s.split(":") for s in specs)]
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev)
    z0 = torch.randn(1, 4, 8, 64, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
    combos = list(itertools.product((True, False), repeat=len(sw)))
    times = {c: [] for c in combos}
    for rnd in range(rounds):
        for c in combos:
            for (mod, name), v in zip(sw, c):
                setattr(mod, name, v)
            torch.cuda.synchronize()
            t0 = time.time()
            bench.run_job(pipe, z0, 50, dev)
            torch.cuda.synchronize()
            if rnd > 0:
                times[c].append(time.time() - t0)
    for c in combos:
        label = " ".join(f"{n}={v}" for (_, n), v in zip(sw, c))
        print(f"{label}: " + " ".join(f"{t:.3f}" for t in times[c]) + f"  s/job (min {min(times[c]):.3f})")


main()
