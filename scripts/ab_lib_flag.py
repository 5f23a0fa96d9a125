#!/usr/bin/env python3
"""Same-process A/B of an `int` switch INSIDE a trial build of the kernel library (symbols like fz_igemm_trial_no_kg2, only present with
-DFZ_IGEMM_TRIALS): whole bench jobs with the switch 0 / 1, interleaved.
    FZ_TRIAL_LIB=build_tmp/libfz_trials.so python scripts/ab_lib_flag.py fz_igemm_trial_no_kg2 [rounds] [frames]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import _native
_native.use_test_backend(os.path.abspath(os.environ["FZ_TRIAL_LIB"]))
_native._is_test_backend = False
import bench  # noqa: E402

flag = ctypes.c_int.in_dll(_native.lib(), sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
z0 = torch.randn(1, 4, frames, 64, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
times = {0: [], 1: []}
for rnd in range(rounds + 1):
    for val in (0, 1):
        flag.value = val
        torch.cuda.synchronize()
        t0 = time.time()
        bench.run_job(pipe, z0, 50, dev)
        torch.cuda.synchronize()
        if rnd > 0:
            times[val].append(time.time() - t0)
for val in (0, 1):
    print(f"{sys.argv[1]}={val}: " + " ".join(f"{t:.3f}" for t in times[val]) + f"  s/job (min {min(times[val]):.3f})")
