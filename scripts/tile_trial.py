#!/usr/bin/env python3
"""Same-box comparison of igemm tile configurations on the conv / GEMM shapes that carry the job (HIP-event timed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import _native, kernels as K
from kbench import timeit

if os.environ.get("FZ_TRIAL_LIB"):  # an alternative build of the kernel library (same ABI) for a same-box A/B
    _native.use_test_backend(os.environ["FZ_TRIAL_LIB"])
    print("library:", os.environ["FZ_TRIAL_LIB"])

CFGS = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,254222,252322,242422,244222".split(","))]
dev = "cuda"
print("conv3x3: frames hw cin cout | " + " ".join(f"{c:>8d}" for c in CFGS))
for (n, hw, cin, cout) in [(8, 64, 320, 320), (16, 64, 320, 320), (8, 64, 640, 320), (8, 64, 960, 320), (8, 32, 640, 640), (16, 32, 640, 640),
                           (8, 32, 1280, 640), (8, 32, 1920, 640), (8, 16, 1280, 1280), (8, 16, 2560, 1280)]:
    x = torch.randn(n, hw * hw, cin).half().to(dev)
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    b = torch.zeros(cout).half().to(dev)
    flops = 2.0 * n * hw * hw * cout * cin * 9
    row = []
    for c in CFGS:
        try:
            ms = timeit(lambda: K.conv3x3(x, wt, b, hw=(hw, hw), tile_cfg=c), iters=8, warm=2)
            row.append(f"{flops / ms / 1e9:8.0f}")
        except Exception as e:
            row.append("     err")
    print(f"{n:3d} {hw:3d} {cin:5d} {cout:5d} | " + " ".join(row))
print("gemm: rows K N | " + " ".join(f"{c:>8d}" for c in CFGS))
for (rows, k, nn) in [(32768, 320, 320), (32768, 320, 1280), (32768, 1280, 320), (8192, 640, 640), (8192, 2560, 640), (8192, 640, 2560), (2048, 1280, 1280),
                      (2048, 5120, 1280), (4096, 1280, 3840)]:
    x = torch.randn(rows, k).half().to(dev)
    w = (torch.randn(nn, k) * 0.02).half().to(dev)
    b = torch.zeros(nn).half().to(dev)
    flops = 2.0 * rows * k * nn
    row = []
    for c in CFGS:
        try:
            ms = timeit(lambda: K.gemm(x, w, b, tile_cfg=c), iters=8, warm=2)
            row.append(f"{flops / ms / 1e9:8.0f}")
        except Exception as e:
            row.append("     err")
    print(f"{rows:6d} {k:5d} {nn:5d} | " + " ".join(row))
print("temporal conv k=3: frames tokens cin cout | TFLOP/s (library's own choice)")
for (n, tokens, cin, cout) in [(8, 4096, 320, 160), (8, 4096, 160, 320), (16, 4096, 320, 160), (16, 4096, 160, 320), (8, 1024, 640, 160),
                                (8, 1024, 160, 640), (16, 256, 1280, 160), (16, 256, 160, 1280)]:
    x = torch.randn(n, tokens, cin).half().to(dev)
    wt = (torch.randn(cout, 3, cin) * 0.03).half().to(dev)
    ms = timeit(lambda: K.temporal_conv3(x, wt, clip_len=8), iters=8, warm=2)
    print(f"{n:3d} {tokens:5d} {cin:5d} {cout:5d} | {2.0 * n * tokens * cin * cout * 3 / ms / 1e9:8.0f}")
