#!/usr/bin/env python3
"""GEGLU projection (fz_gemm, geglu epilogue) on the feed-forward shapes of the UNet at 8 / 16 frames: the library's own tile choice
(column 0) against every tile that can pair (h, gate) columns."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from kbench import timeit

CFGS = [0, 244222, 224223, 222222]
dev = "cuda"
print("geglu: rows K 2*inner | " + " ".join(f"{c:>8d}" for c in CFGS))
for (rows, k, o) in [(32768, 320, 2560), (65536, 320, 2560), (8192, 640, 5120), (16384, 640, 5120), (2048, 1280, 10240), (4096, 1280, 10240),
                     (512, 1280, 10240), (1024, 1280, 10240)]:
    x = torch.randn(rows, k).half().to(dev)
    w = (torch.randn(o, k) * 0.02).half().to(dev)
    b = torch.zeros(o).half().to(dev)
    wp, bp = K.pack_geglu(w, b) if hasattr(K, "pack_geglu") else (w, b)
    flops = 2.0 * rows * k * o
    row = []
    for c in CFGS:
        try:
            ms = timeit(lambda: K.gemm(x, wp, bp, geglu=True, tile_cfg=c), iters=8, warm=2)
            row.append(f"{flops / ms / 1e9:8.0f}")
        except Exception as e:
            row.append("     err")
    print(f"{rows:6d} {k:5d} {o:5d} | " + " ".join(row))
