#!/usr/bin/env python3
"""Same-process interleaved A/B of the two 320 x 128 tile forms on the launches the 8-wave one carries today: 254122 = 2 x 4 waves of
5 x 1 MFMA tiles (five MFMAs per six fragment reads per k sub-step), 522222 = 5 x 2 waves of 2 x 2 tiles (ten waves: four MFMAs per
four fragment reads).  Median of interleaved rounds; TF/s at 2 K N per row."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K

dev = "cuda"


def bench_pair(fa, fb, rounds=9, rep=5):
    ta, tb = [], []
    for r in range(rounds + 1):
        for fn, acc in ((fa, ta), (fb, tb)):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(rep):
                fn()
            en.record()
            torch.cuda.synchronize()
            if r:
                acc.append(st.elapsed_time(en) / rep * 1e3)
    ta.sort()
    tb.sort()
    return ta[len(ta) // 2], tb[len(tb) // 2]


res = {}
for (n, hw, cin, cout) in [(8, 64, 320, 320), (8, 64, 640, 320), (8, 64, 960, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (16, 32, 640, 640)]:
    x = torch.randn(n, hw * hw, cin).half().to(dev)
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    b = torch.zeros(cout).half().to(dev)
    sk = 1 if hw == 64 else 2
    a, c = bench_pair(lambda: K.conv3x3(x, wt, b, hw=(hw, hw), tile_cfg=254122, split_k=sk), lambda: K.conv3x3(x, wt, b, hw=(hw, hw), tile_cfg=522222, split_k=sk))
    fl = 2.0 * n * hw * hw * cout * cin * 9
    res[f"conv n{n} {hw}^2 {cin}->{cout} splitk{sk}"] = (a, c, fl / a / 1e6, fl / c / 1e6)
for (rows, k, o, nres) in [(32768, 320, 320, 1), (32768, 320, 640, 0), (32768, 320, 960, 0), (32768, 1280, 320, 1), (8192, 640, 640, 1), (8192, 2560, 640, 1)]:
    x = torch.randn(rows, k).half().to(dev)
    w = (torch.randn(o, k) * 0.02).half().to(dev)
    r = torch.randn(rows, o).half().to(dev) if nres else None
    a, c = bench_pair(lambda: K.gemm(x, w, None, res=r, tile_cfg=254122, split_k=1), lambda: K.gemm(x, w, None, res=r, tile_cfg=522222, split_k=1))
    fl = 2.0 * rows * k * o
    res[f"gemm {rows}x{k}->{o} res{nres}"] = (a, c, fl / a / 1e6, fl / c / 1e6)
for k_, (a, c, ta, tc) in res.items():
    print(f"{k_:36s} 8-wave 5x1 {a:8.1f} us {ta:7.1f} TF/s   10-wave 2x2 {c:8.1f} us {tc:7.1f} TF/s   {a / c:5.2f}x")
