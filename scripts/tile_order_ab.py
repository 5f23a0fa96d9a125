#!/usr/bin/env python3
"""Same-box A/B of the per-launch tile order (csrc/igemm.hip ig_launch: a-tile fastest vs b-tile fastest inside an XCD's run of tiles):
run once per setting (FZ_IGEMM_NO_TILE_ORDER=1 = always a-fastest, the round-3 behaviour) and compare.  The library's own tile choice."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from scripts.xcd_ks_ab import timeit  # noqa: E402  (prints its own table on import: ignored by the caller)

dev = "cuda"
res = {}
for (rows, k, o, geglu) in [(2048, 1280, 10240, True), (4096, 1280, 10240, True), (512, 1280, 10240, True), (1024, 1280, 10240, True),
                            (8192, 640, 5120, True), (16384, 640, 5120, True), (32768, 320, 2560, True), (65536, 320, 2560, True),
                            (2048, 1280, 3840, False), (4096, 1280, 3840, False), (2048, 1280, 2560, False), (8192, 640, 1920, False),
                            (512, 1280, 3840, False), (32768, 320, 960, False)]:
    x = torch.randn(rows, k).half().to(dev)
    w = (torch.randn(o, k) * 0.02).half().to(dev)
    if geglu:
        w, _ = K.pack_geglu(w, None)
    res[f"gemm{' geglu' if geglu else ''} {rows}x{k}->{o}"] = timeit(lambda: K.gemm(x, w, None, geglu=geglu))
for (n, l, k, c) in [(8, 256, 1280, 1280), (16, 256, 1280, 1280), (8, 64, 1280, 1280), (8, 1024, 640, 640), (8, 4096, 320, 320)]:
    x = torch.randn(n, l, k).half().to(dev)
    w = (torch.randn(3 * c, k) * 0.02).half().to(dev)
    res[f"qkvt n{n} L{l} {k}->3x{c}"] = timeit(lambda: K.gemm_qkvt(x, w, 2 * c))
for (n, hw, cin, cout) in [(8, 64, 320, 320), (16, 64, 320, 320), (8, 64, 960, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (16, 32, 1280, 640)]:
    x = torch.randn(n, hw * hw, cin).half().to(dev)
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    b = torch.zeros(cout).half().to(dev)
    res[f"conv n{n} {hw}^2 {cin}->{cout}"] = timeit(lambda: K.conv3x3(x, wt, b, hw=(hw, hw)))
print(json.dumps({"tile_order": os.environ.get("FZ_IGEMM_NO_TILE_ORDER") is None, "us": {k: round(v, 2) for k, v in res.items()}}))
