#!/usr/bin/env python3
"""A/B of the 160 x 128 four-wave tile (two workgroups per CU) on the rank-160 down projection of the temporal LoRA pair: run once with
FZ_IGEMM_NO_154122=1 (the chooser without it) and once without; plus the forced-tile GEMM proxies in one process."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from scripts.xcd_ks_ab import timeit
dev = "cuda"
res = {}
for (n, tok, cin) in [(8, 4096, 320), (16, 4096, 320), (8, 1024, 640), (16, 1024, 640), (8, 256, 1280), (16, 256, 1280), (8, 4096, 640)]:
    x = torch.randn(n, tok, cin).half().to(dev)
    w = (torch.randn(160, 3, cin) * 0.02).half().to(dev)
    res[f"tconv down n{n} tok{tok} {cin}->160 (chooser)"] = timeit(lambda: K.temporal_conv3(x, w, clip_len=8))
for (rows, k) in [(32768, 960), (65536, 960), (8192, 1920)]:
    x = torch.randn(rows, k).half().to(dev)
    w = (torch.randn(160, k) * 0.02).half().to(dev)
    for cfg in (158122, 154122, 212222):
        res[f"gemm proxy {rows}x{k}->160 tile {cfg}"] = timeit(lambda: K.gemm(x, w, None, tile_cfg=cfg, split_k=1))
print(json.dumps({"with_154122": os.environ.get("FZ_IGEMM_NO_154122") is None, "us": {k: round(v, 2) for k, v in res.items()}}))
