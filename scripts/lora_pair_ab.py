#!/usr/bin/env python3
"""Kernel-level A/B of the one-launch temporal LoRA pair (fz_lora_pair) against fz_temporal_conv3 twice, on the launch shapes of the UNet
(8 and 16 frames, CFG batch 2 -> n = 16 / 32 frames in a launch; every resnet level).  Prints one JSON object {shape: {pair_us, two_us}}."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from scripts.xcd_ks_ab import timeit
dev = "cuda"
res = {}
for clip in (8, 16):
    for (tok, c) in [(4096, 320), (1024, 320), (1024, 640), (256, 640), (256, 1280), (64, 1280), (4096, 640), (1024, 1280)]:
        for batch in (1, 2):
            n = batch * clip
            if not K.lora_pair_ok(n, tok, c, 160, clip):
                continue
            x = torch.randn(n, tok, c).half().to(dev)
            wd = (torch.randn(160, 3, c) * 0.02).half().to(dev)
            wu = (torch.randn(c, 3, 160) * 0.02).half().to(dev)
            r2 = torch.randn(n, tok, c).half().to(dev)
            temb = torch.randn(batch, c).half().to(dev)
            y = torch.empty_like(x)
            d = torch.empty(n, tok, 160, dtype=torch.float16, device=dev)

            def two():
                K.temporal_conv3(x, wd, clip_len=clip, out=d)
                K.temporal_conv3(d, wu, clip_len=clip, res=x, res2=r2, temb=temb, out=y)

            a = timeit(lambda: K.lora_pair(x, wd, wu, clip_len=clip, res2=r2, temb=temb, out=y))
            b = timeit(two)
            res[f"clip{clip} n{n} tok{tok} c{c}"] = {"pair_us": round(a, 1), "two_us": round(b, 1), "ratio": round(b / a, 2)}
            print(f"clip{clip} n{n:3d} tok{tok:5d} c{c:5d}  pair {a:7.1f} us   two launches {b:7.1f} us   {b / a:.2f}x", file=sys.stderr)
print(json.dumps(res))
