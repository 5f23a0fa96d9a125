#!/usr/bin/env python3
"""Kernel-level cost of the GroupNorm partials in fz_lora_pair's epilogue: pair + three-kernel GroupNorm against pair_gn + GroupNorm from the
partials, on the 64^2 launch shapes.  stderr: a table; stdout: JSON."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from scripts.xcd_ks_ab import timeit
dev = "cuda"
res = {}
for (clip, batch, tok, c) in [(8, 1, 4096, 320), (8, 2, 4096, 320), (16, 1, 4096, 320), (16, 2, 4096, 320), (8, 1, 4096, 640), (8, 2, 4096, 640)]:
    n = batch * clip
    x = torch.randn(n, tok, c).half().to(dev)
    wd = (torch.randn(160, 3, c) * 0.02).half().to(dev)
    wu = (torch.randn(c, 3, 160) * 0.02).half().to(dev)
    r2 = torch.randn(n, tok, c).half().to(dev)
    temb = torch.randn(batch, c).half().to(dev)
    gam, bet = torch.ones(c).half().to(dev), torch.zeros(c).half().to(dev)
    y = torch.empty_like(x)
    _, part = K.lora_pair(x, wd, wu, clip_len=clip, res2=r2, temb=temb, out=y, gn_groups=32)
    t = {"pair": timeit(lambda: K.lora_pair(x, wd, wu, clip_len=clip, res2=r2, temb=temb, out=y)),
         "pair_gn": timeit(lambda: K.lora_pair(x, wd, wu, clip_len=clip, res2=r2, temb=temb, out=y, gn_groups=32)),
         "groupnorm": timeit(lambda: K.groupnorm(y, gam, bet, span=clip, groups=32, eps=1e-5, silu=True)),
         "from_partial": timeit(lambda: K.groupnorm_from_partial(y, gam, bet, part, span=clip, groups=32, eps=1e-5, silu=True))}
    res[f"clip{clip} n{n} tok{tok} c{c}"] = {k: round(v, 1) for k, v in t.items()}
    print(f"clip{clip} n{n:3d} tok{tok} c{c:4d}  pair {t['pair']:6.1f} + groupnorm {t['groupnorm']:6.1f} = {t['pair'] + t['groupnorm']:6.1f} us   "
          f"pair_gn {t['pair_gn']:6.1f} + from_partial {t['from_partial']:6.1f} = {t['pair_gn'] + t['from_partial']:6.1f} us", file=sys.stderr)
print(json.dumps(res))
