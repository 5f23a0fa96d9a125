#!/usr/bin/env python3
"""How much of a UNet forward is host (Python / launch) time?  Runs the full-size model on tiny latents (GPU work
negligible -> wall time == host overhead per forward) and on the real 64x64 latents (GPU-bound or not)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fatezero_amd.video_diffusion.models.resnet import Tokens

dev = torch.device("cuda")
pipe = bench.build_pipeline(dev)
unet = pipe.unet
ctx = torch.randn(2, 77, 768, device=dev).half()
for (f, hw, b) in [(8, 8, 1), (8, 8, 2), (8, 64, 1), (8, 64, 2)]:
    x = torch.randn(b * f, hw * hw, 4, device=dev).half()
    tok = Tokens(x, b, f, hw, hw)
    for _ in range(3):
        unet.forward_tokens(tok, 500, ctx[:b])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        unet.forward_tokens(tok, 500, ctx[:b])
    t_host = time.perf_counter() - t0
    e.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    print(f"F={f} latent={hw} B={b}: host-issue {t_host / n * 1e3:.2f} ms/forward, wall {t_wall / n * 1e3:.2f} ms, gpu(events) {s.elapsed_time(e) / n:.2f} ms")
