#!/usr/bin/env python3
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fatezero_amd.video_diffusion.models.resnet import Tokens
dev = torch.device("cuda")
pipe = bench.build_pipeline(dev)
unet = pipe.unet
ctx = torch.randn(1, 77, 768, device=dev).half()
x = torch.randn(8, 64, 4, device=dev).half()
tok = Tokens(x, 1, 8, 8, 8)
for _ in range(3):
    unet.forward_tokens(tok, 500, ctx)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    unet.forward_tokens(tok, 500, ctx)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
