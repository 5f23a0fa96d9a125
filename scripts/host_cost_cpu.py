#!/usr/bin/env python3
"""Pure host cost of one UNet forward (Python orchestration + ctypes marshalling, kernels stubbed out): runs anywhere.
    python scripts/host_cost_cpu.py [--profile]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import _native as N, build
import ctypes as C

real = N._open(build.build_emu())


class Stub:
    """Same ctypes marshalling as the real library (argtypes conversions included), kernels replaced by a C no-op."""
    def __init__(self):
        libc = C.CDLL(None)
        for name, (res, args) in N._SIGS.items():
            if name in ("fz_groupnorm_chunks", "fz_gemm_workspace_floats", "fz_version"):
                setattr(self, name, getattr(real, name))
                continue
            fn = C.CFUNCTYPE(res, *args)(("getpid", libc))  # cheap C function, ignores its arguments
            setattr(self, name, fn)


N._lib, N._lib_path, N._is_test_backend = Stub(), "stub", True
import bench
from fatezero_amd.video_diffusion.models.resnet import Tokens
from fatezero_amd.video_diffusion.models import UNetPseudo3DConditionModel
from fatezero_amd.video_diffusion.prompt_attention import attention_util

torch.manual_seed(0)
unet = UNetPseudo3DConditionModel(**bench.SD15, lora=160).half().eval()
for m in unet.modules():  # tuned-checkpoint-like: temporal branches alive (as bench.py's init_like_tuned_checkpoint)
    if hasattr(m, "up") and hasattr(m, "down"):
        torch.nn.init.normal_(m.up.weight, std=0.01)
N.check = lambda rc, what: None
ctx = torch.randn(1, 77, 768).half()
x = torch.randn(8, 64, 4).half()
tok = Tokens(x, 1, 8, 8, 8)


class P:
    pass


p = P(); p.unet = unet
store = attention_util.AttentionStore()
store.LOW_RESOURCE = True
attention_util.register_attention_control(p, store)
for _ in range(2):
    unet.forward_tokens(tok, 500, ctx); store.step_callback(x)
n = 5
t0 = time.perf_counter()
for _ in range(n):
    unet.forward_tokens(tok, 500, ctx); store.step_callback(x)
dt = (time.perf_counter() - t0) / n
print(f"host cost of one inversion-mode forward (8 frames, capture on): {dt * 1e3:.2f} ms")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        unet.forward_tokens(tok, 500, ctx); store.step_callback(x)
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(25)
