"""A/B of the row-streaming LayerNorm + projection launch (csrc/rowgemm.hip) against the launches it replaces (fz_layernorm + fz_gemm /
fz_gemm_qkvt) at the 64x64-level shapes of the bench job.  Operands cycle through a pool larger than the 256 MB Infinity Cache so that
every launch reads its rows from HBM, as in the job; interleaved (A, B, A, B ...), median of the per-launch HIP-event times."""
import sys
import torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K

dev = "cuda"
torch.manual_seed(0)
POOL = 14


def timeit(fns, n=40):
    ev = {k: [] for k in fns}
    for i in range(n + 5):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            f(i)
            e.record()
            if i >= 5:
                ev[k].append((s, e))
    torch.cuda.synchronize()
    out = {}
    for k, v in ev.items():
        t = sorted(s.elapsed_time(e) * 1e3 for s, e in v)
        out[k] = t[len(t) // 2]
    return out


for frames in (8, 16):
    rows = frames * 4096
    xs = [torch.randn(frames, 4096, 320, device=dev).half() for _ in range(POOL)]
    rs = [torch.randn(frames, 4096, 320, device=dev).half() for _ in range(POOL)]
    gam = (1 + 0.1 * torch.randn(320, device=dev)).half()
    bet = (0.1 * torch.randn(320, device=dev)).half()
    w1 = (torch.randn(320, 320, device=dev) * 320 ** -0.5).half()
    w3 = (torch.randn(960, 320, device=dev) * 320 ** -0.5).half()
    b1 = torch.randn(320, device=dev).half()
    ln = (gam, bet, 1e-5)
    res = {}
    res["LN + to_q (N 320)"] = timeit({
        "two": lambda i: K.gemm(K.layernorm(xs[i % POOL], gam, bet, eps=1e-5), w1),
        "one": lambda i: K.ln_gemm(xs[i % POOL], w1, ln=ln)})
    res["LN + qkv (N 960, plain)"] = timeit({
        "two": lambda i: K.gemm(K.layernorm(xs[i % POOL], gam, bet, eps=1e-5), w3),
        "one": lambda i: K.ln_gemm(xs[i % POOL], w3, ln=ln)})
    res["LN + q|k|Vt (N 960)"] = timeit({
        "two": lambda i: K.gemm_qkvt(K.layernorm(xs[i % POOL], gam, bet, eps=1e-5), w3, 640),
        "one": lambda i: K.ln_gemm_qkvt(xs[i % POOL], w3, 640, ln=ln)})
    res["to_out + bias + res (N 320)"] = timeit({
        "two": lambda i: K.gemm(xs[i % POOL], w1, b1, res=rs[i % POOL]),
        "one": lambda i: K.ln_gemm(xs[i % POOL], w1, b1, res=rs[i % POOL])})
    res["proj_in + bias (N 320)"] = timeit({
        "two": lambda i: K.gemm(xs[i % POOL], w1, b1),
        "one": lambda i: K.ln_gemm(xs[i % POOL], w1, b1)})
    res["LN alone"] = timeit({"two": lambda i: K.layernorm(xs[i % POOL], gam, bet, eps=1e-5), "one": lambda i: None})
    for k, v in res.items():
        print(f"{frames:2d} frames  {k:32s} two launches {v['two']:7.1f} us   one launch {v['one']:7.1f} us   x{v['two'] / max(v['one'], 1e-9):.2f}")
