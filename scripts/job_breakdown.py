#!/usr/bin/env python3
"""Where the GPU time of the bench job goes, per (op, operand shapes): HIP-event brackets (launch stream) around EVERY kernel-launching
function of fatezero_amd.kernels during a short job (default 4 + 4 DDIM steps of the judged 8-frame clip), aggregated and sorted.
The kernel-stats profile (rocprofv3) says which KERNEL is expensive; this says which LAYER SHAPE is.   python scripts/job_breakdown.py [steps]"""
import os
import sys
import collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fatezero_amd import kernels as K

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NAMES = ["attn_self", "attn_cross", "attn_temporal", "blend_mask", "groupnorm", "groupnorm_cat", "groupnorm_stats", "groupnorm_apply",
         "groupnorm_from_partial", "gemm_gn", "gemm_lnout", "ff_chain", "gemm", "gemm_batched", "gemm_vt", "gemm_qkvt", "conv3x3", "temporal_conv3", "lora_pair",
         "layernorm", "geglu", "softmax_rows", "transpose_pad", "latent_update", "accumulate"]
events = []
depth = [0]


def shape_of(a):
    return tuple(a.shape) if isinstance(a, torch.Tensor) else None


def wrap(name):
    orig = getattr(K, name)

    def f(*a, **k):
        if depth[0] > 0 or not rec[0]:   # (gemm_gn falls back to gemm internally: count the outer call only)
            return orig(*a, **k)
        tag = [name] + [s for s in (shape_of(x) for x in a[:3]) if s is not None]
        for kk in ("mode", "res", "res2", "geglu", "stride", "upsample", "gn_groups", "clip_len", "span", "n_frames"):
            if kk in k and k[kk] is not None and k[kk] is not False:
                tag.append(f"{kk}={'y' if isinstance(k[kk], torch.Tensor) else k[kk]}")
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        depth[0] += 1
        s.record()
        try:
            r = orig(*a, **k)
        finally:
            e.record()
            depth[0] -= 1
        events.append((tuple(tag), s, e))
        return r
    setattr(K, name, f)


rec = [False]
for n in NAMES:
    if hasattr(K, n):
        wrap(n)
K._gemm_unwrapped = K.gemm
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev)
z0 = torch.randn(1, 4, 8, 64, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
bench.run_job(pipe, z0, steps, dev)   # warm
rec[0] = True
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
bench.run_job(pipe, z0, steps, dev)
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for tag, s, e in events:
    d = agg.setdefault(tag, [0, 0.0])
    d[0] += 1
    d[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print(f"job ({steps} + {steps} steps) {t0.elapsed_time(t1):.1f} ms; bracketed {tot:.1f} ms in {len(events)} calls")
byop = collections.defaultdict(float)
for tag, (n, ms) in agg.items():
    byop[tag[0]] += ms
print("by op:", ", ".join(f"{k} {100 * v / tot:.1f}%" for k, v in sorted(byop.items(), key=lambda x: -x[1])))
for tag, (n, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:70]:
    print(f"{100 * ms / tot:5.2f}%  {n:5d} x {1e3 * ms / n:8.1f} us   {' '.join(str(t) for t in tag)}")
