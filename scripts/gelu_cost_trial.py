#!/usr/bin/env python3
"""Upper bound on what a cheaper GELU could buy the GEGLU projections: the shipped library against a trial build whose fz_gelu_erf is the
identity (scripts/build_variant.sh build_tmp/libfz_gelu_identity.so -DFZ_GELU_TRIAL_IDENTITY), interleaved, buffers cycling."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import _native as N
from fatezero_amd import kernels as K

dev = "cuda"
libs = {"shipped": N.lib(), "identity": N._open(os.path.abspath("build_tmp/libfz_gelu_identity.so"))}
POOL = 6
for (rows, k, o) in [(32768, 320, 2560), (65536, 320, 2560), (8192, 640, 5120), (16384, 640, 5120), (2048, 1280, 10240), (4096, 1280, 10240)]:
    xs = [torch.randn(rows, k, device=dev).half() for _ in range(POOL)]
    w = (torch.randn(o, k, device=dev) * 0.02).half()
    b = torch.zeros(o, device=dev).half()
    wp, bp = K.pack_geglu(w, b)
    ev = {n: [] for n in libs}
    for i in range(45):
        for n, l in libs.items():
            N._lib = l
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            K.gemm(xs[i % POOL], wp, bp, geglu=True)
            e.record()
            if i >= 5:
                ev[n].append((s, e))
    torch.cuda.synchronize()
    med = {n: sorted(s.elapsed_time(e) * 1e3 for s, e in v)[len(v) // 2] for n, v in ev.items()}
    print(f"GEGLU {rows:6d} x {k:4d} -> {o:5d}: shipped {med['shipped']:7.1f} us   gelu = identity {med['identity']:7.1f} us   ({100 * (1 - med['identity'] / med['shipped']):.1f} % of the launch)")
N._lib = libs["shipped"]
