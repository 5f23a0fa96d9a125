"""A/B of the one-launch FeedForward chain (csrc/ff_chain.hip) against the two launches it replaces (fz_gemm with the GEGLU epilogue +
fz_gemm_lnout) at the 64x64-level shapes of the bench job, plus a bit-equality check.  Operands cycle through a pool larger than the 256 MB
Infinity Cache so that every launch reads its rows from HBM, as in the job.  Every variant is timed as a BATCH of back-to-back launches
between two HIP events (the chain launch goes straight through ctypes on preallocated outputs: ~5 us of host time per call, so the GPU stays
busy), the variants interleaved round by round; median / min per launch.  Trial builds of the kernel (scripts/build_variant.sh
build_tmp/libfz_ff_<name>.so -DFC_TRIAL_...) found under build_tmp/ are timed alongside."""
import glob
import os
import sys
import torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K
from fatezero_amd import _native as N

dev = "cuda"
torch.manual_seed(0)
POOL = 10
BATCH = 10
libs = {"one": N.lib()}
for path in sorted(glob.glob("build_tmp/libfz_ff_*.so")):
    libs[os.path.basename(path)[len("libfz_ff_"):-3]] = N._open(os.path.abspath(path))


def timeit(fns, n=12):
    ev = {k: [] for k in fns}
    for i in range(n + 3):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for r in range(BATCH):
                f(i * BATCH + r)
            e.record()
            if i >= 3:
                ev[k].append((s, e))
    torch.cuda.synchronize()
    out = {}
    for k, v in ev.items():
        t = sorted(s.elapsed_time(e) * 1e3 / BATCH for s, e in v)
        out[k] = (t[len(t) // 2], t[0])
    return out


c, inner = 320, 1280
w1 = (torch.randn(2 * inner, c, device=dev) * c ** -0.5).half()
b1 = (torch.randn(2 * inner, device=dev) * 0.3).half()
w2 = (torch.randn(c, inner, device=dev) * inner ** -0.5).half()
b2 = (torch.randn(c, device=dev) * 0.3).half()
gam = (1 + 0.1 * torch.randn(c, device=dev)).half()
bet = (0.1 * torch.randn(c, device=dev)).half()
ln = (gam, bet, 1e-5)
wp, bp = K.pack_geglu(w1, b1)
packed = K.ff_chain_pack(w1, b1, w2)
flop = lambda rows: rows * (2 * c * 2 * inner + 2 * inner * c)
stream = K._stream(w1)
frames_list = [int(a) for a in sys.argv[1:]] or [2, 4, 6, 8, 16, 24, 32]
for frames in frames_list:
    rows = frames * 4096
    xs = [torch.randn(frames, 4096, c, device=dev).half() for _ in range(POOL)]
    rs = [torch.randn(frames, 4096, c, device=dev).half() for _ in range(POOL)]
    ys = [torch.empty_like(xs[0]) for _ in range(2)]
    y1, l1 = K.ff_chain(xs[0], packed, b2, inner, res=rs[0], ln=ln)
    y0, l0 = K.gemm_lnout(K.gemm(xs[0], wp, bp, geglu=True, split_k=1), w2, b2, ln, res=rs[0], split_k=1)
    same = bool(torch.equal(y1, y0)) and (l0 is None or bool(torch.equal(l1, l0)))

    def direct(lib):
        fn = lib.fz_ff_chain
        args = [(xs[j].data_ptr(), packed.data_ptr(), b2.data_ptr(), rs[j].data_ptr(), ys[0].data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5,
                 ys[1].data_ptr(), rows, c, inner, stream) for j in range(POOL)]
        return lambda i: fn(*args[i % POOL])

    fns = {"two": lambda i: K.gemm_lnout(K.gemm(xs[i % POOL], wp, bp, geglu=True), w2, b2, ln, res=rs[i % POOL])}
    for name, lib in libs.items():
        fns[name] = direct(lib)
    r = timeit(fns)
    t2, t1 = r["two"], r["one"]
    print(f"{frames:2d} frames ({rows:6d} rows, {(rows + 127) // 128:4d} workgroups)  two launches {t2[0]:7.1f} us (min {t2[1]:7.1f})   one launch {t1[0]:7.1f} us "
          f"(min {t1[1]:7.1f})   x{t2[0] / t1[0]:.2f}   {flop(rows) / t1[0] / 1e6:7.1f} TF/s = {flop(rows) / t1[0] / 1e6 / 2500:.3f} of the MFMA roof "
          f"(two: {flop(rows) / t2[0] / 1e6 / 2500:.3f})   bit-identical: {same}", flush=True)
    if "timing" in libs:  # -DFC_TIMING: cycle totals per loop segment of the UP / DOWN wave of pair 0 of workgroup 0
        import ctypes
        buf = (ctypes.c_longlong * 16)()
        libs["timing"].fz_ff_chain_timing.argtypes = [ctypes.c_void_p]
        fns["timing"](0)
        torch.cuda.synchronize()
        libs["timing"].fz_ff_chain_timing(buf)
        v = list(buf)
        print("      cycles UP   wave: wait_vm %d | barrier A %d | DMA issue %d | 40 reads + MFMAs %d | bias %d | barrier B %d | U write %d | loop total %d" % tuple(v[:8]))
        print("      cycles DOWN wave: wait_vm %d | barrier A %d | DMA issue %d | U read + gate %d | 20 reads + MFMAs %d | barrier B %d | - %d | loop total %d" % tuple(v[8:]), flush=True)
    if len(libs) > 1:
        print("      trial builds (median us): " + "  ".join(f"{k} {v[0]:.1f}" for k, v in r.items() if k not in ("two", "one")), flush=True)
