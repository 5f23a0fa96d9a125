"""A/B of the one-launch cross-attention chain (csrc/xattn_chain.hip) against the launches it replaces at the 64x64-level shapes of the bench
job: plain form = fz_gemm (to_q) + fz_attn_cross + fz_gemm_lnout (to_out + residual + norm3); front form = fz_gemm_lnout (attn1.to_out +
residual + norm2) in front of those.  Operands cycle through a pool larger than the 256 MB Infinity Cache; every variant is timed as a BATCH of
back-to-back launches between two HIP events (the chain launch goes straight through ctypes on preallocated outputs), the variants interleaved
round by round; median / min per launch; bit-equality checked first.     python scripts/xattn_chain_ab.py [frames ...]"""
import ctypes as C
import glob
import os
import sys
import torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K
from fatezero_amd import _native as N

dev = "cuda"
torch.manual_seed(0)
POOL, BATCH = 10, 10
trial = {}  # trial builds of the kernel (scripts/xattn_chain_variants.sh) are timed alongside
for path in sorted(glob.glob("build_tmp/libfz_xc_*.so")):
    trial[os.path.basename(path)[len("libfz_xc_"):-3]] = N._open(os.path.abspath(path))


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


c, heads, lk, clip = 320, 8, 77, 8
scale = 40 ** -0.5
mk = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).half()
wq, wo, wo1 = mk(c, c, k=c ** -0.5 * 2), mk(c, c, k=c ** -0.5), mk(c, c, k=c ** -0.5)
bo, bo1 = mk(c, k=0.3), mk(c, k=0.3)
g1, b1, g2, b2 = (1 + 0.1 * torch.randn(c, device=dev)).half(), mk(c, k=0.1), (1 + 0.1 * torch.randn(c, device=dev)).half(), mk(c, k=0.1)
ln1, ln2 = (g1, b1, 1e-5), (g2, b2, 1e-5)
stream = K._stream(wq)
frames_list = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 24, 32]
for frames in frames_list:
    nb = (frames + clip - 1) // clip
    ctx = mk(nb, lk, 768)
    kk = K.gemm(ctx, mk(c, 768, k=768 ** -0.5 * 2))
    vt = K.gemm_vt(ctx, mk(c, 768, k=768 ** -0.5), K.CROSS_KEYS)
    kvp = K.xattn_chain_kv_pack(kk, vt, lk)
    rows = frames * 4096
    xs = [mk(frames, 4096, c) for _ in range(POOL)]
    rs = [mk(frames, 4096, c, k=1.5) for _ in range(POOL)]
    kw = dict(frames_per_batch=clip, heads=heads, lk=lk, scale=scale)

    def launches(x, r, front):
        if front:
            r, x = K.gemm_lnout(x, wo1, bo1, ln1, res=r)
            x = x if x is not None else K.layernorm(r, g1, b1, eps=1e-5)
        q = K.gemm(x, wq)
        o = torch.empty_like(q)
        K.attn_cross(q, kk, vt, o, clip_len=clip, heads=heads, lk=lk, scale=scale)
        return K.gemm_lnout(o, wo, bo, ln2, res=r)

    for front in (False, True):
        packed = K.xattn_chain_pack(wq, wo, (wo1, bo1, g1, b1) if front else None)
        got = K.xattn_chain(xs[0], packed, kvp, bo, res=rs[0], ln=ln2, front_eps=1e-5 if front else None, **kw)
        y0, l0 = launches(xs[0], rs[0], front)
        same = bool(torch.equal(got[0], y0)) and (l0 is None or bool(torch.equal(got[1], l0)))
        outs = [torch.empty_like(xs[0]) for _ in range(3)]
        descs = []
        for j in range(POOL):
            d = N.FzXattnChain()
            d.x, d.res, d.packed, d.kv_packed, d.bias_out, d.y, d.y_ln = (xs[j].data_ptr(), rs[j].data_ptr(), packed.data_ptr(), kvp.data_ptr(),
                                                                            bo.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr())
            d.ln_gamma, d.ln_beta, d.ln_eps = g2.data_ptr(), b2.data_ptr(), 1e-5
            if front:
                d.front, d.y1, d.ln1_eps = 1, outs[2].data_ptr(), 1e-5
            d.rows, d.rows_per_frame, d.frames_per_batch, d.channels, d.heads, d.lk, d.scale = rows, 4096, clip, c, heads, lk, scale
            descs.append(d)
        L = N.lib()
        fn = L.fz_xattn_chain
        # the separate launches, straight through ctypes on preallocated buffers as well (the Python wrappers cost 20-30 us of host time per
        # call: three or four of them per round would make this side of the A/B host-bound)
        gd = N.FzGemmDesc()
        gd.rows, gd.in_features, gd.out_features, gd.ldx, gd.ldw, gd.ldy, gd.ldres, gd.batch = rows, c, c, c, c, c, c, 1
        cd = N.FzAttnCrossDesc()
        cd.n_frames, cd.frame0, cd.clip_len, cd.heads, cd.head_dim, cd.lq, cd.lk, cd.scale, cd.mode = frames, 0, clip, heads, 40, 4096, lk, scale, 0
        cd.q_frame_stride, cd.q_row_stride, cd.o_frame_stride, cd.o_row_stride = 4096 * c, c, 4096 * c, c
        cd.k_batch_stride, cd.k_row_stride, cd.vt_batch_stride, cd.vt_chan_stride = kk.stride(0), kk.stride(1), vt.stride(0), vt.stride(1)
        tmp = [torch.empty_like(xs[0]) for _ in range(6)]
        P = lambda t: t.data_ptr()

        def direct_launches(i, front=front):
            x, r = xs[i % POOL], rs[i % POOL]
            if front:
                L.fz_gemm_lnout(C.byref(gd), P(x), P(wo1), P(bo1), P(r), None, P(tmp[0]), P(g1), P(b1), 1e-5, P(tmp[1]), c, None, stream)
                x, r = tmp[1], tmp[0]
            L.fz_gemm(C.byref(gd), P(x), P(wq), None, None, None, P(tmp[2]), None, stream)
            L.fz_attn_cross(C.byref(cd), P(tmp[2]), P(kk), P(vt), P(tmp[3]), None, None, None, None, stream)
            L.fz_gemm_lnout(C.byref(gd), P(tmp[3]), P(wo), P(bo), P(r), None, P(tmp[4]), P(g2), P(b2), 1e-5, P(tmp[5]), c, None, stream)

        direct_launches(0)
        torch.cuda.synchronize()
        same = same and bool(torch.equal(tmp[4], y0)) and (l0 is None or bool(torch.equal(tmp[5], l0)))
        fns = {"launches": direct_launches, "one": lambda i: fn(C.byref(descs[i % POOL]), stream)}
        for name, lib in trial.items():
            fns[name] = (lambda f: (lambda i: f(C.byref(descs[i % POOL]), stream)))(lib.fz_xattn_chain)
        r = timeit(fns)
        t2, t1 = r["launches"], r["one"]
        print(f"{frames:2d} frames ({rows:6d} rows, {rows // 128:4d} workgroups) {'front + ' if front else '        '}attn2: "
              f"{4 if front else 3} launches {t2[0]:7.1f} us (min {t2[1]:7.1f})   one launch {t1[0]:7.1f} us (min {t1[1]:7.1f})   x{t2[0] / t1[0]:.2f}"
              f"   bit-identical: {same}", flush=True)
        if "timing" in trial:
            buf = (C.c_longlong * 16)()
            trial["timing"].fz_xattn_chain_timing.argtypes = [C.c_void_p]
            fns["timing"](0)
            torch.cuda.synchronize()
            trial["timing"].fz_xattn_chain_timing(buf)
            v = list(buf)
            for role, o in (("QA", 0), ("O ", 8)):
                print(f"      cycles {role} wave: DMA wait {v[o]} | barriers {v[o + 1]} | DMA issue {v[o + 2]} | prologue + sub-steps {v[o + 3]} | whole kernel {v[o + 4]}"
                      + (f" | q projections {v[o + 5]} | heads {v[o + 6]}" if o == 0 else ""))
        if trial:
            print("      trial builds (median us): " + "  ".join(f"{k} {v[0]:.1f}" for k, v in r.items() if k not in ("launches", "one")), flush=True)
