"""Same-process A/B of fz_gemm between TWO builds of the kernel library loaded side by side (ctypes, RTLD_LOCAL): the projection shapes of the bench job
with a bias, interleaved batches between HIP events; us per launch, median.   usage: ab_two_libs_gemm.py <libA.so> <libB.so>"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from fatezero_amd import _native as N

dev = "cuda"
torch.manual_seed(0)
libs = {os.path.basename(p): N._open(os.path.abspath(p)) for p in sys.argv[1:3]}
stream = K._stream(torch.zeros(1, device=dev))
ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
POOL, BATCH, ROUNDS = 4, 8, 12
P = lambda t: None if t is None else t.data_ptr()


def timeit(fns):
    ev = {k: [] for k in fns}
    for i in range(ROUNDS + 2):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for r in range(BATCH):
                f(i * BATCH + r)
            e.record()
            if i >= 2:
                ev[k].append((s, e))
    torch.cuda.synchronize()
    return {k: sorted(s.elapsed_time(e) * 1e3 / BATCH for s, e in v)[len(v) // 2] for k, v in ev.items()}


print("fz_gemm: rows K O geglu res | " + " ".join(f"{k:>26s}" for k in libs) + "   (us per launch)")
for (rows, k, o, geglu, res) in [(16384, 640, 640, 0, 1), (8192, 640, 640, 0, 1), (4096, 1280, 1280, 0, 1), (2048, 1280, 1280, 0, 1), (1024, 1280, 1280, 0, 1), (512, 1280, 1280, 0, 1),
                                  (16384, 640, 5120, 1, 0), (8192, 640, 5120, 1, 0), (4096, 1280, 10240, 1, 0), (2048, 1280, 10240, 1, 0), (16384, 2560, 640, 0, 1), (4096, 5120, 1280, 0, 1),
                                  (65536, 320, 320, 0, 1), (32768, 320, 320, 0, 1), (16384, 640, 640, 0, 0), (4096, 1280, 1280, 0, 0)]:
    xs = [torch.randn(rows, k, device=dev).half() for _ in range(POOL)]
    w = (torch.randn(o, k, device=dev) * 0.03).half()
    b = (torch.randn(o, device=dev) * 0.1).half()
    oo = o // 2 if geglu else o
    r = torch.randn(rows, oo, device=dev).half() if res else None
    y = torch.empty(rows, oo, device=dev, dtype=torch.float16)
    d = N.FzGemmDesc()
    d.rows, d.in_features, d.out_features, d.ldx, d.ldw, d.ldy, d.ldres, d.batch = rows, k, o, k, k, oo, oo, 1
    d.epilogue = 1 if geglu else 0
    d.workspace_floats = 0 if geglu else ws.numel()

    def mk(L):
        def f(i):
            rc = L.fz_gemm(C.byref(d), P(xs[i % POOL]), P(w), P(b), P(r), None, P(y), None if geglu else P(ws), stream)
            assert rc == 0, rc
        return f
    t = timeit({name: mk(L) for name, L in libs.items()})
    print(f"{rows:6d} {k:5d} {o:6d} {geglu} {res} | " + " ".join(f"{t[name]:26.1f}" for name in libs), flush=True)
