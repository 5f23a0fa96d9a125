"""Same-process A/B of fz_conv3x3 / fz_conv3x3_up2 between builds of the kernel library loaded side by side (ctypes): the halo-path shapes of the bench job,
interleaved batches between HIP events; us per launch, median.   usage: ab_two_libs_conv.py <libA.so> <libB.so> [...]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K
from fatezero_amd import _native as N

dev = "cuda"
torch.manual_seed(0)
libs = {os.path.basename(p): N._open(os.path.abspath(p)) for p in sys.argv[1:]}
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


print("fz_conv3x3 (library's own path): frames hw cin cout up | " + " ".join(f"{k:>24s}" for k in libs) + "   (us per launch)")
for (n, hw, cin, cout, up) in [(8, 64, 320, 320, 0), (16, 64, 320, 320, 0), (8, 64, 640, 320, 0), (16, 64, 640, 320, 0), (16, 32, 640, 640, 0), (16, 32, 1280, 640, 0), (8, 32, 640, 640, 0),
                               (16, 16, 1280, 1280, 0), (8, 16, 1280, 1280, 0), (16, 8, 1280, 1280, 0), (16, 32, 640, 640, 1), (8, 16, 1280, 1280, 1)]:
    xs = [torch.randn(n, hw * hw, cin, device=dev).half() for _ in range(POOL)]
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    wup = K.pack_conv3x3_up2_weight(wt) if up else None
    b = torch.zeros(cout, device=dev).half()
    y = torch.empty(n, (4 if up else 1) * hw * hw, cout, device=dev, dtype=torch.float16)

    def mk(L):
        def f(i):
            if up:
                rc = L.fz_conv3x3_up2(P(xs[i % POOL]), P(wup), P(b), P(y), n, hw, hw, cin, cout, stream)
            else:
                rc = L.fz_conv3x3(P(xs[i % POOL]), P(wt), P(b), None, 0, None, P(y), n, hw, hw, cin, cout, 1, 0, 8, P(ws), ws.numel(), 0, 0, stream)
            assert rc == 0, rc
        return f
    t = timeit({name: mk(L) for name, L in libs.items()})
    print(f"{n:3d} {hw:3d} {cin:5d} {cout:5d} {up} | " + " ".join(f"{t[name]:24.1f}" for name in libs), flush=True)
