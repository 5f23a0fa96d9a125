"""Interleaved A/B of igemm tile 252222 (320 x 128 as two K groups of 2 x 2 waves of 5 x 2 MFMA tiles, csrc/igemm.hip IgCfg::KG) against
254122 (the same output tile as 8 waves of 5 x 1) and the library's own choice, on the 3x3 convolutions and projections of the bench job that
run on the 320 x 128 tile.  Every variant goes straight through ctypes on preallocated buffers, a BATCH of launches between two HIP events,
round-robin; operands cycle through a pool (weights + activations from HBM / Infinity Cache as in the job).  TFLOP/s, median of the rounds."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K
from fatezero_amd import _native as N
import os
if os.environ.get("FZ_TRIAL_LIB"):  # an alternative build of the kernel library (same ABI)
    N.use_test_backend(os.path.abspath(os.environ["FZ_TRIAL_LIB"]))
    N._is_test_backend = False
    print("library:", os.environ["FZ_TRIAL_LIB"])

dev = "cuda"
torch.manual_seed(0)
POOL, BATCH, ROUNDS = 6, 8, 10
L = N.lib()
stream = K._stream(torch.zeros(1, device=dev))
ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
P = lambda t: None if t is None else t.data_ptr()
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 254122, 252222]


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


print("conv3x3: frames hw cin cout | " + " ".join(f"{c:>9d}" for c in cfgs) + "   (TFLOP/s; us)")
for (n, hw, cin, cout) in [(8, 64, 320, 320), (16, 64, 320, 320), (8, 64, 640, 320), (16, 64, 640, 320), (8, 64, 960, 320), (8, 32, 640, 640), (16, 32, 640, 640), (8, 32, 1280, 640), (16, 32, 1280, 640),
                           (8, 32, 1920, 640), (16, 32, 320, 640), (8, 16, 1280, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 640, 1280)]:
    xs = [torch.randn(n, hw * hw, cin, device=dev).half() for _ in range(POOL)]
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    b = torch.zeros(cout, device=dev).half()
    y = torch.empty(n, hw * hw, cout, device=dev, dtype=torch.float16)
    flops = 2.0 * n * hw * hw * cout * cin * 9

    def mk(cfg):
        def f(i):
            rc = L.fz_conv3x3(P(xs[i % POOL]), P(wt), P(b), None, 0, None, P(y), n, hw, hw, cin, cout, 1, 0, 8, P(ws), ws.numel(), cfg, 0 if cfg == 0 else 1, stream)
            assert rc == 0, rc
        return f
    ok = [c for c in cfgs if L.fz_conv3x3(P(xs[0]), P(wt), P(b), None, 0, None, P(y), n, hw, hw, cin, cout, 1, 0, 8, P(ws), ws.numel(), c, 0 if c == 0 else 1, stream) == 0]
    r = timeit({c: mk(c) for c in ok})
    for c in cfgs:
        r.setdefault(c, float("nan"))
    print(f"{n:3d} {hw:3d} {cin:5d} {cout:5d} | " + " ".join(f"{flops / r[c] / 1e6:9.0f}" for c in cfgs) + "   | " + " ".join(f"{r[c]:7.1f}" for c in cfgs), flush=True)
