"""Interleaved A/B of the halo convolution under split-K (csrc/conv_halo.hip, tile id 154299 with split_k = 1 / 2 / 3 / 4: fp32 slabs + the split-K tail
kernel) against the library's own choice, on the 3x3 convolutions of the bench job whose 160 x 256 halo tiles do not fill the chip (8 frames x 32^2,
8 / 16 frames x 16^2).  Straight ctypes launches on preallocated buffers, a batch of launches between two HIP events, round-robin; us per launch,
median of the rounds."""
import sys
import torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K
from fatezero_amd import _native as N

dev = "cuda"
torch.manual_seed(0)
POOL, BATCH, ROUNDS = 6, 8, 10
L = N.lib()
stream = K._stream(torch.zeros(1, device=dev))
ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
P = lambda t: None if t is None else t.data_ptr()
variants = [("library", 0, 0), ("halo", 154299, 1), ("halo/2", 154299, 2), ("halo/3", 154299, 3), ("halo/4", 154299, 4), ("halo/5", 154299, 5), ("halo/6", 154299, 6), ("halo/8", 154299, 8), ("halo/10", 154299, 10)]
if "--narrow" in sys.argv:   # the 64-channel form of the kernel (tile id 154264) beside the library
    sys.argv.remove("--narrow")
    variants = [("library", 0, 0)] + [(f"n64/{k}", 154264, k) for k in (1, 2, 3, 4, 5, 6, 8)]


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


shapes = [(8, 32, 320, 640), (8, 32, 640, 640), (8, 32, 960, 640), (8, 32, 1280, 640), (8, 32, 1920, 640), (16, 32, 1920, 640),
          (8, 16, 640, 1280), (8, 16, 1280, 1280), (8, 16, 1920, 1280), (8, 16, 2560, 1280),
          (16, 16, 640, 1280), (16, 16, 1280, 1280), (16, 16, 1920, 1280), (16, 16, 2560, 1280), (24, 16, 1280, 1280), (32, 16, 1280, 1280)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
print("conv3x3: frames hw cin cout tiles | " + " ".join(f"{v[0]:>8s}" for v in variants) + "   (us per launch; tiles = 160 x 256 halo tiles)")
for (n, hw, cin, cout) in shapes:
    xs = [torch.randn(n, hw * hw, cin, device=dev).half() for _ in range(POOL)]
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    b = torch.zeros(cout, device=dev).half()
    res = torch.randn(n, hw * hw, cout, device=dev).half()
    y = torch.empty(n, hw * hw, cout, device=dev, dtype=torch.float16)

    def call(cfg, sk, i):
        return L.fz_conv3x3(P(xs[i % POOL]), P(wt), P(b), None, 0, P(res), P(y), n, hw, hw, cin, cout, 1, 0, 8, P(ws), ws.numel(), cfg, sk, stream)

    def mk(cfg, sk):
        def f(i):
            rc = call(cfg, sk, i)
            assert rc == 0, rc
        return f
    ok = [v for v in variants if call(v[1], v[2], 0) == 0]
    r = timeit({v[0]: mk(v[1], v[2]) for v in ok})
    print(f"{n:3d} {hw:3d} {cin:5d} {cout:5d} {(cout // 160) * (n * hw * hw // 256):5d} | " + " ".join(f"{r.get(v[0], float('nan')):8.1f}" for v in variants), flush=True)
