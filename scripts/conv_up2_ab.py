"""fz_conv3x3_up2 (nearest-2x + 3x3 as four 2x2 convolutions on summed weights, csrc/conv_halo.hip) against fz_conv3x3(upsample=1) on the UNet's
upsampler shapes: straight ctypes launches, a batch between two HIP events, round-robin; us per launch, median."""
import sys
import torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K
from fatezero_amd import _native as N

dev = "cuda"
torch.manual_seed(0)
POOL, BATCH, ROUNDS = 4, 8, 10
L = N.lib()
stream = K._stream(torch.zeros(1, device=dev))
ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
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


print("nearest-2x + conv3x3: frames hw cin cout | nine taps   four 2x2   (us per launch)")
for (n, hw, cin, cout) in [(8, 8, 1280, 1280), (16, 8, 1280, 1280), (8, 16, 1280, 1280), (16, 16, 1280, 1280), (8, 32, 640, 640), (16, 32, 640, 640), (24, 32, 640, 640), (32, 16, 1280, 1280)]:
    xs = [torch.randn(n, hw * hw, cin, device=dev).half() for _ in range(POOL)]
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    wup = K.pack_conv3x3_up2_weight(wt)
    b = torch.zeros(cout, device=dev).half()
    y = torch.empty(n, 4 * hw * hw, cout, device=dev, dtype=torch.float16)

    def nine(i):
        assert L.fz_conv3x3(P(xs[i % POOL]), P(wt), P(b), None, 0, None, P(y), n, hw, hw, cin, cout, 1, 1, 8, P(ws), ws.numel(), 0, 0, stream) == 0

    def four(i):
        assert L.fz_conv3x3_up2(P(xs[i % POOL]), P(wup), P(b), P(y), n, hw, hw, cin, cout, stream) == 0
    r = timeit({"nine": nine, "four": four})
    print(f"{n:3d} {hw:3d} {cin:5d} {cout:5d} | {r['nine']:9.1f} {r['four']:9.1f}", flush=True)
