# One launch per (shape, path) for a rocprofv3 --pmc pass: which path of the 3x3 convolution moves how many bytes (scripts/trials/pmc_conv_shapes.sh).
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fatezero_amd import kernels as K, _native as N
L = N.lib(); dev = "cuda"
stream = K._stream(torch.zeros(1, device=dev)); ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
P = lambda t: None if t is None else t.data_ptr()
for (n, hw, cin, cout, variants) in [(8, 64, 320, 320, [(254122, 1), (154299, 1)]), (16, 32, 640, 640, [(254222, 1), (154299, 1)]), (16, 16, 1280, 1280, [(0, 0), (154299, 2)]),
                                     (8, 32, 640, 640, [(254122, 2), (154299, 2)]), (8, 16, 1280, 1280, [(254122, 4), (154299, 4)])]:
    x = torch.randn(n, hw * hw, cin, device=dev).half(); wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
    b = torch.zeros(cout, device=dev).half(); y = torch.empty(n, hw * hw, cout, device=dev, dtype=torch.float16)
    for cfg, sk in variants:
        for _ in range(3):
            rc = L.fz_conv3x3(P(x), P(wt), P(b), None, 0, None, P(y), n, hw, hw, cin, cout, 1, 0, 8, P(ws), ws.numel(), cfg, sk, stream)
            assert rc == 0, (n, hw, cin, cout, cfg, sk, rc)
        torch.cuda.synchronize()
