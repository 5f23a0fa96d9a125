#!/usr/bin/env python3
"""Same-box A/B of the split-K launches with / without the K-slice -> XCD mapping (csrc/igemm.hip: flat grid): run once per setting
(the switch is read once per process: FZ_IGEMM_NO_XCD_KS=1 = the old 2-D grid) and compare the printed times.  The library's own
(tile, split-K) choice on the convolutions / projections / temporal convolutions of the 32^2, 16^2 and 8^2 levels at 8 and 16 frames."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd import kernels as K


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3  # us


def main():
    dev = "cuda"
    res = {}
    for (n, hw, cin, cout) in [(8, 32, 640, 640), (8, 32, 1280, 640), (8, 32, 1920, 640), (8, 16, 1280, 1280), (8, 16, 2560, 1280), (8, 16, 1920, 1280),
                               (8, 8, 1280, 1280), (8, 8, 2560, 1280), (16, 32, 640, 640), (16, 32, 1920, 640), (16, 16, 1280, 1280),
                               (16, 16, 2560, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280)]:
        x = torch.randn(n, hw * hw, cin).half().to(dev)
        wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev))
        b = torch.zeros(cout).half().to(dev)
        res[f"conv n{n} {hw}^2 {cin}->{cout}"] = timeit(lambda: K.conv3x3(x, wt, b, hw=(hw, hw)))
    for (rows, k, o) in [(8192, 2560, 640), (16384, 2560, 640), (2048, 5120, 1280), (4096, 5120, 1280), (512, 5120, 1280), (1024, 5120, 1280),
                         (2048, 1280, 1280), (4096, 1280, 1280), (512, 1280, 1280), (1024, 1280, 1280), (2048, 1280, 3840), (4096, 1280, 3840)]:
        x = torch.randn(rows, k).half().to(dev)
        w = (torch.randn(o, k) * 0.02).half().to(dev)
        r = torch.randn(rows, o).half().to(dev)
        res[f"gemm+res {rows}x{k}->{o}"] = timeit(lambda: K.gemm(x, w, None, res=r))
    for (n, tok, cin, cout) in [(8, 256, 1280, 160), (8, 256, 160, 1280), (8, 64, 1280, 160), (8, 64, 160, 1280), (8, 1024, 640, 160), (8, 1024, 160, 640),
                                (16, 256, 1280, 160), (16, 256, 160, 1280), (16, 1024, 640, 160)]:
        x = torch.randn(n, tok, cin).half().to(dev)
        w = (torch.randn(cout, 3, cin) * 0.02).half().to(dev)
        res[f"tconv n{n} tok{tok} {cin}->{cout}"] = timeit(lambda: K.temporal_conv3(x, w, clip_len=8))
    print(json.dumps({"xcd_ks": os.environ.get("FZ_IGEMM_NO_XCD_KS") is None, "us": {k: round(v, 2) for k, v in res.items()}}))



if __name__ == "__main__":
    main()
