import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K, _native as N
N.use_test_backend(os.path.abspath("build_tmp/libfz_ch_timing.so")); N._is_test_backend = False
L = N.lib(); dev = "cuda"
for (n, hw, cin, cout) in [(8, 64, 320, 320), (16, 64, 320, 320), (16, 32, 640, 640)]:
    x = torch.randn(n, hw * hw, cin, device=dev).half()
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev)); b = torch.zeros(cout, device=dev).half()
    for _ in range(3):
        y = K.conv3x3(x, wt, b, hw=(hw, hw), tile_cfg=154299)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 12)(); L.fz_conv_halo_timing.argtypes = [C.c_void_p]; L.fz_conv_halo_timing(buf)
    v = list(buf); steps = cin // 64 * 18
    print(f"{n} f x {hw}^2 x {cin} -> {cout}: {steps} steps")
    print(f"   consumer 0     : barriers {v[1]} | loop total {v[3]}  ({v[3] / steps:.0f} per step, MFMA 640)")
    print(f"   weight loader  : issue {v[6]} | issue + wait {v[4]} | barriers {v[5]} | total {v[7]}")
    print(f"   pixel loader   : chunk-end waits {v[8]} | barriers {v[9]} | total {v[11]}")
