# Cycle counters of conv_halo_kernel (trial builds of scripts/conv_halo_variants.sh).  usage: [FZ_CH_LIB=build_tmp/libfz_ch_<name>.so] ch_timing.py
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from fatezero_amd import kernels as K, _native as N
lib = os.environ.get("FZ_CH_LIB", "build_tmp/libfz_ch_timing.so")
N.use_test_backend(os.path.abspath(lib)); N._is_test_backend = False
L = N.lib(); dev = "cuda"
print(f"== {lib}")
for (n, hw, cin, cout) in [(8, 64, 320, 320), (16, 64, 320, 320), (16, 32, 640, 640)]:
    x = torch.randn(n, hw * hw, cin, device=dev).half()
    wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3) * 0.02).half().to(dev)); b = torch.zeros(cout, device=dev).half()
    L.fz_conv_halo_timing2.argtypes = [C.c_void_p, C.c_int]
    for _ in range(3):
        L.fz_conv_halo_timing2(None, 1)
        y = K.conv3x3(x, wt, b, hw=(hw, hw), tile_cfg=154299)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 12)(); L.fz_conv_halo_timing.argtypes = [C.c_void_p]; L.fz_conv_halo_timing(buf)
    v = list(buf); steps = cin // 64 * 18
    b2 = (C.c_longlong * 12)(); L.fz_conv_halo_timing2(b2, 0); w = list(b2)
    print(f"{n} f x {hw}^2 x {cin} -> {cout}: {steps} steps")
    print(f"   consumer 0     : barriers {v[1]} | entry .. loop end {v[3]}  ({v[3] / steps:.0f} per step, MFMA 640)")
    print(f"   weight loader  : issue {v[6]} | issue + wait {v[4]} | barriers {v[5]} | total {v[7]}")
    print(f"   pixel loader   : chunk-end waits {v[8]} | barriers {v[9]} | total {v[11]}")
    wall = [w[0], w[2], w[4], w[6]]; clk = [w[1], w[3], w[5], w[7]]
    dw = [(wall[i + 1] - wall[i]) / 100.0 for i in range(3)]; dc = [clk[i + 1] - clk[i] for i in range(3)]
    mhz = (clk[3] - clk[0]) / max(1e-9, (wall[3] - wall[0]) / 100.0)
    print(f"   workgroup 0 (wall us / clock64): prologue {dw[0]:.2f} / {dc[0]} | loop {dw[1]:.2f} / {dc[1]} ({dc[1] / steps:.0f} per step) | epilogue {dw[2]:.2f} / {dc[2]}"
          f" | clock64 runs at {mhz:.0f} MHz | all workgroups: first entry .. last exit {(w[9] - w[8]) / 100.0:.2f} us, wg 0 entered {(w[0] - w[8]) / 100.0:.2f} us after the first")
    b3 = (C.c_longlong * 16)(); L.fz_conv_halo_timing3.argtypes = [C.c_void_p]; L.fz_conv_halo_timing3(b3); t = [x / 100.0 for x in b3]
    print(f"   wall us since entry: pixel loader set up {t[0]:.2f}, issued {t[1]:.2f}, landed {t[2]:.2f} | weight loader set up {t[4]:.2f}, issued {t[5]:.2f}, landed {t[6]:.2f}"
          f" | epilogue: loop end {(w[4] - w[0]) / 100.0:.2f}, sync {t[8]:.2f}, staged {t[9]:.2f}, sync {t[10]:.2f}, stored {t[11]:.2f}")
