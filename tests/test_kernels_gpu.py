"""MI355X parity of every HIP kernel, called through the C ABI (libfatezero_hip.so), against fp32 torch references
(kernel_cases.py).  Shapes include the real SD-1.x levels at 512^2 (d = 40/80/160)."""
import pytest
import torch

from fatezero_amd import _native
from fatezero_amd import kernels as K

import kernel_cases as KC

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def hip_backend():
    _native.reset_backend()
    assert "hip" in K.version()
    assert _native.loaded_path().endswith("libfatezero_hip.so")
    yield


@pytest.mark.parametrize("d", [16, 32, 40, 64, 80, 128, 160])
def test_self_flash_all_head_dims(d):
    KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=d, lq=256, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH)


def test_self_flash_sd_level_64():
    # the 64x64 level of SD-1.x at 512^2: Lq 4096, Lk 8192, d 40 (reduced to 2 frames x 2 heads for the CPU reference)
    r = KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=40, lq=4096, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH)
    print("flash 64^2", r)


@pytest.mark.parametrize("lq,qk_scale,shape", [(64, 1.5, None), (600, 1.5, None), (1024, 6.0, None), (576, 3.0, "ramp"),
                                               (320, 3.0, "negative"), (4096, 1.5, None)])
def test_self_flash_log2_folded_q(lq, qk_scale, shape):
    # d=40 with q delivered in the log2 domain: the running max rides in contraction slot 40 of the QK^T MFMA
    r = KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=40, lq=lq, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH,
                          qk_scale=qk_scale, shape=shape, fold=True)
    print("flash folded", lq, qk_scale, shape, r)


@pytest.mark.parametrize("batch,clip,heads,lq,index_list", [(2, 3, 8, 64, [-1, "first"]), (1, 3, 16, 100, [-1, "first"]),
                                                             (2, 4, 8, 64, [-1, "first", "first"]), (1, 8, 8, 64, [-1, "first"])])
def test_self_flash_coinciding_kv_slots(batch, clip, heads, lq, index_list):
    # frames 0 and 1 of a clip see frame 0 in both slots: the kernel reads each distinct source once (same softmax), and -- when every
    # XCD owns whole heads (8 | heads) -- the launcher dispatches the full-length frames first; the oracle lists the keys twice
    KC.case_attn_self(DEV, batch=batch, clip=clip, heads=heads, d=40, lq=lq, index_list=index_list, mode=K.FZ_ATTN_FLASH, fold=True)
    KC.case_attn_self(DEV, batch=batch, clip=clip, heads=heads, d=80, lq=lq, index_list=index_list, mode=K.FZ_ATTN_FLASH)


@pytest.mark.parametrize("lq,index_list,shape", [(576, ["mid"], "ramp"), (600, ["mid"], None), (640, [-1, "mid", 1], "ramp")])
def test_self_flash_pair_ring_odd_tiles(lq, index_list, shape):
    # the d=40 log2-domain variant meets at a barrier every SECOND 64-key tile (4-stage K/V ring): odd tile counts (9, 10
    # with a ragged last tile whose padded keys are neutralised in the stash, 30 over three kv slots)
    KC.case_attn_self(DEV, batch=1, clip=3, heads=1, d=40, lq=lq, index_list=index_list, mode=K.FZ_ATTN_FLASH,
                      qk_scale=3.0, shape=shape, fold=True)


def test_self_capture_and_inject_log2_folded_q():
    KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=40, lq=324, index_list=[-1, "first"], mode=K.FZ_ATTN_CAPTURE,
                      fold=True)
    KC.case_attn_self(DEV, batch=2, clip=2, heads=2, d=40, lq=256, index_list=["mid"], mode=K.FZ_ATTN_INJECT,
                      mask_kind="random", fold=True)


def test_self_flash_masking_and_slots():
    KC.case_attn_self(DEV, batch=1, clip=3, heads=1, d=32, lq=200, index_list=[-1, "mid", 1], mode=K.FZ_ATTN_FLASH)
    KC.case_attn_self(DEV, batch=2, clip=2, heads=8, d=64, lq=64, index_list=[], mode=K.FZ_ATTN_FLASH)
    KC.case_attn_self(DEV, batch=1, clip=4, heads=8, d=40, lq=1296, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH)


@pytest.mark.parametrize("d,lq", [(80, 1024), (160, 256), (160, 64), (40, 324), (160, 81)])
def test_self_capture(d, lq):
    r = KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=d, lq=lq, index_list=[-1, "first"], mode=K.FZ_ATTN_CAPTURE)
    print("capture", d, lq, r)


@pytest.mark.parametrize("index_list", [[-1, "first"], ["mid"], [-1, "mid", 1]])
@pytest.mark.parametrize("d,lq", [(80, 1024), (160, 256), (160, 64)])
def test_self_capture_three_frames(d, lq, index_list):
    # clip = 3: the K/V slots of a frame are DISTINCT frames (with clip = 2 and [-1, 'first'] both resolve to frame 0)
    r = KC.case_attn_self(DEV, batch=1, clip=3, heads=2, d=d, lq=lq, index_list=index_list, mode=K.FZ_ATTN_CAPTURE, seed=3)
    print("capture clip3", d, lq, index_list, r)


@pytest.mark.parametrize("mask_kind", [None, "random"])
@pytest.mark.parametrize("d,lq,index_list", [(80, 1024, [-1, "first"]), (160, 256, [-1, "first"]), (80, 1024, ["mid"]),
                                             (160, 64, [-1, "mid", 1])])
def test_self_inject_three_frames(mask_kind, d, lq, index_list):
    KC.case_attn_self(DEV, batch=2, clip=3, heads=2, d=d, lq=lq, index_list=index_list, mode=K.FZ_ATTN_INJECT,
                      mask_kind=mask_kind, seed=4)


@pytest.mark.parametrize("mask_kind", [None, "random", "rows"])
@pytest.mark.parametrize("d,lq", [(80, 1024), (160, 256), (32, 81)])
def test_self_inject(mask_kind, d, lq):
    KC.case_attn_self(DEV, batch=2, clip=2, heads=2, d=d, lq=lq, index_list=[-1, "first"], mode=K.FZ_ATTN_INJECT,
                      mask_kind=mask_kind)


@pytest.mark.parametrize("mode", [K.FZ_ATTN_FLASH, K.FZ_ATTN_CAPTURE, K.FZ_ATTN_INJECT])
@pytest.mark.parametrize("d,lq", [(40, 4096), (80, 1024), (160, 256), (160, 64), (16, 100)])
def test_cross(mode, d, lq):
    KC.case_attn_cross(DEV, batch=2, clip=2, heads=2, d=d, lq=lq, mode=mode)


def test_temporal():
    KC.case_attn_temporal(DEV, batch=2, clip=8, heads=8, d=40, tokens=300)
    KC.case_attn_temporal(DEV, batch=1, clip=3, heads=2, d=160, tokens=17)
    KC.case_attn_temporal(DEV, batch=1, clip=16, heads=8, d=80, tokens=1024)
    KC.case_attn_temporal(DEV, batch=2, clip=8, heads=8, d=160, tokens=256)
    KC.case_attn_temporal(DEV, batch=1, clip=8, heads=8, d=40, tokens=4096)
    # BASELINE cfg4 / cfg5 clip lengths: register-resident forms for 24 and 32 frames (one token of a 32-frame clip per workgroup at
    # 320 channels = 40 KB of K | V; at 1280 channels 160 KB: the whole LDS) and a clip length without one (generic paths)
    KC.case_attn_temporal(DEV, batch=2, clip=32, heads=8, d=40, tokens=5184)
    KC.case_attn_temporal(DEV, batch=1, clip=24, heads=8, d=40, tokens=4096)
    KC.case_attn_temporal(DEV, batch=1, clip=32, heads=8, d=80, tokens=1296)
    KC.case_attn_temporal(DEV, batch=1, clip=32, heads=8, d=160, tokens=324)
    KC.case_attn_temporal(DEV, batch=1, clip=24, heads=8, d=160, tokens=81)
    KC.case_attn_temporal(DEV, batch=1, clip=20, heads=8, d=40, tokens=300)


@pytest.mark.parametrize("span,c,groups,tokens", [(8, 320, 32, 4096), (1, 640, 32, 1024), (4, 2560, 32, 64),
                                                  (2, 1920, 32, 256), (2, 960, 32, 1000), (2, 80, 16, 100)])
def test_groupnorm(span, c, groups, tokens):
    KC.case_groupnorm(DEV, n=span * 2, span=span, tokens=tokens, c=c, groups=groups, silu=True)
    KC.case_groupnorm(DEV, n=span, span=span, tokens=tokens, c=c, groups=groups, silu=False, eps=1e-6)

@pytest.mark.parametrize("span,tokens,c,sets", [(1, 64, 1280, 8), (1, 300, 640, 4), (1, 256, 1280, 16), (2, 256, 960, 2), (8, 64, 1280, 2),
                                                (8, 100, 1280, 1), (8, 64, 2560, 2), (3, 37, 960, 1), (1, 1024, 640, 8)])
def test_groupnorm_one_launch_form(span, tokens, c, sets):
    # launches that take the one-launch form (csrc/norms.hip gn_fused_kernel: <= 20 channel pairs per thread, <= 2560 pairs-per-thread x
    # workgroups): every register bucket (2, 4, 5, 8, 10, 16, 20), the rule's boundary (512 workgroups x 5; 256 x 10), a ragged tail
    KC.case_groupnorm(DEV, n=span * sets, span=span, tokens=tokens, c=c, groups=32, silu=True)


def test_groupnorm_three_kernel_form_beyond_the_rule():
    KC.case_groupnorm(DEV, n=8, span=8, tokens=256, c=1280, groups=32, silu=True)    # 40 pairs per thread
    KC.case_groupnorm(DEV, n=16, span=1, tokens=1024, c=640, groups=32, silu=True)   # 10 pairs per thread x 512 workgroups
    KC.case_groupnorm_cat(DEV, n=4, span=4, tokens=1500, c1=320, c2=640, groups=32)  # 90 000 pairs per group, groups straddle the seam


def test_groupnorm_one_launch_form_of_a_lazy_concatenation():
    KC.case_groupnorm_cat(DEV, n=8, span=8, tokens=64, c1=1280, c2=1280, groups=32)   # seam on a group boundary
    KC.case_groupnorm_cat(DEV, n=16, span=8, tokens=64, c1=1280, c2=640, groups=32)   # 60-channel groups: group 21 straddles the seam
    KC.case_groupnorm_cat(DEV, n=4, span=1, tokens=256, c1=640, c2=320, groups=32)


@pytest.mark.parametrize("n,span,tokens,c1,c2", [(16, 8, 4096, 320, 320), (8, 8, 4096, 320, 640), (16, 8, 1024, 640, 1280),
                                                 (8, 8, 256, 1280, 1280), (16, 8, 64, 1280, 1280), (2, 1, 100, 24, 40)])
def test_groupnorm_of_a_lazy_concatenation(n, span, tokens, c1, c2):
    # the up blocks' torch.cat([x, skip]) -> GroupNorm without the concatenated copy; real SD-1.x up-path shapes
    print(KC.case_groupnorm_cat(DEV, n=n, span=span, tokens=tokens, c1=c1, c2=c2, groups=32 if c1 >= 320 else 8))


def test_layernorm_geglu_transpose_latent():
    KC.case_layernorm(DEV, rows=4099, c=320)
    KC.case_layernorm(DEV, rows=513, c=1280)
    KC.case_layernorm(DEV, rows=64, c=640)
    KC.case_geglu(DEV, rows=1000, inner=1280)
    KC.case_transpose_pad(DEV, n=2, l=77, c=320, lp=96)
    KC.case_transpose_pad(DEV, n=3, l=1000, c=640, lp=1024)
    KC.case_latent_update(DEV, frames=8, hw=4096, blend=False, cfg=False)
    KC.case_latent_update(DEV, frames=8, hw=4096, blend=True, cfg=True)


@pytest.mark.parametrize("res,out_hw,prompts,or_first", [(16, (32, 32), 1, False), (16, (16, 16), 1, False),
                                                         (16, (8, 8), 1, False), (16, (64, 64), 2, True),
                                                         (18, (36, 36), 1, False), (18, (9, 9), 1, False)])
def test_blend_mask_bit_exact(res, out_hw, prompts, or_first):
    for seed in range(4):
        KC.case_blend_mask(DEV, prompts=prompts, frames=8, heads=8, res=res, out_hw=out_hw, or_first=or_first, seed=seed)


@pytest.mark.parametrize("kw", [dict(n=8, h=64, w=64, cin=320, cout=320, with_temb=True, fpb=8),
                                dict(n=2, h=64, w=64, cin=960, cout=320, with_res=True),
                                dict(n=4, h=32, w=32, cin=640, cout=640, stride=2),
                                dict(n=4, h=16, w=16, cin=1280, cout=1280, upsample=True),
                                dict(n=2, h=8, w=8, cin=2560, cout=1280, with_temb=True, with_res=True, fpb=2),
                                dict(n=3, h=9, w=9, cin=96, cout=40)])
def test_conv3x3(kw):
    r = KC.case_conv3x3(DEV, **kw)
    print(kw, r)


@pytest.mark.parametrize("kw", [dict(n=8, h=64, w=64, cin=4, cout=320, with_temb=True, fpb=8),        # conv_in (direct)
                                dict(n=8, h=64, w=64, cin=320, cout=4),                                # conv_out
                                dict(n=8, h=16, w=16, cin=2560, cout=1280, with_temb=True, fpb=8),     # split-K territory
                                dict(n=8, h=8, w=8, cin=1280, cout=1280, with_res=True),
                                dict(n=16, h=8, w=8, cin=2560, cout=1280, with_temb=True, with_res=True, fpb=8)])
def test_conv3x3_small_levels_and_ends(kw):
    r = KC.case_conv3x3(DEV, **kw)
    print(kw, r)


@pytest.mark.parametrize("tile_cfg,split_k", [(254222, 1), (254122, 1), (244222, 1), (224223, 1), (222222, 1), (212222, 1), (222222, 4), (244222, 2), (254222, 4), (254122, 2), (158122, 1), (158122, 2),
                                              (254218, 1), (244218, 1), (254218, 4), (244218, 2), (252222, 1), (252222, 2), (252218, 1), (252218, 2)])
def test_conv3x3_every_tile_shape(tile_cfg, split_k):
    KC.case_conv3x3(DEV, n=4, h=16, w=16, cin=640, cout=320, with_temb=True, with_res=True, fpb=2, tile_cfg=tile_cfg,
                    split_k=split_k)


@pytest.mark.parametrize("rows,k,o,kw", [(32768, 320, 320, dict(n_res=1)), (32768, 320, 640, dict(bias=False)),
                                         (65536, 320, 960, dict(bias=False)), (8192, 640, 640, dict(n_res=2)),
                                         (8192, 2560, 640, dict(n_res=1)), (2048, 1280, 1280, dict()),
                                         (4096, 5120, 1280, dict(n_res=1)), (512, 1280, 3840, dict(bias=False)),
                                         (2, 1280, 14080, dict()), (154, 768, 1280, dict(bias=False)),
                                         (1000, 320, 328, dict(n_res=1, ldx_extra=320, ldy_extra=8))])
def test_gemm_sd_shapes(rows, k, o, kw):
    r = KC.case_gemm(DEV, rows=rows, k=k, o=o, **kw)
    print(rows, k, o, kw, r)


@pytest.mark.parametrize("rows,k,o", [(32768, 320, 2560), (65536, 320, 2560), (8192, 640, 5120), (4100, 640, 5120), (16384, 640, 5120),
                                      (2048, 1280, 10240), (300, 64, 256)])
def test_gemm_geglu(rows, k, o):
    # (the first four take the 128 x 256 two-workgroups-per-CU tile by the library's rule, csrc/igemm.hip ig_run)
    r = KC.case_gemm(DEV, rows=rows, k=k, o=o, geglu=True)
    print(rows, k, o, r)


def test_gemm_geglu_two_workgroups_per_cu_tile_forced():
    KC.case_gemm(DEV, rows=3000, k=640, o=1024, geglu=True, tile_cfg=224212)
    KC.case_gemm(DEV, rows=333, k=320, o=2560, geglu=True, bias=False, tile_cfg=224212)


def test_conv3x3_with_the_pixel_halo_resident_in_lds():
    """csrc/conv_halo.hip at the judged shapes: 8 / 16 frames x 64^2 x 320 -> 320 (+ time embedding + residual), 640 -> 320, 32^2 x 640 -> 640."""
    print(KC.case_conv3x3(DEV, n=8, h=64, w=64, cin=320, cout=320, with_temb=True, with_res=True, fpb=8, tile_cfg=154299))
    print(KC.case_conv3x3(DEV, n=16, h=64, w=64, cin=640, cout=320, with_res=True, fpb=8, tile_cfg=154299, seed=1))
    print(KC.case_conv3x3(DEV, n=16, h=32, w=32, cin=640, cout=640, with_temb=True, fpb=8, tile_cfg=154299, seed=2))
    print(KC.case_conv3x3(DEV, n=3, h=8, w=32, cin=64, cout=160, tile_cfg=154299, seed=3))
    # split-K (fp32 slabs + the tail kernel) where the halo tiles alone do not fill the chip: 8 frames x 32^2, the 16^2 level (W = 16: a tile is a frame)
    print(KC.case_conv3x3(DEV, n=8, h=32, w=32, cin=640, cout=640, with_temb=True, with_res=True, fpb=8, tile_cfg=154299, split_k=2, seed=4))
    print(KC.case_conv3x3(DEV, n=16, h=16, w=16, cin=1280, cout=1280, with_temb=True, fpb=8, tile_cfg=154299, split_k=2, seed=5))
    print(KC.case_conv3x3(DEV, n=8, h=16, w=16, cin=2560, cout=1280, with_res=True, fpb=8, tile_cfg=154299, split_k=4, seed=6))
    print(KC.case_conv3x3(DEV, n=8, h=16, w=16, cin=640, cout=1280, fpb=8, tile_cfg=154299, seed=7))
    # the 8 x 8 level: four whole frames per tile, each with its own halo
    print(KC.case_conv3x3(DEV, n=16, h=8, w=8, cin=1280, cout=1280, with_temb=True, with_res=True, fpb=8, tile_cfg=154299, split_k=8, seed=8))
    print(KC.case_conv3x3(DEV, n=8, h=8, w=8, cin=2560, cout=1280, with_temb=True, fpb=8, tile_cfg=154299, split_k=4, seed=9))


def test_conv3x3_up2_four_subpixel_convolutions():
    """fz_conv3x3_up2 at the UNet's upsamplers (16^2 x 1280 and 32^2 x 640, 8 / 16 frames) against torch fp32 and the nine-tap path."""
    print(KC.case_conv3x3_up2(DEV, n=8, h=16, w=16, cin=1280, cout=1280))
    print(KC.case_conv3x3_up2(DEV, n=16, h=32, w=32, cin=640, cout=640, seed=1))
    print(KC.case_conv3x3_up2(DEV, n=3, h=32, w=16, cin=128, cout=160, seed=2))
    print(KC.case_conv3x3_up2(DEV, n=8, h=8, w=8, cin=1280, cout=1280, seed=3))


@pytest.mark.parametrize("tile_cfg", [254222, 254122, 158122, 244222, 224223, 222222, 212222, 254218, 244218, 252222, 252218])
def test_gemm_every_tile_shape(tile_cfg):
    KC.case_gemm(DEV, rows=3000, k=640, o=960, n_res=1, tile_cfg=tile_cfg)
    KC.case_gemm(DEV, rows=520, k=1280, o=320, tile_cfg=tile_cfg, split_k=4)


@pytest.mark.parametrize("ring,pp", [(254222, 254218), (244222, 244218)])
def test_pingpong_loop_bit_equal_to_ring_loop(ring, pp):
    """The phase-interleaved K loop contracts in the same k order into the same accumulators as the 2-stage ring loop of the
    same tile: outputs must be BIT-equal (real SD shapes: tap-outer and chunk-outer 3x3 convs, stride 2, nearest-2x, ragged
    Cin, split-K, a long-K and a short-K GEMM)."""
    g = torch.Generator().manual_seed(7)
    # exact = the launch runs tap-outer (K order (tap, k) whatever the K step); the chunk-outer launches (igemm.hip fz_conv3x3:
    # per-XCD tap window above L2) walk (k chunk, tap) and so contract in a different order with K step 32 than with 64: those
    # agree to an fp16 rounding of the result
    for (n, hw, cin, cout, stride, up, sk, exact) in [(8, 64, 320, 320, 1, False, 1, True), (16, 64, 320, 320, 1, False, 1, False),
                                                      (4, 32, 1920, 640, 1, False, 1, False), (4, 32, 640, 640, 2, False, 1, False),
                                                      (4, 32, 320, 640, 2, False, 1, True),
                                                      (4, 16, 1280, 1280, 1, True, 1, True), (8, 16, 2560, 1280, 1, False, 4, False),
                                                      (2, 24, 352, 320, 1, False, 1, True)]:
        x = torch.randn(n, hw * hw, cin, generator=g).half().to(DEV)
        wt = K.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3, generator=g) * 0.02).half().to(DEV))
        b = torch.randn(cout, generator=g).half().to(DEV)
        ya, _ = K.conv3x3(x, wt, b, hw=(hw, hw), stride=stride, upsample=up, tile_cfg=ring, split_k=sk)
        yb, _ = K.conv3x3(x, wt, b, hw=(hw, hw), stride=stride, upsample=up, tile_cfg=pp, split_k=sk)
        assert torch.isfinite(ya.float()).all()
        if exact:
            assert torch.equal(ya, yb), (n, hw, cin, cout, stride, up, sk)
        else:
            assert float((ya.float() - yb.float()).abs().max()) <= 2.0 ** -9 * float(ya.float().abs().max()), (n, hw, cin, cout)
    for (rows, k, o) in [(32768, 320, 320), (8192, 2560, 640), (4096, 1280, 3840), (1000, 352, 648)]:
        x = torch.randn(rows, k, generator=g).half().to(DEV)
        w = (torch.randn(o, k, generator=g) * 0.03).half().to(DEV)
        b = torch.randn(o, generator=g).half().to(DEV)
        assert torch.equal(K.gemm(x, w, b, tile_cfg=ring), K.gemm(x, w, b, tile_cfg=pp)), (rows, k, o)


@pytest.mark.parametrize("kw", [dict(n=8, l=4096, k=320, c=320), dict(n=16, l=4096, k=320, c=320), dict(n=8, l=1024, k=640, c=640),
                                dict(n=16, l=256, k=1280, c=1280), dict(n=8, l=64, k=1280, c=1280), dict(n=2, l=5184, k=320, c=320),
                                dict(n=4, l=1024, k=640, c=640, tile_cfg=254222), dict(n=4, l=1024, k=640, c=640, tile_cfg=222222)])
def test_gemm_qkvt_one_launch(kw):
    """q | k | V^T of a self-attention in one launch at every SD-1.x level (8- and 16-frame launches, 576^2's 5184 tokens): vs fp32
    torch and bit for bit vs the two launches it replaces (same products, same K order, fp32 accumulation, one rounding)."""
    r = KC.case_gemm_qkvt(DEV, **kw)
    print("qkvt", kw, r)
    assert r["qk_bit_equal"] and r["vt_bit_equal"], r


@pytest.mark.parametrize("kw", [dict(n=8, clip=8, tokens=4096, cin=160, cout=320), dict(n=16, clip=8, tokens=4096, cin=160, cout=320),
                                dict(n=16, clip=8, tokens=1024, cin=160, cout=640), dict(n=8, clip=8, tokens=4096, cin=320, cout=320, producer="gemm"),
                                dict(n=16, clip=8, tokens=4096, cin=320, cout=320, producer="gemm")])
def test_groupnorm_statistics_from_the_producing_epilogue(kw):
    """The 64^2-level producers of the bench job (LoRA up convolution at 8 / 16 frames, proj_out) and the 640-wide 32^2 one at 16 frames:
    output bit-identical to the plain launch, GroupNorm from the epilogue's partials vs the three-kernel form and vs fp32 torch."""
    r = KC.case_gn_from_epilogue(DEV, **kw)
    print("gn from epilogue", kw, r)
    if kw["tokens"] == 4096:
        assert r is not None, "the 64^2 producers run on 320-wide tiles: the statistics must come from their epilogue"


def test_groupnorm_statistics_epilogue_ragged_last_tile():
    """128 * odd rows on the 320 x 256 tile: the last tile's second statistics pass is beyond the rows and writes nothing (advisor, round 4)."""
    r = KC.case_gn_epilogue_ragged_last_tile(DEV)
    assert r["records"] == 3 * 32
    KC.case_gn_epilogue_ragged_last_tile(DEV, frames=5, cin=320)


def test_gemm_qkvt_rejects_a_pinned_tile_across_the_kv_boundary():
    """fz_gemm_qkvt with an explicit tile_cfg whose width does not divide the k | v boundary: FZ_ERR_BAD_ARG, nothing launched."""
    x = torch.zeros(1, 64, 64, dtype=torch.float16, device=DEV)
    w = torch.zeros(3 * 64, 64, dtype=torch.float16, device=DEV)
    K.gemm_qkvt(x, w, 128, tile_cfg=212222)  # 64-wide tile: divides 128
    with pytest.raises(RuntimeError):
        K.gemm_qkvt(x, w, 128, tile_cfg=254222)  # 320 does not divide 128
    with pytest.raises(RuntimeError):
        K.gemm_qkvt(x, w, 128, tile_cfg=123456)  # not a qkvt tile at all


def test_layernorm_out_of_the_producing_gemm_epilogue():
    """fz_gemm_lnout at the 64x64-level shapes of the bench job (8 / 16 frames x 4096 tokens -> 320 channels, K = 320 and K = 1280: the 320 x 128
    ring tile, the 320 x 256 ring tile and the ping-pong tile), ragged rows: y bit-identical to fz_gemm, LN(y) vs fp32 torch and fz_layernorm."""
    for kw in (dict(rows=32768, k=320, n_res=1), dict(rows=65536, k=320, n_res=1, seed=1), dict(rows=32768, k=1280, n_res=1, seed=2),
               dict(rows=65536, k=1280, n_res=1, mean_shift=4.0, seed=3), dict(rows=32768, k=320, n_res=0, seed=4), dict(rows=32768 + 72, k=320, n_res=2, seed=5, tile_cfg=254122)):  # (ragged last tile: pinned)
        r = KC.case_gemm_lnout(DEV, **kw)
        print("gemm_lnout", kw, r)
        assert r is not None
    assert KC.case_gemm_lnout(DEV, rows=512, k=1280, n_res=1, expect=False) is None   # 8x8 level rows: small tiles / split-K: reports it


def test_feed_forward_chain_in_one_launch():
    """csrc/ff_chain.hip at the 64x64-level shapes (8 / 16 frames x 4096 tokens x 320, inner 1280): bit-identical to the two launches it
    replaces, within fp16 rounding of fp32 torch; a ragged row count; no bias / residual / LayerNorm."""
    print(KC.case_ff_chain(DEV, rows=32768))
    print(KC.case_ff_chain(DEV, rows=65536, seed=1))
    print(KC.case_ff_chain(DEV, rows=4096 * 3 + 200, seed=2))
    print(KC.case_ff_chain(DEV, rows=8192, bias=False, res=False, ln=False, seed=3))
    print(KC.case_ff_chain(DEV, rows=1000, inner=96, seed=4))


def test_cross_attention_chain_in_one_launch():
    """csrc/xattn_chain.hip at the 64x64-level shapes (8 / 16 frames x 4096 tokens x 320, 8 heads of 40, 77 text keys, two text contexts at
    16 frames): bit-identical to fz_gemm + fz_attn_cross + fz_gemm_lnout (and fz_gemm_lnout in front), within fp16 rounding of fp32 torch."""
    print(KC.case_xattn_chain(DEV, n=8, tokens=4096, clip=8))
    print(KC.case_xattn_chain(DEV, n=16, tokens=4096, clip=8, seed=1))
    print(KC.case_xattn_chain(DEV, n=8, tokens=4096, clip=8, front=True, seed=2))
    print(KC.case_xattn_chain(DEV, n=16, tokens=4096, clip=8, front=True, seed=3))
    print(KC.case_xattn_chain(DEV, n=3, tokens=1024, clip=2, front=True, bias=False, ln=False, seed=4, lk=60))
    print(KC.case_gemm_vt(DEV, n=2, l=50, k=64, c=80, lp=96))


def test_gemm_transposed_output():
    KC.case_gemm_vt(DEV, n=8, l=4096, k=320, c=320, lp=4096)
    KC.case_gemm_vt(DEV, n=4, l=1024, k=640, c=640, lp=1024)
    KC.case_gemm_vt(DEV, n=2, l=77, k=768, c=1280, lp=96)
    KC.case_gemm_vt(DEV, n=3, l=1296, k=320, c=320, lp=1344)


@pytest.mark.parametrize("kw", [dict(batch=2, clip=8, tokens=4096, c=320), dict(batch=1, clip=16, tokens=1024, c=640),
                                dict(batch=2, clip=8, tokens=256, c=1280), dict(batch=1, clip=8, tokens=64, c=1280, with_temb=False),
                                dict(batch=1, clip=4, tokens=1024, c=320, with_res2=False), dict(batch=3, clip=1, tokens=128, c=640),
                                dict(batch=1, clip=8, tokens=4096, c=320, gn_groups=32), dict(batch=2, clip=16, tokens=4096, c=320, gn_groups=32),
                                dict(batch=1, clip=8, tokens=4096, c=640, gn_groups=32), dict(batch=2, clip=2, tokens=256, c=640, gn_groups=32)],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_lora_pair_one_launch(kw):
    """fz_lora_pair: bit-identical to fz_temporal_conv3 twice, and fp32 torch within fp16 rounding; gn_groups: fz_lora_pair_gn's partials
    normalise like the three-kernel GroupNorm (thousands of records per statistics set: the four-wave finalize)."""
    r = KC.case_lora_pair(DEV, **kw)
    assert r["bit_identical_to_two_launches"]
    assert not kw.get("gn_groups") or r["partial"] is not None


def test_temporal_conv3():
    KC.case_temporal_conv3(DEV, batch=2, clip=8, tokens=4096, cin=320, cout=160, with_res=False)
    KC.case_temporal_conv3(DEV, batch=2, clip=8, tokens=4096, cin=160, cout=320, with_res=True)
    KC.case_temporal_conv3(DEV, batch=1, clip=8, tokens=64, cin=160, cout=1280, with_res=True)
    KC.case_temporal_conv3(DEV, batch=1, clip=3, tokens=100, cin=1280, cout=160, with_res=False)
    # conv_out's channel counts (rank-2 LoRA pair 4 -> 2 -> 4, plain Conv1d 4 -> 4): the direct kernel
    KC.case_temporal_conv3(DEV, batch=2, clip=8, tokens=4096, cin=4, cout=2, with_res=False)
    KC.case_temporal_conv3(DEV, batch=2, clip=8, tokens=4096, cin=2, cout=4, with_res=True)
    KC.case_temporal_conv3(DEV, batch=1, clip=3, tokens=77, cin=4, cout=4, with_res=True, with_rows=True)
    KC.case_temporal_conv3(DEV, batch=2, clip=4, tokens=256, cin=320, cout=320, with_res=True, with_rows=True)


@pytest.mark.parametrize("lo,hi", [(0, 2), (2, 4), (3, 5)])
def test_frame_shard_kernel_forms(lo, hi):
    # what a rank owning frames [lo, hi) of a 5-frame clip launches, against the single-GPU kernels on the whole clip
    KC.case_sharded_pieces(DEV, batch=2, clip=5, lo=lo, hi=hi, heads=2, d=40, tokens=256, groups=8)


@pytest.mark.parametrize("kw", [dict(rows=32768, c=320, o=320), dict(rows=8192, c=640, o=5120, geglu=True), dict(rows=2048, c=1280, o=3840),
                                dict(rows=4100, c=320, o=2560, geglu=True, n_res=2, mean_shift=2.0), dict(rows=512, c=1280, o=1280),
                                dict(rows=1000, c=640, o=640, tile_cfg=212222), dict(rows=1000, c=640, o=640, tile_cfg=244222)])
def test_gemm_layernorm_fusion(kw):
    print(KC.case_gemm_ln(DEV, **kw))


def test_launch_stream_follows_torch_current_stream_split_k():
    """kernels._stream reads PyTorch's current stream of the operand's device PER LAUNCH, and the split-K scratch is keyed on that same
    handle.  Two side streams each run a split-K GEMM (fp32 slabs + reduce kernel) on their own operands, repeatedly and
    concurrently: with a stale cached stream handle the launches of stream B would land on stream A -- racing B's operand producer --
    and with a shared scratch the two reduce kernels would sum each other's slabs."""
    torch.manual_seed(3)
    rows, k, o = 512, 2560, 1280  # long K, few rows: the chooser splits K (forced below as well)
    w = (torch.randn(o, k, device=DEV) * 0.03).half()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs, refs = {}, {}
    for rep in range(4):
        for i, s in enumerate((s1, s2)):
            with torch.cuda.stream(s):
                x = (torch.randn(rows, k, device=DEV) * (1 + i)).half()  # produced ON the side stream
                assert K._raw_stream(0) == s.cuda_stream
                outs[(rep, i)] = K.gemm(x, w, split_k=4)
                refs[(rep, i)] = x
    torch.cuda.synchronize()
    assert len({k_[1] for k_ in K._ws}) >= 2, "each stream must have got its own split-K scratch"
    for key, y in outs.items():
        want = refs[key].float() @ w.float().t()
        err = float((y.float() - want).abs().max()) / float(want.abs().max())
        assert err < 4e-3, (key, err)
    # back on the default stream the default handle is used again
    assert K._raw_stream(0) == torch.cuda.current_stream().cuda_stream
