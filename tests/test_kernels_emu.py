"""Kernel index math / fragment layouts / masking checked WITHOUT a GPU: the kernel sources are compiled against the
CPU emulation of csrc/fz_rt.h (libfatezero_emu.so) and run on small shapes.  This is test infrastructure -- the product
only ever loads libfatezero_hip.so; the MI355X versions of these checks live in tests/test_kernels_gpu.py."""
import pytest
import torch

from fatezero_amd import _native, build
from fatezero_amd import kernels as K

import kernel_cases as KC


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    import os
    # FZ_EMU_LIB: an emulator build of a trial variant of the kernels (scripts/emu_variant.sh), validated before it costs GPU time
    _native.use_test_backend(os.environ.get("FZ_EMU_LIB") or build.build_emu())
    yield
    _native.reset_backend()


DEV = "cpu"


@pytest.mark.parametrize("d", [16, 40, 80, 160])
def test_self_flash(d):
    KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=d, lq=64, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH)


def test_self_flash_multi_tile_and_masking():
    # lq=200: partial query block, key padding inside every kv slot (200 -> 256), three kv slots
    KC.case_attn_self(DEV, batch=1, clip=3, heads=1, d=32, lq=200, index_list=[-1, "mid", 1], mode=K.FZ_ATTN_FLASH)


def test_self_flash_two_query_blocks_per_wave():
    # lq >= 512 at d=40 selects the QB=2 variant (each wave owns two 32-row query blocks); 600 is not a multiple of 256
    KC.case_attn_self(DEV, batch=1, clip=2, heads=1, d=40, lq=600, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH)


@pytest.mark.parametrize("lq,qk_scale,shape", [(64, 1.5, None), (600, 1.5, None), (600, 6.0, None), (576, 3.0, "ramp"),
                                               (320, 3.0, "negative")])
def test_self_flash_log2_folded_q(lq, qk_scale, shape):
    # d=40 with q delivered in the log2 domain: the running max rides in contraction slot 40 of the QK^T MFMA
    KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=40, lq=lq, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH,
                      qk_scale=qk_scale, shape=shape, fold=True)


@pytest.mark.parametrize("lq,index_list,shape", [(576, ["mid"], "ramp"), (600, ["mid"], None), (640, [-1, "mid", 1], "ramp")])
def test_self_flash_pair_ring_odd_tiles(lq, index_list, shape):
    # the d=40 log2-domain variant meets at a barrier every SECOND 64-key tile (4-stage K/V ring): odd tile counts (9, 10
    # with a ragged last tile whose padded keys are neutralised in the stash, 30 over three kv slots)
    KC.case_attn_self(DEV, batch=1, clip=3, heads=1, d=40, lq=lq, index_list=index_list, mode=K.FZ_ATTN_FLASH,
                      qk_scale=3.0, shape=shape, fold=True)


@pytest.mark.parametrize("batch,clip,heads,lq,index_list", [(2, 3, 8, 64, [-1, "first"]), (1, 3, 16, 100, [-1, "first"]),
                                                             (2, 4, 8, 64, [-1, "first", "first"]), (1, 8, 8, 64, [-1, "first"])])
def test_self_flash_coinciding_kv_slots(batch, clip, heads, lq, index_list):
    # frames 0 and 1 of a clip see frame 0 in both slots: the kernel reads each distinct source once (same softmax), and -- when every
    # XCD owns whole heads (8 | heads) -- the launcher dispatches the full-length frames first; the oracle lists the keys twice
    KC.case_attn_self(DEV, batch=batch, clip=clip, heads=heads, d=40, lq=lq, index_list=index_list, mode=K.FZ_ATTN_FLASH, fold=True)
    KC.case_attn_self(DEV, batch=batch, clip=clip, heads=heads, d=80, lq=lq, index_list=index_list, mode=K.FZ_ATTN_FLASH)


def test_self_flash_more_frames_than_the_dispatch_order_lists():
    # 72 frames in one launch (9 clips of 8): beyond the 64 entries of the launcher's frame order -> plain group order, same results
    KC.case_attn_self(DEV, batch=9, clip=8, heads=8, d=40, lq=32, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH, fold=True)


def test_self_capture_and_inject_log2_folded_q():
    KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=40, lq=64, index_list=[-1, "first"], mode=K.FZ_ATTN_CAPTURE,
                      fold=True)
    KC.case_attn_self(DEV, batch=2, clip=2, heads=2, d=40, lq=64, index_list=["mid"], mode=K.FZ_ATTN_INJECT,
                      mask_kind="random", fold=True)


def test_self_own_frame_only():
    KC.case_attn_self(DEV, batch=2, clip=2, heads=2, d=64, lq=64, index_list=[], mode=K.FZ_ATTN_FLASH)


@pytest.mark.parametrize("d,lq", [(40, 64), (80, 144), (160, 36)])
def test_self_capture(d, lq):
    KC.case_attn_self(DEV, batch=1, clip=2, heads=2, d=d, lq=lq, index_list=[-1, "first"], mode=K.FZ_ATTN_CAPTURE)


@pytest.mark.parametrize("d,lq,index_list", [(80, 72, [-1, "first"]), (160, 36, ["mid"])])
def test_self_capture_inject_three_frames(d, lq, index_list):
    # clip = 3: distinct source frames per K/V slot at the head dims of the 32^2 / 16^2 levels
    KC.case_attn_self(DEV, batch=1, clip=3, heads=1, d=d, lq=lq, index_list=index_list, mode=K.FZ_ATTN_CAPTURE, seed=3)
    KC.case_attn_self(DEV, batch=2, clip=3, heads=1, d=d, lq=lq, index_list=index_list, mode=K.FZ_ATTN_INJECT,
                      mask_kind="random", seed=4)


@pytest.mark.parametrize("mask_kind", [None, "random", "rows"])
def test_self_inject(mask_kind):
    KC.case_attn_self(DEV, batch=2, clip=2, heads=2, d=40, lq=64, index_list=["mid"], mode=K.FZ_ATTN_INJECT,
                      mask_kind=mask_kind)


def test_self_inject_unaligned():
    KC.case_attn_self(DEV, batch=2, clip=2, heads=1, d=32, lq=81, index_list=[-1, "first"], mode=K.FZ_ATTN_INJECT,
                      mask_kind="random")


@pytest.mark.parametrize("mode", [K.FZ_ATTN_FLASH, K.FZ_ATTN_CAPTURE, K.FZ_ATTN_INJECT])
@pytest.mark.parametrize("d", [40, 160])
def test_cross(mode, d):
    KC.case_attn_cross(DEV, batch=2, clip=2, heads=2, d=d, lq=80, mode=mode)


def test_temporal():
    KC.case_attn_temporal(DEV, batch=2, clip=3, heads=2, d=40, tokens=10)
    # clip lengths 8 / 16: register-resident scores, odd tokens' rows rotated in LDS (C = 320: 40 chunks per row)
    KC.case_attn_temporal(DEV, batch=1, clip=8, heads=8, d=40, tokens=7)
    KC.case_attn_temporal(DEV, batch=1, clip=16, heads=4, d=16, tokens=5)
    # 24 / 32 frames (BASELINE cfg4 / cfg5): register-resident forms; 20 frames: the generic LDS form
    KC.case_attn_temporal(DEV, batch=1, clip=32, heads=8, d=40, tokens=5)
    KC.case_attn_temporal(DEV, batch=2, clip=24, heads=2, d=16, tokens=9)
    KC.case_attn_temporal(DEV, batch=1, clip=20, heads=2, d=16, tokens=6)


@pytest.mark.parametrize("span,c,groups", [(2, 80, 16), (1, 64, 8), (3, 320, 32)])
def test_groupnorm(span, c, groups):
    KC.case_groupnorm(DEV, n=span * 2, span=span, tokens=100, c=c, groups=groups, silu=True)
    KC.case_groupnorm(DEV, n=span, span=span, tokens=37, c=c, groups=groups, silu=False, eps=1e-6)


def test_layernorm_geglu_transpose_latent():
    KC.case_layernorm(DEV, rows=11, c=320)
    KC.case_layernorm(DEV, rows=5, c=1280)
    KC.case_geglu(DEV, rows=7, inner=128)
    KC.case_transpose_pad(DEV, n=2, l=77, c=80, lp=96)
    KC.case_transpose_pad(DEV, n=1, l=100, c=40, lp=128)
    KC.case_latent_update(DEV, frames=2, hw=64, blend=False, cfg=False)
    KC.case_latent_update(DEV, frames=3, hw=100, blend=True, cfg=True)


@pytest.mark.parametrize("res,out_hw,prompts,or_first", [(16, (32, 32), 1, False), (16, (8, 8), 1, False),
                                                         (16, (64, 64), 2, True), (18, (36, 36), 1, False)])
def test_blend_mask_bit_exact(res, out_hw, prompts, or_first):
    KC.case_blend_mask(DEV, prompts=prompts, frames=2, heads=2, res=res, out_hw=out_hw, or_first=or_first)


@pytest.mark.parametrize("kw", [dict(n=2, h=8, w=8, cin=32, cout=64), dict(n=2, h=6, w=10, cin=64, cout=40, with_temb=True, fpb=2),
                                dict(n=1, h=8, w=8, cin=96, cout=128, stride=2, with_res=True),
                                dict(n=2, h=4, w=4, cin=64, cout=64, upsample=True, with_temb=True, with_res=True)])
def test_conv3x3(kw):
    KC.case_conv3x3(DEV, **kw)


def test_conv3x3_small_cin_and_cout():
    KC.case_conv3x3(DEV, n=2, h=8, w=8, cin=4, cout=32, with_temb=True, fpb=2)      # conv_in: direct convolution
    KC.case_conv3x3(DEV, n=2, h=8, w=8, cin=32, cout=4)                             # conv_out: 4 of 64 tile rows live


@pytest.mark.parametrize("tile_cfg,split_k", [(254222, 1), (254122, 1), (244222, 1), (224223, 1), (222222, 1), (212222, 1), (222222, 3), (254222, 2), (254122, 4), (158122, 1),
                                              (254218, 1), (244218, 1), (254218, 4), (244218, 2), (252222, 1), (252222, 2), (252218, 1), (252218, 2)])
def test_conv3x3_every_tile_shape(tile_cfg, split_k):
    # (the ping-pong tiles ..18 have no ragged-K path: Cin a multiple of 32 there)
    KC.case_conv3x3(DEV, n=2, h=7, w=9, cin=96 if tile_cfg % 100 == 18 else 72, cout=48, with_temb=True, with_res=True, fpb=2, tile_cfg=tile_cfg,
                    split_k=split_k)


@pytest.mark.parametrize("tile_cfg", [0, 254222, 254122, 158122, 244222, 224223, 222222, 212222, 254218, 244218, 252222, 252218])
def test_gemm_tile_shapes(tile_cfg):
    KC.case_gemm(DEV, rows=300, k=96, o=136, n_res=2, tile_cfg=tile_cfg)


@pytest.mark.parametrize("k", [32, 64, 96, 128, 160, 224, 256, 320])
def test_gemm_pingpong_tile_counts(k):
    # 1 .. 10 K tiles of the ping-pong loop (prologue / steady-state / tail forms of its ring); both tile widths
    KC.case_gemm(DEV, rows=70, k=k, o=72, tile_cfg=254218)
    KC.case_gemm(DEV, rows=300, k=k, o=264, n_res=1, tile_cfg=244218)

@pytest.mark.parametrize("span,tokens,c,sets", [(1, 64, 1280, 8), (1, 300, 640, 4), (1, 256, 1280, 16), (2, 256, 960, 2), (8, 64, 1280, 2),
                                                (8, 100, 1280, 1), (8, 64, 2560, 2), (3, 37, 960, 1), (1, 1024, 640, 8)])
def test_groupnorm_one_launch_form(span, tokens, c, sets):
    # launches that take the one-launch form (csrc/norms.hip gn_fused_kernel: <= 20 channel pairs per thread, <= 2560 pairs-per-thread x
    # workgroups): every register bucket (2, 4, 5, 8, 10, 16, 20), the rule's boundary (512 workgroups x 5; 256 x 10), a ragged tail
    KC.case_groupnorm(DEV, n=span * sets, span=span, tokens=tokens, c=c, groups=32, silu=True)


def test_groupnorm_one_launch_form_edges():
    KC.case_groupnorm(DEV, n=2, span=2, tokens=50, c=16, groups=8, silu=True)       # 2 channels per group: one pair per row
    KC.case_groupnorm(DEV, n=2, span=1, tokens=33, c=24, groups=8, silu=False)     # 3 channels per group (odd): three-kernel form
    KC.case_groupnorm(DEV, n=4, span=2, tokens=7, c=2048, groups=16, silu=True)    # 128 channels per group: the LDS scale / shift bound
    KC.case_groupnorm_cat(DEV, n=2, span=2, tokens=20, c1=24, c2=40, groups=8)     # 8-channel groups, seam on a group boundary


def test_groupnorm_three_kernel_form_beyond_the_rule():
    KC.case_groupnorm(DEV, n=8, span=8, tokens=256, c=1280, groups=32, silu=True)    # 40 pairs per thread
    KC.case_groupnorm(DEV, n=16, span=1, tokens=1024, c=640, groups=32, silu=True)   # 10 pairs per thread x 512 workgroups
    KC.case_groupnorm_cat(DEV, n=4, span=4, tokens=1500, c1=320, c2=640, groups=32)  # 90 000 pairs per group, groups straddle the seam


def test_groupnorm_one_launch_form_of_a_lazy_concatenation():
    KC.case_groupnorm_cat(DEV, n=8, span=8, tokens=64, c1=1280, c2=1280, groups=32)   # seam on a group boundary
    KC.case_groupnorm_cat(DEV, n=16, span=8, tokens=64, c1=1280, c2=640, groups=32)   # 60-channel groups: group 21 straddles the seam
    KC.case_groupnorm_cat(DEV, n=4, span=1, tokens=256, c1=640, c2=320, groups=32)


def test_groupnorm_of_a_lazy_concatenation():
    KC.case_groupnorm_cat(DEV, n=4, span=2, tokens=50, c1=64, c2=32, groups=8)
    KC.case_groupnorm_cat(DEV, n=2, span=2, tokens=33, c1=320, c2=640, groups=32)   # groups straddle the seam (30 channels each)
    KC.case_groupnorm_cat(DEV, n=2, span=1, tokens=20, c1=24, c2=40, groups=8, silu=False)


@pytest.mark.parametrize("dma", ["late", "early"])
def test_igemm_trial_forms(dma, monkeypatch):
    """Trial forms of the ping-pong loop (-DFZ_IGEMM_TRIALS builds only: FZ_EMU_LIB=build_tmp/libemu_trials.so from scripts/emu_variant.sh),
    validated on the emulator under both DMA-landing models before they cost GPU time."""
    import os
    if not os.environ.get("FZ_EMU_LIB"):
        pytest.skip("needs an emulator build of the trial variants (FZ_EMU_LIB)")
    if dma == "early":
        monkeypatch.setenv("FZ_EMU_DMA", "early")
    for cfg in (4254218, 4244218, 4254118, 4158118, 1254218, 254118, 158118, 2254218, 3254218):
        KC.case_conv3x3(DEV, n=2, h=7, w=9, cin=96, cout=48, with_temb=True, with_res=True, fpb=2, tile_cfg=cfg, split_k=1)
        KC.case_conv3x3(DEV, n=2, h=7, w=9, cin=96, cout=48, fpb=2, tile_cfg=cfg, split_k=3)
        for k in (32, 64, 96, 128, 160, 224, 320):
            KC.case_gemm(DEV, rows=300, k=k, o=330, n_res=1, tile_cfg=cfg)


def test_gemm_forms():
    KC.case_gemm(DEV, rows=77, k=64, o=40, bias=False)
    KC.case_gemm(DEV, rows=130, k=320, o=320, n_res=1, ldx_extra=64, ldy_extra=24)          # strided x / y views
    KC.case_gemm(DEV, rows=96, k=32, o=128, lead=(2, 48))                                   # [B, L, K] input
    KC.case_gemm(DEV, rows=64, k=128, o=36, split_k=2, tile_cfg=212222)                       # split-K partial slabs
    KC.case_gemm(DEV, rows=2, k=1280, o=96)                                                 # time-embedding shaped


@pytest.mark.parametrize("tile_cfg", [0, 244222, 224223, 222222, 244218, 224212])
def test_gemm_geglu(tile_cfg):
    # (224212: the 128 x 256 tile with K step 32 and a 2-deep ring -- two workgroups per CU -- that carries the short-K GEGLU launches)
    KC.case_gemm(DEV, rows=70, k=64, o=256, geglu=True, tile_cfg=tile_cfg)
    KC.case_gemm(DEV, rows=33, k=64 if tile_cfg % 100 == 18 else 40, o=128, geglu=True, bias=False, tile_cfg=tile_cfg)
    if tile_cfg == 224212:
        KC.case_gemm(DEV, rows=300, k=320, o=512, geglu=True, tile_cfg=tile_cfg)  # 10 K steps, ragged row tile, two column tiles


@pytest.mark.parametrize("kw", [dict(n=2, l=64, k=64, c=64), dict(n=3, l=128, k=96, c=128, tile_cfg=212222),
                                dict(n=2, l=256, k=64, c=320, tile_cfg=254222), dict(n=2, l=256, k=64, c=320, tile_cfg=254122),
                                dict(n=1, l=192, k=40, c=128, tile_cfg=222222, ldx_extra=8), dict(n=2, l=64, k=72, c=128, tile_cfg=224223)])
def test_gemm_qkvt_one_launch(kw):
    """q | k | V^T of a self-attention in one launch: the transposed column tiles of every instantiated tile shape, ragged K, a strided
    x view -- vs fp32 torch and bit for bit vs the two launches it replaces."""
    r = KC.case_gemm_qkvt(DEV, **kw)
    assert r["qk_bit_equal"] and r["vt_bit_equal"], r


def test_gemm_qkvt_refuses_what_it_cannot_carry():
    x = torch.zeros(2, 72, 64, dtype=torch.float16)  # 72 tokens per frame: V^T rows would need zero padding to 128
    w = torch.zeros(192, 64, dtype=torch.float16)
    assert not K.gemm_qkvt_ok(x, w, 128)
    with pytest.raises(ValueError):
        K.gemm_qkvt(x, w, 128)
    with pytest.raises(RuntimeError):  # a tile whose width does not divide the k | v boundary
        K.gemm_qkvt(torch.zeros(2, 64, 64, dtype=torch.float16), w, 128, tile_cfg=244222)


@pytest.mark.parametrize("rows,o,tile_cfg,split_k", [(300, 136, 212222, 8), (512, 128, 212222, 2), (256, 136, 222222, 4), (700, 640, 254122, 4),
                                                     (130, 72, 212222, 16)])
def test_gemm_split_k_ragged_tile_counts(rows, o, tile_cfg, split_k):
    """Split-K projections with workgroup counts that are multiples of 8 run on the FLAT grid (csrc/igemm.hip: K slices -> XCDs): whole
    slices per XCD (split 8, 16), half / quarter slices (split 2, 4), ragged tile counts; the result must not depend on the mapping."""
    a = KC.case_gemm(DEV, rows=rows, k=1024, o=o, n_res=1, tile_cfg=tile_cfg, split_k=split_k)
    assert a["max_err"] < 4e-3 * 64


@pytest.mark.parametrize("n,h,w,cin,cout,tile_cfg,split_k", [(2, 8, 8, 128, 64, 212222, 8), (2, 16, 16, 64, 128, 222222, 2), (4, 8, 8, 96, 320, 254122, 4),
                                                             (1, 16, 8, 64, 640, 254222, 8)])
def test_conv3x3_split_k_slices_mapped_onto_xcds(n, h, w, cin, cout, tile_cfg, split_k):
    """The same for the convolutions, the launches the flat grid is shipped for (both K orders)."""
    KC.case_conv3x3(DEV, n=n, h=h, w=w, cin=cin, cout=cout, with_temb=True, with_res=True, fpb=n, tile_cfg=tile_cfg, split_k=split_k)


def test_k_group_tile_merges_its_two_halves():
    """Tile 252222 (csrc/igemm.hip IgCfg::KG): 320 x 128 as TWO K groups of 2 x 2 waves of 5 x 2 MFMA tiles -- each group contracts half of
    every K-64 step, the accumulators meet through LDS in two passes (7 + 3 tiles).  Full tiles, ragged rows / columns / K tails, every conv
    mode (stride 2, nearest-2x, chunk-outer K order), the temporal convolution and split-K slabs; the result is that of tile 254122 to the
    rounding of the other summation order."""
    KC.case_gemm(DEV, rows=384, k=320, o=640, n_res=1, tile_cfg=252222)
    KC.case_gemm(DEV, rows=200, k=72, o=328, n_res=2, tile_cfg=252222)
    KC.case_gemm(DEV, rows=130, k=1024, o=320, n_res=1, tile_cfg=252222, split_k=4)
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=128, cout=320, with_temb=True, with_res=True, fpb=2, tile_cfg=252222)
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=64, cout=96, stride=2, tile_cfg=252222)
    KC.case_conv3x3(DEV, n=1, h=8, w=8, cin=64, cout=64, upsample=True, tile_cfg=252222)
    # 252226: the same tile with the next tile's LDS-DMA pieces issued one by one between the MFMAs (FZ_KGSPREAD)
    KC.case_gemm(DEV, rows=200, k=72, o=328, n_res=2, tile_cfg=252226)
    KC.case_gemm(DEV, rows=130, k=1024, o=320, n_res=1, tile_cfg=252226, split_k=4)
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=128, cout=320, with_temb=True, with_res=True, fpb=2, tile_cfg=252226)
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=72, cout=96, stride=2, tile_cfg=252226)


@pytest.mark.parametrize("k", [32, 64, 96, 128, 160, 224, 256, 320])
def test_k_group_pingpong_tile_counts(k):
    """Tile 252218: the two K groups in ping-pong (csrc/igemm.hip KGPP: K tiles of 32, group g contracts sub-step g of every tile while the
    other group reads / fires LDS-DMA, one barrier per sub-step): 1 .. 10 K tiles walk the prologue / steady state / tail of its ring."""
    KC.case_gemm(DEV, rows=200, k=k, o=328, n_res=1, tile_cfg=252218)


@pytest.mark.parametrize("k", [32, 64, 96, 128, 160, 224, 320])
def test_loader_consumer_tile_counts(k):
    """Tile 252214 (csrc/igemm.hip FZ_LC): four consumer waves (fragment reads + MFMAs only) + four loader waves (every LDS-DMA piece): 1 .. 10
    K tiles walk the prologue / steady state / tail of its 4-slot ring."""
    KC.case_gemm(DEV, rows=200, k=k, o=328, n_res=1, tile_cfg=252214)


def test_loader_consumer_conv_modes():
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=128, cout=320, with_temb=True, with_res=True, fpb=2, tile_cfg=252214)
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=64, cout=96, stride=2, tile_cfg=252214)
    KC.case_conv3x3(DEV, n=1, h=8, w=8, cin=64, cout=64, upsample=True, tile_cfg=252214)
    KC.case_conv3x3(DEV, n=4, h=8, w=8, cin=96, cout=320, with_res=True, fpb=4, tile_cfg=252214, split_k=3)
    KC.case_gemm(DEV, rows=130, k=1024, o=320, n_res=1, tile_cfg=252214, split_k=4)
    KC.case_gemm(DEV, rows=384, k=320, o=640, n_res=1, tile_cfg=252214)


def test_conv3x3_with_the_pixel_halo_resident_in_lds():
    """csrc/conv_halo.hip (tile id 154299): the stride-1 3x3 convolution whose workgroups keep their pixel rows + halo in LDS across the nine
    taps (4 consumer waves, 2 weight-loader waves, 2 pixel-loader waves): one / several tiles per frame, image borders at every tile edge,
    W = 32 and 64, one and several Cin chunks (the halo double buffer), two channel tiles, time-embedding rows + residual."""
    KC.case_conv3x3(DEV, n=2, h=8, w=32, cin=64, cout=160, tile_cfg=154299)
    KC.case_conv3x3(DEV, n=2, h=16, w=32, cin=128, cout=320, with_temb=True, with_res=True, fpb=2, tile_cfg=154299)
    KC.case_conv3x3(DEV, n=1, h=8, w=64, cin=192, cout=160, with_res=True, tile_cfg=154299, seed=3)
    KC.case_conv3x3(DEV, n=4, h=8, w=32, cin=64, cout=320, with_temb=True, fpb=2, tile_cfg=154299, seed=4)   # two time-embedding rows, no residual
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=128, cout=160, with_temb=True, with_res=True, tile_cfg=154299, seed=5)   # W = 16: a tile is a frame
    # frames smaller than a tile (the 8 x 8 level): four whole frames per tile, each with its own halo; 8 x 16: two
    KC.case_conv3x3(DEV, n=4, h=8, w=8, cin=128, cout=160, with_temb=True, with_res=True, fpb=4, tile_cfg=154299, seed=6)
    KC.case_conv3x3(DEV, n=8, h=8, w=8, cin=64, cout=320, with_temb=True, fpb=4, tile_cfg=154299, seed=7)
    KC.case_conv3x3(DEV, n=2, h=8, w=16, cin=64, cout=160, with_res=True, tile_cfg=154299, seed=8)
    # weights the larger operand and whole tiles per XCD: the XCDs split along the channel tiles as well (csrc/conv_halo.hip, ChArgs::xa = 2)
    KC.case_conv3x3(DEV, n=8, h=16, w=16, cin=64, cout=320, with_temb=True, with_res=True, fpb=4, tile_cfg=154299, seed=9)
    with pytest.raises(Exception):   # shapes it does not carry are refused, not mangled
        KC.case_conv3x3(DEV, n=3, h=8, w=8, cin=64, cout=160, tile_cfg=154299)   # not whole tiles of four frames


def test_conv3x3_pixel_halo_narrow_tile():
    """conv_halo_kernel<9, 2> (tile id 154264): 64 output channels per workgroup -- whole and in K slices, time embedding + residual, W = 8 / 16 / 32,
    several channel tiles (the bias / time-embedding lanes and the staging stride follow the tile width)."""
    KC.case_conv3x3(DEV, n=1, h=16, w=16, cin=128, cout=64, with_temb=True, with_res=True, tile_cfg=154264)
    KC.case_conv3x3(DEV, n=4, h=8, w=8, cin=192, cout=192, with_temb=True, with_res=True, fpb=4, tile_cfg=154264, seed=1)
    KC.case_conv3x3(DEV, n=2, h=8, w=32, cin=256, cout=128, with_res=True, tile_cfg=154264, split_k=2, seed=2)
    KC.case_conv3x3(DEV, n=8, h=8, w=8, cin=320, cout=320, with_temb=True, fpb=8, tile_cfg=154264, split_k=3, seed=3)


def test_conv3x3_pixel_halo_in_k_slices():
    """The halo kernel under split-K: slice ks contracts its 64-channel chunks under all nine taps into an fp32 slab, the split-K tail kernel of
    igemm.hip sums the slabs and applies bias / time embedding / residual.  Even and ragged chunk counts per slice, odd first chunks (the halo
    double buffer starts on its second half), W = 16."""
    KC.case_conv3x3(DEV, n=1, h=16, w=16, cin=256, cout=160, with_temb=True, with_res=True, tile_cfg=154299, split_k=2)
    KC.case_conv3x3(DEV, n=2, h=8, w=32, cin=320, cout=320, with_res=True, fpb=2, tile_cfg=154299, split_k=3, seed=1)
    KC.case_conv3x3(DEV, n=1, h=16, w=16, cin=256, cout=160, tile_cfg=154299, split_k=4, seed=2)
    KC.case_conv3x3(DEV, n=8, h=8, w=8, cin=256, cout=160, with_temb=True, with_res=True, fpb=8, tile_cfg=154299, split_k=2, seed=3)


def test_conv3x3_up2_four_subpixel_convolutions():
    """fz_conv3x3_up2: nearest-2x + 3x3 as four 2x2 convolutions of the input on summed weights (conv_halo_kernel<4>): every output parity, image
    borders (the zero row / column of the UPSAMPLED image is the zero row / column of the input), W = 16 and 32, several chunks, two channel tiles."""
    KC.case_conv3x3_up2(DEV, n=1, h=16, w=16, cin=64, cout=160)
    KC.case_conv3x3_up2(DEV, n=2, h=8, w=32, cin=192, cout=320, seed=1)
    KC.case_conv3x3_up2(DEV, n=1, h=32, w=16, cin=128, cout=160, seed=2)   # two tiles per frame
    KC.case_conv3x3_up2(DEV, n=4, h=8, w=8, cin=128, cout=160, seed=3)     # four frames per tile
    assert not K.conv3x3_up2_ok(3, 8, 8, 64, 160)


def test_upsampler_module_takes_the_subpixel_form_and_caches_its_pack():
    """UpsamplePseudo3D through PseudoConv3d.forward_tokens: where fz_conv3x3_up2_preferred says so the launch is the four-2x2 form on a weight pack
    made ONCE (in the first forward that reaches the upsampler, whichever form that launch takes -- issue plans need packs to exist before a forward is
    recorded), dropped with the other packs by invalidation; FZ_NO_CONV_UP2's switch gives the nine-tap launch; both within the kernels' tolerance of
    torch's interpolate + conv2d."""
    import torch.nn.functional as F
    from fatezero_amd.video_diffusion.models import resnet as R
    torch.manual_seed(0)
    assert K.conv3x3_up2_preferred(16, 16, 16, 64, 320) and not K.conv3x3_up2_preferred(1, 16, 16, 64, 320)   # 128 workgroups / 8
    up = R.UpsamplePseudo3D(64, use_conv=True, out_channels=320).half()
    with torch.no_grad():
        up.conv.weight.mul_(0.3)
    x16 = R.Tokens(torch.randn(16, 256, 64).half(), 2, 8, 16, 16)
    x1 = R.Tokens(torch.randn(1, 256, 64).half(), 1, 1, 16, 16)

    @torch.no_grad()
    def ref(t):
        xi = F.interpolate(t.data.float().reshape(-1, 16, 16, 64).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
        y = F.conv2d(xi, up.conv.weight.float().reshape(320, 64, 3, 3), up.conv.bias.float(), padding=1)
        return y.permute(0, 2, 3, 1).reshape(t.data.shape[0], 1024, 320)
    assert up.conv._packed_up is None
    y1 = up.forward_tokens(x1)                       # 8 workgroups: the nine-tap launch -- and the pack is made all the same
    pack = up.conv._packed_up
    assert pack is not None and (y1.h, y1.w) == (32, 32)
    y16 = up.forward_tokens(x16)                     # 128 workgroups: the sub-pixel form, on the same pack
    assert up.conv._packed_up is pack
    for t, y in ((x1, y1), (x16, y16)):
        r = ref(t)
        assert (y.data.float() - r).abs().max() < 4e-3 * max(1.0, float(r.abs().max()))
    old = R.CONV_UP2
    try:
        R.CONV_UP2 = False
        y9 = up.forward_tokens(x16)
    finally:
        R.CONV_UP2 = old
    assert (y9.data.float() - y16.data.float()).abs().max() < 4e-3 * max(1.0, float(ref(x16).abs().max()))   # (the two forms round differently)


def test_k_group_pingpong_conv_modes():
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=128, cout=320, with_temb=True, with_res=True, fpb=2, tile_cfg=252218)
    KC.case_conv3x3(DEV, n=2, h=16, w=16, cin=64, cout=96, stride=2, tile_cfg=252218)
    KC.case_conv3x3(DEV, n=1, h=8, w=8, cin=64, cout=64, upsample=True, tile_cfg=252218)
    KC.case_conv3x3(DEV, n=4, h=8, w=8, cin=96, cout=320, with_res=True, fpb=4, tile_cfg=252218, split_k=3)
    KC.case_gemm(DEV, rows=130, k=1024, o=320, n_res=1, tile_cfg=252218, split_k=4)


def test_gemm_grouped_tile_order_covers_every_tile_once():
    """Launches of the 8-wave tiles with many weight panels take the 2-D grouped tile order (csrc/igemm.hip ig_launch: groups of group_b
    b-tiles, b fastest inside a group): 30 a-tiles x 6 b-tiles of the 128 x 256 tile -> groups of 4 and a ragged 2; the GEGLU form on the
    256 x 256 tile (20 x 3 tiles: b-tile fastest).  Every output element must still be right."""
    KC.case_gemm(DEV, rows=1500, k=64, o=3840, n_res=1, tile_cfg=224223)
    KC.case_gemm(DEV, rows=700, k=64, o=5120, geglu=True, tile_cfg=244222)


def test_groupnorm_statistics_from_the_producing_epilogue():
    """fz_temporal_conv3_gn / fz_gemm_gn on shapes whose launch is a 320-wide tile (the 64^2 level: 8 x 4096 rows), and on shapes whose
    launch is not (small / narrow: the call reports it and the plain launch ran)."""
    r = KC.case_gn_from_epilogue(DEV, n=8, clip=4, tokens=4096, cin=160, cout=320, producer="tconv")
    assert r is not None, "the 64^2 LoRA up convolution runs on a 320-wide tile: its epilogue must have written the statistics"
    r = KC.case_gn_from_epilogue(DEV, n=8, clip=2, tokens=4096, cin=320, cout=320, producer="gemm")   # proj_out of the 64^2 level
    assert r is not None
    assert KC.case_gn_from_epilogue(DEV, n=2, clip=2, tokens=128, cin=32, cout=320, producer="tconv") is None    # too few rows: another tile
    assert KC.case_gn_from_epilogue(DEV, n=2, clip=2, tokens=200, cin=32, cout=320, producer="gemm") is None     # 200 % 128 != 0
    assert KC.case_gn_from_epilogue(DEV, n=2, clip=2, tokens=128, cin=32, cout=96, groups=8, producer="gemm") is None  # 96 % 320 != 0


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
    """fz_gemm_lnout on the emulator: the 320 x 128 tile (rows <= 32768), the 320 x 256 ring tile, the ping-pong tile (K >= 640), ragged rows,
    rows with a large mean; shapes whose launch is not a whole-row tile report it and y is still right."""
    assert KC.case_gemm_lnout(DEV, rows=300, k=64, n_res=1, tile_cfg=254122) is not None
    assert KC.case_gemm_lnout(DEV, rows=600, k=64, n_res=2, bias=False, seed=1, tile_cfg=254222) is not None
    assert KC.case_gemm_lnout(DEV, rows=384, k=640, n_res=1, mean_shift=5.0, seed=2, tile_cfg=254218) is not None   # the ping-pong loop's tile
    assert KC.case_gemm_lnout(DEV, rows=100, k=64, n_res=1, seed=3, expect=False) is None     # the library picks a 64-wide tile: no LayerNorm from it
    assert KC.case_gemm_lnout(DEV, rows=256, k=64, n_res=1, seed=4, expect=False, tile_cfg=212222) is None
    assert KC.case_gemm_lnout(DEV, rows=100, k=512, n_res=1, seed=5, tile_cfg=212222, split_k=4, expect=False) is None   # split-K: reports it


def test_feed_forward_module_takes_the_chain_launch(monkeypatch):
    """models/attention.py FeedForward.apply at 320 channels: with fz_ff_chain preferred the module issues ONE launch and returns (y, the
    LayerNorm of y as a Prenormed); same numbers as the GEGLU + output-projection launches (their split-K choice at this tiny row count may
    round h differently: a few fp16 ulp)."""
    from fatezero_amd.video_diffusion.models import attention as A
    from fatezero_amd.video_diffusion.models.resnet import _NormParams
    torch.manual_seed(3)
    ff, norm = A.FeedForward(320), _NormParams(320)
    with torch.no_grad():
        ff.net[0].proj.bias.normal_(0, 0.2)
        ff.net[2].bias.normal_(0, 0.2)
        norm.weight.normal_(1, 0.2)
        norm.bias.normal_(0, 0.1)
    x = torch.randn(2, 64, 320).half()
    xn = torch.nn.functional.layer_norm(x.float(), (320,)).half()
    log = []
    real = K.ff_chain
    monkeypatch.setattr(K, "ff_chain", lambda *a, **k: (log.append(1), real(*a, **k))[1])
    monkeypatch.setattr(K, "ff_chain_preferred", lambda rows, c, inner: K.ff_chain_ok(rows, c, inner))
    y1, st1 = ff.apply(x, res=x, stats=A.Prenormed(xn), ln_next=norm)
    assert log == [1] and isinstance(st1, A.Prenormed)
    monkeypatch.setattr(A, "FF_CHAIN", False)
    y0, st0 = ff.apply(x, res=x, stats=A.Prenormed(xn), ln_next=norm)
    assert log == [1]
    scale = float(y0.float().abs().max())
    assert float((y1.float() - y0.float()).abs().max()) <= 2 * 2.0 ** -10 * scale
    ln0 = st0.t if isinstance(st0, A.Prenormed) else K.layernorm(y0, *norm.packed(x.device), eps=norm.eps)
    assert float((st1.t.float() - ln0.float()).abs().max()) <= 8 * 2.0 ** -10 * max(1.0, float(ln0.float().abs().max()))
    # a new weight set re-packs the stream
    ff2 = A.FeedForward(320)
    assert ff2._chain is None and ff._chain is not None


def test_feed_forward_chain_in_one_launch():
    """csrc/ff_chain.hip on the emulator: 128-row workgroups (whole, several, ragged last one), with / without bias, residual, LayerNorm; a
    small `inner` walks the stage ring's first / last iterations; bit-identical to fz_gemm(GEGLU) + fz_gemm_lnout."""
    r = KC.case_ff_chain(DEV, rows=128, inner=64)
    assert r["vs_two_launches"] == 0.0
    KC.case_ff_chain(DEV, rows=200, inner=96, seed=1, lead=(5, 40))
    KC.case_ff_chain(DEV, rows=40, inner=32, bias=False, res=False, ln=False, seed=2)
    KC.case_ff_chain(DEV, rows=300, inner=1280, seed=3)


def test_cross_attention_chain_in_one_launch():
    """csrc/xattn_chain.hip on the emulator: to_q -> 77-key cross-attention -> to_out + residual + LayerNorm in one launch, with and without
    attn1's output projection + norm2 in front; several frames / text contexts / workgroups per frame, fewer keys, no bias / LayerNorm;
    bit-identical to fz_gemm + fz_attn_cross + fz_gemm_lnout."""
    r = KC.case_xattn_chain(DEV, n=1, tokens=128)
    assert r["vs_launches"] == 0.0
    KC.case_xattn_chain(DEV, n=4, tokens=256, clip=2, seed=1)
    KC.case_xattn_chain(DEV, n=2, tokens=128, bias=False, ln=False, seed=2, lk=50)
    r = KC.case_xattn_chain(DEV, n=1, tokens=128, front=True, seed=3)
    assert r["vs_launches"] == 0.0 and r["y1_vs_launch"] == 0.0
    KC.case_xattn_chain(DEV, n=3, tokens=256, clip=2, front=True, seed=4)
    KC.case_xattn_chain(DEV, n=2, tokens=128, front=True, bias=False, ln=False, seed=5, lk=96)


def test_gemm_vt_pads_rows_beyond_a_short_context():
    """fz_gemm(transpose_out): columns [rows, rows_store) are zeros also when they lie in a column tile no input row falls into (a context of
    <= 64 keys padded to 96: the second 64-wide tile holds padding only)."""
    for l in (50, 56, 64):
        KC.case_gemm_vt(DEV, n=2, l=l, k=64, c=80, lp=96)


def test_transformer_block_takes_the_cross_attention_chain(monkeypatch):
    """models/attention.py SpatioTemporalTransformerBlock at 320 channels with more than 32 x 32 tokens per frame: with fz_xattn_chain preferred
    attn1's output projection, norm2, attn2 and norm3 are ONE launch (front form) -- or attn2 + norm3 alone (XATTN_CHAIN_FRONT off) -- and the
    block's output is that of the separate launches.  A controller still sees its layer calls in the reference's order."""
    from fatezero_amd.video_diffusion.models import attention as A
    from fatezero_amd.video_diffusion.models.resnet import Tokens
    torch.manual_seed(5)
    blk = A.SpatioTemporalTransformerBlock(320, 8, 40, cross_attention_dim=64).half()
    with torch.no_grad():
        for prm in blk.parameters():
            if prm.dim() == 1:
                prm.add_(torch.randn_like(prm) * 0.1)
        blk.attn_temporal.to_out[0].weight.normal_(0, 0.02)
    n, l = 2, 1152
    x = Tokens((torch.randn(n, l, 320) * 0.8).half(), 1, 2, 36, 32)
    ctx = torch.randn(1, 77, 64).half()

    class Ctl:  # the built-in controllers' planning interface: counts the layer calls, everything plain
        def __init__(self):
            self.calls = []

        def attention_plan(self, is_cross, place, n_frames, clip_len, heads, lq, lk, device):
            self.calls.append((is_cross, lq, lk))
            return A.AttnPlan(n_frames)

    log = []
    real = K.xattn_chain
    monkeypatch.setattr(K, "xattn_chain", lambda *a, **k: (log.append(k.get("front_eps") is not None), real(*a, **k))[1])
    monkeypatch.setattr(K, "xattn_chain_preferred", lambda rows, rpf, c, h, lk: K.xattn_chain_ok(rows, rpf, c, h, lk))
    monkeypatch.setattr(K, "ff_chain_preferred", lambda rows, c, inner: False)
    outs = []
    for front, chain in ((True, True), (False, True), (False, False)):
        monkeypatch.setattr(A, "XATTN_CHAIN_FRONT", front)
        monkeypatch.setattr(A, "XATTN_CHAIN", chain)
        ctl = Ctl()
        blk.attn1.controller = blk.attn2.controller = ctl
        blk.attn1.place_in_unet = blk.attn2.place_in_unet = "down"
        outs.append(blk.forward_tokens(x, ctx).data)
        assert ctl.calls == [(False, l, 2 * l), (True, l, 77)]
    assert log == [True, False]
    assert torch.isfinite(outs[0].float()).all()
    # (bit-identity with the separate launches is case_xattn_chain's subject, on the whole-row tile the 64x64 level takes; at this row count the
    #  separate launches get another tile and a stand-alone fz_layernorm, whose summation order differs from the epilogue's by an fp16 ulp)
    tol = 4 * 2.0 ** -10 * float(outs[2].float().abs().max())
    assert float((outs[0].float() - outs[2].float()).abs().max()) <= tol and float((outs[1].float() - outs[2].float()).abs().max()) <= tol
    # a controller that wants the maps of this call gets the separate launches
    monkeypatch.setattr(A, "XATTN_CHAIN_FRONT", True)
    monkeypatch.setattr(A, "XATTN_CHAIN", True)

    class Capt(Ctl):
        def attention_plan(self, is_cross, place, n_frames, clip_len, heads, lq, lk, device):
            self.calls.append((is_cross, lq, lk))
            if is_cross:
                p = torch.zeros(n_frames, heads, lq, K.CROSS_P_STRIDE, dtype=torch.float16)
                return A.AttnPlan(0, mode=K.FZ_ATTN_CAPTURE, p=p)
            return A.AttnPlan(n_frames)
    ctl = Capt()
    blk.attn1.controller = blk.attn2.controller = ctl
    y = blk.forward_tokens(x, ctx).data
    assert log == [True, False] and float((y.float() - outs[2].float()).abs().max()) <= tol and len(ctl.calls) == 2


def test_gemm_transposed_output():
    KC.case_gemm_vt(DEV, n=2, l=64, k=64, c=80, lp=64)
    KC.case_gemm_vt(DEV, n=3, l=77, k=64, c=40, lp=96)
    KC.case_gemm_vt(DEV, n=1, l=200, k=32, c=32, lp=256, tile_cfg=222222)


def test_lora_pair_one_launch():
    """fz_lora_pair (both temporal LoRA convolutions in one launch) on the emulator: bit-identical to fz_temporal_conv3 twice; clip
    lengths 1 ... 16 (tokens per workgroup 128 ... 8), every channel count of the UNet, with / without the time-embedding row and the
    second residual; and the shapes it refuses."""
    for kw in [dict(batch=1, clip=8, tokens=32, c=320, gn_groups=32), dict(batch=2, clip=4, tokens=64, c=640, with_temb=False, gn_groups=32),
               dict(batch=1, clip=16, tokens=16, c=320, with_res2=False, gn_groups=32), dict(batch=2, clip=1, tokens=256, c=320, gn_groups=32),
               dict(batch=1, clip=2, tokens=64, c=1280), dict(batch=1, clip=32, tokens=8, c=320, gn_groups=32),
               dict(batch=3, clip=2, tokens=192, c=320, gn_groups=32), dict(batch=1, clip=128, tokens=3, c=320)]:   # (one token per frame and workgroup)
        r = KC.case_lora_pair(DEV, **kw)   # gn_groups: + fz_lora_pair_gn's partials -> GroupNorm like the three-kernel form (span 1 and clip)
        assert r["bit_identical_to_two_launches"] and (not kw.get("gn_groups") or r["partial"] is not None)
    assert KC.case_lora_pair(DEV, batch=1, clip=2, tokens=64, c=1280, gn_groups=32)["partial"] is None   # 40 channels per group: no statistics form
    # 8 frames x 2048 tokens: 128 records per (frame, group), 1024 per clip -> still the one-wave finalize; x 4096 tokens -> the four-wave one
    assert KC.case_lora_pair(DEV, batch=1, clip=8, tokens=4096, c=320, gn_groups=32)["partial"] == (8, 32, 256, 3)
    assert not K.lora_pair_ok(6, 64, 320, 160, 3)      # 3 frames do not divide the 128-row tile
    assert not K.lora_pair_ok(8, 40, 320, 160, 8)      # 40 tokens: not a multiple of 128 / 8
    assert not K.lora_pair_ok(8, 64, 160, 160, 8)      # channels % 320
    assert not K.lora_pair_ok(8, 64, 320, 2, 8)        # conv_out's rank-2 pair stays on the direct kernel
    x = torch.zeros(6, 64, 320, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        K.lora_pair(x, torch.zeros(160, 3, 320, dtype=torch.float16), torch.zeros(320, 3, 160, dtype=torch.float16), clip_len=3)


def test_temporal_conv3():
    KC.case_temporal_conv3(DEV, batch=2, clip=3, tokens=20, cin=64, cout=32, with_res=False)
    KC.case_temporal_conv3(DEV, batch=1, clip=4, tokens=70, cin=32, cout=64, with_res=True)
    KC.case_temporal_conv3(DEV, batch=1, clip=1, tokens=16, cin=32, cout=40, with_res=True)
    # conv_out's channel counts (rank-2 LoRA pair 4 -> 2 -> 4, plain Conv1d 4 -> 4): the direct kernel; rows = bias / time embedding
    KC.case_temporal_conv3(DEV, batch=2, clip=3, tokens=50, cin=4, cout=2, with_res=False)
    KC.case_temporal_conv3(DEV, batch=2, clip=3, tokens=50, cin=2, cout=4, with_res=True)
    KC.case_temporal_conv3(DEV, batch=2, clip=4, tokens=33, cin=4, cout=4, with_res=True, with_rows=True)
    KC.case_temporal_conv3(DEV, batch=2, clip=2, tokens=40, cin=32, cout=32, with_res=True, with_rows=True)


@pytest.mark.parametrize("lo,hi", [(0, 2), (2, 4), (3, 5)])
def test_frame_shard_kernel_forms(lo, hi):
    # what a rank owning frames [lo, hi) of a 5-frame clip launches, against the single-GPU kernels on the whole clip
    KC.case_sharded_pieces(DEV, batch=2, clip=5, lo=lo, hi=hi, heads=2, d=40, tokens=64, groups=8)


@pytest.mark.parametrize("kw", [dict(rows=70, c=64, o=128), dict(rows=200, c=320, o=320, n_res=2),
                                dict(rows=96, c=128, o=256, geglu=True), dict(rows=130, c=320, o=640, tile_cfg=212222, mean_shift=2.0)])
def test_gemm_layernorm_fusion(kw):
    KC.case_gemm_ln(DEV, **kw)


def test_igemm_with_early_landing_dma(monkeypatch):
    """The emulator's LDS-DMA lands as LATE as the kernel's vmcnt waits allow by default (a too-loose fz_wait_vm<N>() reads stale LDS);
    FZ_EMU_DMA=early makes it land at issue instead (a DMA issued before every reader of the recycled ring stage passed the barrier
    corrupts their tile).  Every tile shape and K order of the implicit-GEMM kernel once more under the early model."""
    monkeypatch.setenv("FZ_EMU_DMA", "early")
    for tile_cfg, split_k in [(254222, 1), (254122, 1), (244222, 1), (224223, 1), (222222, 3), (212222, 1), (158122, 1), (254222, 2),
                              (254218, 1), (244218, 1), (254218, 4), (244218, 2)]:
        KC.case_conv3x3(DEV, n=2, h=7, w=9, cin=96 if tile_cfg % 100 == 18 else 72, cout=48, with_temb=True, with_res=True, fpb=2,
                        tile_cfg=tile_cfg, split_k=split_k)
    for tile_cfg in (0, 254222, 244222, 224223, 212222, 254218, 244218):
        KC.case_gemm(DEV, rows=300, k=96, o=136, n_res=2, tile_cfg=tile_cfg)
    for k in (32, 64, 96, 128, 160, 256):  # 1 .. 8 K tiles of the ping-pong loop (prologue / steady / tail forms)
        KC.case_gemm(DEV, rows=70, k=k, o=72, tile_cfg=254218)
    KC.case_temporal_conv3(DEV, batch=1, clip=4, tokens=70, cin=32, cout=64, with_res=True)
    KC.case_gemm_vt(DEV, n=3, l=77, k=64, c=40, lp=96)


def test_resnet_block_over_the_one_launch_lora_pair():
    """ResnetBlockPseudo3D at 320 channels through its three temporal-LoRA forms -- fz_temporal_conv3 twice, fz_lora_pair, fz_lora_pair_gn
    (the second GroupNorm and the next block's first one then normalise from the launch's partials) -- agree to an fp16 ulp of the output
    (the two-launch form may split K; the GroupNorm statistics round differently)."""
    from fatezero_amd.video_diffusion.models import lora as L
    from fatezero_amd.video_diffusion.models import resnet as R
    torch.manual_seed(0)
    blk = R.ResnetBlockPseudo3D(in_channels=320, out_channels=320, temb_channels=64, groups=32, model_config={"lora": 160}).half()
    for m in blk.modules():
        if isinstance(m, L.LoRALinearLayer):
            torch.nn.init.normal_(m.up.weight, std=0.05)   # (the reference initialises `up` with zeros: the pair would be skipped)
    for name, p in blk.named_parameters():
        if p.dim() == 4:
            torch.nn.init.normal_(p, std=(p.shape[1] * 9) ** -0.5)
    b, f, h, w = 1, 8, 4, 4
    x = R.Tokens(torch.randn(b * f, h * w, 320).half(), b, f, h, w)
    temb = torch.randn(b, 64).half()
    saved = (L.LORA_PAIR_FUSION, L.LORA_PAIR_ALWAYS, L.LORA_PAIR_GN)
    outs = {}
    try:
        for mode in ("two", "pair", "pair_gn"):
            L.LORA_PAIR_FUSION, L.LORA_PAIR_ALWAYS, L.LORA_PAIR_GN = mode != "two", True, mode == "pair_gn"
            o = blk.forward_tokens(x, temb)
            assert (o.gn is not None) == (mode == "pair_gn")   # the block's output carries partials for the NEXT GroupNorm
            outs[mode] = o.data.float()
    finally:
        L.LORA_PAIR_FUSION, L.LORA_PAIR_ALWAYS, L.LORA_PAIR_GN = saved
    scale = float(outs["two"].abs().max())
    for mode in ("pair", "pair_gn"):
        assert float((outs["two"] - outs[mode]).abs().max()) <= 2 * 2.0 ** -10 * max(1.0, scale), mode
