"""Host orchestration (pipeline loops, controller plans, arena bookkeeping, step reversal) checked end to end on the CPU
emulation backend against the unmodified reference's outputs -- small-latent scenarios only (the 64x64-latent blend
scenarios run on the MI355X, tests/test_pipeline_gpu.py)."""
import pytest
import torch

from fatezero_amd import _native, build

import pipeline_cases as PC


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


# (pipe_small_replace: oracle + MI355X suites only, 37 s here)
@pytest.mark.parametrize("name", ["pipe_small_refine_reweight", "pipe_f3_mid_next", "pipe_f4_prev_first"])  # all 8 scenarios run on MI355X (test_pipeline_gpu.py)
def test_pipeline_small(name):
    res = PC.run_pipeline_case(name, "cpu")
    print(name, res)
    PC.check(res)


@pytest.mark.parametrize("name", ["pipe_f3_mid_next"])   # (pipe_small_refine_reweight: under the issue plans, tests/test_issue_plan_emu.py)
def test_disk_store_spill_tier_is_bit_identical(name, monkeypatch):
    """disk_store=True (attention_store.py:103-108: a .pt file per step in the reference) with an HBM budget of 0: every step behind the first
    is captured into the 2-slab staging ring, copied to the host tier, and comes back for the edit -- one still in its slab, the others by
    the H2D path with the next step prefetched.  Same kernels on the same bytes: the whole job must reproduce the resident run bit for bit."""
    from fatezero_amd.video_diffusion.prompt_attention import attention_store as AS
    base, pipe0 = PC.run_pipeline_case(name, "cpu", return_pipe=True)
    maps0 = [{k: [m.clone() for m in v] for k, v in st.items()} for st in pipe0.store_controller.attention_store_all_step]
    edited0 = pipe0.last_edited_latents
    monkeypatch.setenv("FZ_ARENA_HBM_GB", "0")
    monkeypatch.setattr(AS, "SPILL_RING", 2)
    res, pipe = PC.run_pipeline_case(name, "cpu", return_pipe=True, disk_store=True)
    store = pipe.store_controller
    T = len(store.attention_store_all_step)
    assert sorted(store.arena.spilled) == list(range(1, T)), (sorted(store.arena.spilled), T)   # the first step sizes the slab: resident
    assert store.arena.fetch_stats["h2d"] >= T - 3 and store.arena.fetch_stats["hits"] >= 1, store.arena.fetch_stats
    assert torch.equal(pipe.last_edited_latents, edited0)
    assert res == base, (res, base)
    for s, (st, st0) in enumerate(zip(store.attention_store_all_step, maps0)):   # the reference-shaped views: host copies of the spilled steps
        for k in st0:
            assert len(st[k]) == len(st0[k])
            for a, b in zip(st[k], st0[k]):
                assert torch.equal(a.cpu(), b.cpu()), (s, k)
    PC.check(res)


def test_unet_tiny16_conv1d_reference_golden_emu():
    # configs WITHOUT a `lora` key: the plain nn.Conv1d temporal convolution (resnet.py:42-55) with non-identity recorded weights,
    # on fz_temporal_conv3 (MFMA form for the wide layers, the direct kernel for conv_out's 4 channels)
    r = PC.run_unet_golden("unet_tiny16_conv1d", "cpu")
    print(r)
    assert r["err"] <= 1.5e-2 * r["scale"], r


def test_unet_tiny40_reference_golden_emu():
    # head dims 40 / 80 / 160 (the log2-folded flash path) on a vector recorded from the unmodified reference UNet
    r = PC.run_unet_golden("unet_tiny40_default", "cpu")
    print(r)
    assert r["err"] <= 1.5e-2 * r["scale"], r


def test_unet_tiny40_layernorm_fusion_emu(monkeypatch):
    # same vector with LayerNorm folded into the surrounding GEMMs (fz_gemm_ln; off by default, attention.py LN_FUSION)
    from fatezero_amd.video_diffusion.models import attention as A
    monkeypatch.setattr(A, "LN_FUSION", True)
    r = PC.run_unet_golden("unet_tiny40_default", "cpu")
    print(r)
    assert r["err"] <= 1.5e-2 * r["scale"], r


@pytest.mark.skipif(__import__("os").environ.get("FZ_FULL_PARITY") != "1", reason="5 minutes of emulation: opt in with FZ_FULL_PARITY=1")
def test_whole_job_with_every_window_toggling_mini_emu():
    """(opt-in) The harness of the long-clip / window-transition cases (pipeline_cases.run_geometry_case: cfg3 / cfg4 / cfg5 / full-width cfg2 on
    MI355X) on a miniature: tiny16, 2 frames, 64^2 latents, T = 6 -- cross-replace, self-replace and latent-blend windows all open and
    close inside the run; per-step latents, masks, applied masks vs the CPU oracle."""
    res = PC.run_geometry_case("mini_emu", "cpu")
    print(res)
    PC.check_geometry(res)
