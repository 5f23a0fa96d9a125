"""Host orchestration (pipeline loops, controller plans, arena bookkeeping, step reversal) checked end to end on the CPU
emulation backend against the unmodified reference's outputs -- small-latent scenarios only (the 64x64-latent blend
scenarios run on the MI355X, tests/test_pipeline_gpu.py)."""
import pytest

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
