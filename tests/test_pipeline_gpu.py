"""MI355X end-to-end parity: native pipeline (HIP kernels through the C ABI) vs vectors from the unmodified reference."""
import pytest

from fatezero_amd import _native

import pipeline_cases as PC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_backend():
    _native.reset_backend()
    yield


@pytest.mark.parametrize("name", ["pipe_small_refine_reweight", "pipe_small_replace", "pipe_replace_blend",
                                  "pipe_refine_reweight_latentblend", "pipe_refine_noblend", "pipe_f4_prev_first",
                                  "pipe_f3_mid_next", "pipe_l72_replace_blend"])
def test_pipeline(name):
    res = PC.run_pipeline_case(name, "cuda", mixed_oracle=True)
    print(name, res)
    PC.check(res)
    assert _native.loaded_path().endswith("libfatezero_hip.so")


@pytest.mark.parametrize("name", ["pipe_replace_blend", "pipe_refine_reweight_latentblend"])
def test_blend_mask_png_dumps(name, tmp_path):
    """Row (f)-3 (spatial_blend.py:43-55): the edit run with `save_path` leaves one PNG per blender call whose decoded bits
    equal the native mask drawn the way torchvision's save_image(normalize=True) draws it; the copies and the encoding run off
    the denoise loop (side stream + writer thread), the parity of the run itself is unchanged."""
    res = PC.run_pipeline_case(name, "cuda", save_path=str(tmp_path))
    print(name, res)
    PC.check(res)
    assert res["mask_pngs_checked"] > 0


@pytest.mark.parametrize("variant", ["replace_blend", "refine_reweight_mid", "cfg2_8f"])
def test_fullwidth_sd15_pipeline_vs_oracle(variant):
    """BASELINE architecture at real width (d = 40 / 80 / 160, lora 160, 64x64 latents), THREE frames (two distinct K/V
    source frames, GroupNorm over three), 2 + 2 steps -- native HIP path vs oracle.OracleUNet / ddim_inversion / ddim_edit:
    cfg2's model config with the bench's controller (Replace + blend-masked self-attention), and cfg1 / cfg3's model config
    ({SparseCausalAttention_index: ['mid'], least_sc_channel: 640}) with Refine + Reweight.  `cfg2_8f`: the JUDGED launch shapes --
    F = 8 (one 8-frame inversion launch, one 16-frame CFG edit launch per layer: the tiles, flash dispatch order and one-launch
    GroupNorms bench.py's job takes), 1 + 1 steps, always with the all-fp32 leg.  The blend threshold is set so that 20-80 % of the
    mask rows keep the live attention (asserted).  FZ_FULL_PARITY=1 adds the all-fp32 edit run (oracle edit on the oracle's own
    maps) to the 3-frame cases as well."""
    import os
    res = PC.run_fullwidth_case("cuda", pure_edit=os.environ.get("FZ_FULL_PARITY") == "1", variant=variant)
    print("fullwidth", res)
    PC.check_fullwidth(res)
    assert _native.loaded_path().endswith("libfatezero_hip.so")


@pytest.mark.parametrize("frames,variant", [(16, "refine_reweight_mid"), (16, "replace_blend")])
def test_fullwidth_unet_forward_long_clips_vs_oracle(frames, variant):
    """BASELINE cfg3's clip length (16 frames: its own temporal-attention instantiation, GroupNorm over 16 frames, flash dispatch and
    sparse-causal sources by clip_len = 16) under both model configs: one full-width UNet forward vs oracle.OracleUNet."""
    r = PC.run_fullwidth_forward("cuda", F=frames, variant=variant)
    print("fullwidth forward", r)
    assert r["err"] <= 1.5e-2 * r["scale"] and r["err_q99"] <= 4e-3 * r["scale"], r
    assert _native.loaded_path().endswith("libfatezero_hip.so")


@pytest.mark.parametrize("name", ["unet_tiny40_default", "unet_tiny40_l72", "unet_tiny16_default", "unet_tiny16_mid", "unet_tiny16_conv1d"])
def test_unet_vs_reference_golden(name):
    r = PC.run_unet_golden(name, "cuda")
    print(name, r)
    assert r["err"] <= 1.5e-2 * r["scale"], r


def test_drift_50_steps():
    res = PC.run_drift_case("cuda")
    print("drift (max latent error / max |latent| at steps 10, 25, 50):", res)
    PC.check_drift(res)


def test_foreign_controllers_and_edit_types():
    import protocol_cases as PR
    r = PR.foreign_store_inversion("cuda")
    print(r)
    PR.check_foreign_store(r)
    r = PR.foreign_edit("cuda", L=64, blend=True)
    print(r)
    PR.check_foreign_edit(r)
    r = PR.edit_type_none_and_save("cuda")
    print(r)
    PR.check_none_save(r)
