"""Boundary contract on the CPU emulation backend: foreign (tensor-protocol) controllers, edit_type None / 'save', and the
PRODUCT's host-side prompt constants against the values recorded from the unmodified reference."""
import pytest
import torch

from fatezero_amd import _native, build
from fatezero_amd.video_diffusion.prompt_attention import attention_util, ptp_utils, seq_aligner
from fatezero_amd.video_diffusion.prompt_attention.spatial_blend import SpatialBlender

from helpers import ReplayTokenizer, load_json
import protocol_cases as PR


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


def test_product_host_constants_exact():
    """SURVEY §8 a-14 ("must be bit-exact"), on fatezero_amd.video_diffusion.prompt_attention.* (the oracle's copies are pinned
    by test_oracle_golden.py::test_host_constants_exact against the same file)."""
    tok = ReplayTokenizer()
    gold = load_json("host_constants.json")
    assert len(gold) >= 6
    for name, c in gold.items():
        prompts, T = c["prompts"], c["T"]
        for key, inds in c["word_inds"].items():
            p, w = key.rsplit("|", 1)
            assert ptp_utils.get_word_inds(p, w, tok).tolist() == inds, (name, key)
        crs = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c["cross_replace_steps"].items()}
        a = ptp_utils.get_time_words_attention_alpha(prompts, T, crs, tok)
        assert tuple(a.shape) == (T + 1, 1, 1, 1, 77)
        assert a.reshape(T + 1, 77).to(torch.uint8).tolist() == c["cross_replace_alpha"], name
        if c["is_replace"]:
            assert seq_aligner.get_replacement_mapper(prompts, tok)[0].tolist() == c["replacement_mapper"], name
        else:
            mp, al = seq_aligner.get_refinement_mapper(prompts, tok)
            assert mp[0].tolist() == c["refinement_mapper"], name
            assert al[0].tolist() == c["refinement_alphas"], name
        if c["eq_params"] is not None:
            eq = attention_util.get_equalizer(prompts[1], c["eq_params"]["words"], c["eq_params"]["values"], tok)
            assert eq[0].tolist() == c["equalizer"], name
        if c["blend_words"] is not None:
            bl = SpatialBlender(prompts, c["blend_words"], tokenizer=tok, NUM_DDIM_STEPS=T)
            assert bl.alpha_layers.reshape(2, 77).tolist() == c["alpha_layers"], name
        # the constants as the controller folds them for the kernels: num_self_replace (attention_util.py:195-197)
        ctrl = attention_util.make_controller(tok, prompts, c["is_replace"], crs, 0.5, NUM_DDIM_STEPS=T)
        assert ctrl.num_self_replace == (0, int(T * 0.5)), name


@pytest.mark.parametrize("frames", [1, 3, 10])
def test_blend_mask_png_dumps_off_loop(tmp_path, frames):
    """Row (f)-3, spatial_blend.py:43-55: with `save_path` every blender call leaves `.../<prompt_choose>/step_in_store_NNNN/
    mask_<timestamp>_<count>.png` holding torchvision's save_image(normalize=True) picture of the native mask (grid of 8 columns,
    2 px padding; a single frame is written bare) -- queued to the writer thread, on disk after flush_mask_dumps()."""
    import pipeline_cases as PC
    from fatezero_amd.video_diffusion.prompt_attention import spatial_blend
    tok = ReplayTokenizer()
    c = load_json("host_constants.json")["teaser_posche"]
    g = torch.Generator().manual_seed(frames)
    store = {"down_cross": [(torch.rand(frames, 2, 16, 77, generator=g) ** 6) for _ in range(4)],
             "up_cross": [(torch.rand(frames, 2, 16, 77, generator=g) ** 6) for _ in range(3)]}
    att = SpatialBlender(c["prompts"], c["blend_words"], tokenizer=tok, NUM_DDIM_STEPS=4, th=(0.7, 0.7), prompt_choose="source",
                         save_path=str(tmp_path / "attention_blend_mask"))
    for step in (0, 1):
        for hw in (8, 8, 4):   # several layers of one step ask for the same (cached) mask: one D2H copy, one file per call
            att(store, step_in_store=step, target_h=hw, target_w=hw)
    spatial_blend.flush_mask_dumps()

    class Ctrl:
        attention_blend, latent_blend = att, None
    assert PC.check_mask_dumps(str(tmp_path), Ctrl) == 6
    assert 0 < sum(float(m.sum()) for m in att.mask_list) < sum(m.numel() for m in att.mask_list)


def test_foreign_store_controller_inversion():
    r = PR.foreign_store_inversion("cpu")
    print(r)
    PR.check_foreign_store(r)


def test_foreign_edit_controller():
    r = PR.foreign_edit("cpu")
    print(r)
    PR.check_foreign_edit(r)


def test_edit_type_none_and_save():
    r = PR.edit_type_none_and_save("cpu")
    print(r)
    PR.check_none_save(r)


def test_arena_that_does_not_fit_says_so(monkeypatch):
    """A job whose map arena exceeds the device memory fails with a message that names the size and the ways out, not a bare OOM."""
    import torch
    from fatezero_amd.video_diffusion.prompt_attention.attention_store import MapArena
    real_empty = torch.empty

    def empty(*a, **k):
        if k.get("dtype") == torch.uint8 and a and isinstance(a[0], int) and a[0] > 1 << 30:
            raise torch.OutOfMemoryError("HIP out of memory (simulated)")
        return real_empty(*a, **k)
    monkeypatch.setattr(torch, "empty", empty)
    MapArena._pool.clear()
    arena = MapArena()
    arena.first_step_done, arena.step_bytes = True, 6 * (1 << 30)
    with pytest.raises(RuntimeError, match=r"arena of this job needs 3\d\d\.\d GB"):
        arena.reserve(49, "cpu")
