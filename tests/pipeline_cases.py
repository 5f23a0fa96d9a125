"""End-to-end parity of the native pipeline (inversion with capture -> attention-fusion edit) against vectors recorded
from the UNMODIFIED reference pipeline (oracle/gen_golden.py: gen_pipeline).  Shared by the CPU-emulation suite and
the MI355X suite."""
import numpy as np
import torch

from helpers import ReplayTokenizer, load_json, load_npz, unpack_bits
from oracle.weights import procedural_state_dict

from fatezero_amd.video_diffusion.models import UNetPseudo3DConditionModel
from fatezero_amd.video_diffusion.pipelines.p2p_ddim_spatial_temporal import P2pDDIMSpatioTemporalPipeline
from fatezero_amd.video_diffusion.schedulers import DDIMScheduler

TINY = {
    "tiny16": dict(block_out_channels=(32, 64, 128, 128), norm_num_groups=8, cross_attention_dim=64, attention_head_dim=2),
    "tiny40": dict(block_out_channels=(80, 160, 320, 320), norm_num_groups=16, cross_attention_dim=64, attention_head_dim=2),
}


def build_unet(kind, model_config, device):
    unet = UNetPseudo3DConditionModel(sample_size=64, **TINY[kind], **model_config)
    shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    unet.load_state_dict(procedural_state_dict(shapes))
    return unet.half().to(device).eval()


def oracle_edit_on_native_maps(meta, consts, gz, store, tok):
    """Run the fp32 CPU oracle's edit pass on the maps / latents captured by the NATIVE inversion.  With identical
    (fp16) inversion maps on both sides the attention-blend masks must agree bit for bit, and the edited latents
    isolate the edit-pass arithmetic from upstream fp16 noise."""
    from oracle import fatezero_oracle as O
    shapes = load_json("unet_meta.json")["unet_tiny16_default"]["state_dict_shapes"]
    cfg = O.UNetConfig(**TINY["tiny16"], model_config=meta["model_config"])
    unet = O.OracleUNet(procedural_state_dict([(n, tuple(s)) for n, s in shapes]), cfg)
    ost = O.StoreController()
    ost.attention_store_all_step = [{k: [t.float().cpu() for t in v] for k, v in d.items()}
                                    for d in store.attention_store_all_step]
    ost.latents_store = [t.float().cpu() for t in store.latents_store]
    kw = meta["kwargs"]
    ctrl = O.make_edit_controller(
        tok, consts["prompts"], ost, meta["T"], kw["is_replace_controller"], dict(kw["cross_replace_steps"]),
        kw["self_replace_steps"], blend_words=kw.get("blend_words"), eq_params=kw.get("eq_params"),
        blend_th=tuple(kw["blend_th"]), blend_self_attention=kw.get("blend_self_attention", False),
        blend_latents=kw.get("blend_latents", False), save_self_attention=kw["save_self_attention"])
    sched = O.DDIMSchedule(meta["T"])
    edited = O.ddim_edit(unet, sched, torch.from_numpy(gz["zT"]), torch.from_numpy(gz["emb_tgt"]), ctrl,
                         guidance_scale=kw["guidance_scale"])
    return edited, ctrl


def _mask_flips(native_list, oracle_list):
    flips = total = 0
    assert len(native_list) == len(oracle_list), (len(native_list), len(oracle_list))
    for a, b in zip(native_list, oracle_list):
        a, b = a.bool().cpu(), b.bool().cpu()
        assert a.shape == b.shape, (a.shape, b.shape)
        flips += int((a != b).sum())
        total += a.numel()
    return flips, total


def check_mask_dumps(save_path, ctrl):
    """Row (f)-3: the blend-mask PNGs the reference writes from inside its hot loop (spatial_blend.py:43-55) -- here queued
    off-loop.  File k of a blender (`..._{k:02d}.png`, k = its call count) must decode to exactly the picture
    torchvision.utils.save_image(normalize=True) draws from the k-th native mask: frames on a grid of 8 columns, 2 px padding,
    white where the mask is 1.  The grid is rebuilt here independently with torch ops."""
    import glob
    import numpy as np
    from PIL import Image
    n_checked = 0
    for sub, blender in (("attention_blend_mask", ctrl.attention_blend), ("latent_blend_mask", ctrl.latent_blend)):
        if blender is None:
            continue
        files = sorted(glob.glob(f"{save_path}/{sub}/{blender.prompt_choose}/**/mask_*.png", recursive=True),
                       key=lambda f: int(f.rsplit("_", 1)[1].split(".")[0]))
        assert len(files) == len(blender.mask_list) == blender.count, (sub, len(files), len(blender.mask_list), blender.count)
        for k, f in enumerate(files):
            assert int(f.rsplit("_", 1)[1].split(".")[0]) == k
            if blender.prompt_choose == "source":
                assert "/step_in_store_" in f
            # 'source': the PNG shows mask_list's mask[0]; 'both': it shows mask[-1] = (source mask OR target mask)
            # (spatial_blend.py:39-41,49-50), which the blender keeps in dumped_mask_list -- compared bit for bit either way
            assert len(blender.dumped_mask_list) == blender.count
            m = blender.dumped_mask_list[k].float().cpu()          # [F, h, w] of 0 / 1
            if blender.prompt_choose == "source":
                assert torch.equal(m, blender.mask_list[k].float().cpu()[:, 0])
            else:
                assert bool((m >= blender.mask_list[k].float().cpu()[:, 0]).all())  # OR-ed with the source mask: a superset
            fr, h, w = m.shape
            lo, hi = float(m.min()), float(m.max())
            mn = (m - lo) / max(hi - lo, 1e-5)
            if fr == 1:
                grid = mn[0]
            else:
                cols = min(8, fr)
                rows = (fr + cols - 1) // cols
                grid = torch.zeros(rows * (h + 2) + 2, cols * (w + 2) + 2)
                for i in range(fr):
                    grid[(i // cols) * (h + 2) + 2:(i // cols) * (h + 2) + 2 + h, (i % cols) * (w + 2) + 2:(i % cols) * (w + 2) + 2 + w] = mn[i]
            want = grid.mul(255).add(0.5).clamp(0, 255).to(torch.uint8).numpy()
            got = np.asarray(Image.open(f))
            assert got.shape == want.shape + (3,), (f, got.shape, want.shape)
            assert (got == want[:, :, None]).all(), f
            n_checked += 1
    return n_checked


def run_pipeline_case(name, device, return_pipe=False, mixed_oracle=False, save_path=None, disk_store=False):
    """`disk_store=True`: the reference's switch of the same name (attention_store.py:103-108) -- here the arena's spill tier; the budget
    comes from FZ_ARENA_HBM_GB (the tests set 0: every step behind the first goes through the staging ring to the host tier and back)."""
    meta = load_json("pipeline_meta.json")[name]
    consts = load_json("host_constants.json")[meta["prompt_case"]]
    gz = load_npz(name + ".npz")
    unet = build_unet("tiny16", meta["model_config"], device)
    T = meta["T"]
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=ReplayTokenizer(), unet=unet,
                                         scheduler=DDIMScheduler(), disk_store=disk_store)
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler.set_timesteps(T)
    assert [int(t) for t in pipe.scheduler.timesteps] == meta["timesteps"]
    emb_src = torch.from_numpy(gz["emb_src"]).to(device)
    emb_tgt = torch.from_numpy(gz["emb_tgt"]).to(device)
    z0 = torch.from_numpy(gz["z0"]).to(device)
    lat_all = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1,
                                                 text_embeddings=emb_src, store_attention=True, LOW_RESOURCE=True,
                                                 latents=z0)
    res = {}
    zT_ref = torch.from_numpy(gz["zT"])
    res["inv_err"] = float((lat_all[-1].float().cpu() - zT_ref).abs().max())
    res["inv_scale"] = float(zT_ref.abs().max())
    store = pipe.store_controller
    m0 = store.attention_store_all_step[0]
    for key, lst in meta["map_shapes"].items():
        assert [list(t.shape) for t in m0[key]] == lst, (key, [list(t.shape) for t in m0[key]], lst)
    ref_map = torch.from_numpy(gz["inv_step0_down_cross2"]).float()
    res["map_err"] = float((m0["down_cross"][min(2, len(m0["down_cross"]) - 1)].float().cpu() - ref_map).abs().max())
    ref_map2 = torch.from_numpy(gz["inv_step0_mid_self0"]).float()
    res["self_map_err"] = float((m0["mid_self"][0].float().cpu() - ref_map2).abs().max())

    kw = dict(meta["kwargs"])
    kw.pop("save_path", None)
    if save_path is not None:
        kw["save_path"] = save_path
    if disk_store:
        kw["disk_store"] = True
    pipe._encode_prompt = lambda *a, **k: emb_tgt
    # start the edit from the reference's own inverted latent so that the two halves are checked independently
    out = pipe(latents=zT_ref.to(device), output_type="latent", **kw)
    edited = out["sdimage_output"].images.float().cpu()
    ref = torch.from_numpy(gz["edited"])
    res["edit_err"] = float((edited - ref).abs().max())
    res["edit_scale"] = float(ref.abs().max())
    if return_pipe:
        pipe.last_edited_latents = edited
    # robust view of the same comparison: a blend mask is a hard threshold, so ONE pixel whose normalised score sits within
    # fp16 noise of the threshold flips and moves that latent pixel by |x - inverted| (order of the latent scale itself)
    res["edit_err_q99"] = float(torch.quantile((edited - ref).abs().flatten(), 0.99))
    ctrl = pipe.last_edit_controller
    if save_path is not None:
        res["mask_pngs_checked"] = check_mask_dumps(save_path, ctrl)
    if ctrl.attention_blend is not None:
        packed = {}
        for m in ctrl.attention_blend.mask_list:
            packed.setdefault(m.shape[-1], []).append(m.bool().cpu())
        flips = total = 0
        for r, v in packed.items():
            got = torch.stack(v).numpy()
            want = unpack_bits(gz[f"attn_mask_r{r}_bits"], gz[f"attn_mask_r{r}_shape"])
            assert got.shape == want.shape, (got.shape, want.shape)
            flips += int((got != want).sum())
            total += got.size
        res["attn_mask_flips"], res["attn_mask_total"] = flips, total
    if ctrl.latent_blend is not None:
        got = torch.stack([m.cpu() for m in ctrl.latent_blend.mask_list]).bool().numpy()
        want = unpack_bits(gz["latent_mask_bits"], gz["latent_mask_shape"])
        assert got.shape == want.shape, (got.shape, want.shape)
        res["latent_mask_flips"], res["latent_mask_total"] = int((got != want).sum()), got.size
        # The recorded `mask_list` holds mask[0] (the source prompt's mask, spatial_blend.py:113-115); what blends the EDITED latents is
        # mask[1] = mask[0] OR the TARGET prompt's mask, thresholded from the LIVE cross maps of the edit pass -- no recording of the
        # reference exposes it.  A flipped pixel there moves that latent by |x - inverted| (bounded by the latent scale and by nothing
        # smaller), so against the recording the max is replaced by a COUNT: positions whose error leaves the band.
        emap = (edited - ref).abs().amax(dim=(0, 1))                                       # [F, h, w]
        res["edit_positions_beyond_band"] = int((emap > EDIT_TOL_VS_REFERENCE * res["edit_scale"]).sum())
        res["edit_positions"] = emap.numel()
        if "latent_applied_mask_bits" in gz:
            # the reference's APPLIED masks, recorded by a subclass hook in oracle/gen_golden.py (the reference run itself is
            # unchanged: every other array of the recording came out bit-identical): flips against them, and the max error away
            # from the flipped pixels and their 3x3 neighbourhood (the next UNet step's convolutions spread a jump)
            want_a = unpack_bits(gz["latent_applied_mask_bits"], gz["latent_applied_mask_shape"])
            got_a = torch.stack([m.bool().cpu() for m in ctrl.latent_blend.applied_mask_list]).numpy()
            assert got_a.shape == want_a.shape, (got_a.shape, want_a.shape)
            res["applied_mask_flips"], res["applied_mask_flips_total"] = int((got_a != want_a).sum()), got_a.size
            fl = torch.from_numpy(got_a != want_a).reshape(-1, *emap.shape).any(0)
            near = torch.nn.functional.max_pool2d(fl[None].float(), 3, 1, 1)[0].bool()
            res["edit_err_off_applied_flips"] = float(emap[~near].max())
    if mixed_oracle:
        o_edit, o_ctrl = oracle_edit_on_native_maps(meta, consts, gz, store, ReplayTokenizer())
        res["edit_err_vs_oracle_on_native_maps"] = float((edited - o_edit).abs().max())
        res["edit_err_vs_oracle_on_native_maps_q99"] = float(torch.quantile((edited - o_edit).abs().flatten(), 0.99))
        if ctrl.attention_blend is not None:
            res["attn_mask_flips_same_maps"], _ = _mask_flips(ctrl.attention_blend.mask_list, o_ctrl.attention_blend.mask_list)
        if ctrl.latent_blend is not None:
            res["latent_mask_flips_same_inv_maps"], _ = _mask_flips(ctrl.latent_blend.mask_list, o_ctrl.latent_blend.mask_list)
            # the APPLIED masks of the two runs (same inversion maps, live cross maps each its own) and the error away from their flips
            na, oa = ctrl.latent_blend.applied_mask_list, o_ctrl.latent_blend.applied_mask_list
            res["applied_mask_flips_same_inv_maps"], res["applied_mask_total"] = _mask_flips(na, oa)
            fl = torch.stack([(a.bool().cpu() != b.bool().cpu()).reshape(-1, *a.shape[-2:]) for a, b in zip(na, oa)]).any(0)
            assert fl.shape == edited.shape[2:], (fl.shape, edited.shape)
            near = torch.nn.functional.max_pool2d(fl[None].float(), 3, 1, 1)[0].bool()
            em = (edited - o_edit).abs().amax(dim=(0, 1))
            res["edit_err_vs_oracle_off_applied_flips"] = float(em[~near].max())
    return (res, pipe) if return_pipe else res


# Stated tolerances (fp16 storage / MFMA inputs with fp32 accumulation vs the fp32 reference, 4 DDIM steps each way).  Re-measured
# in round 3 on the all-native build (every kernel of the path is our own and bit-deterministic run to run, so the bands only
# cover the fp16 arithmetic itself -- profiles/r03_parity_numbers*.txt holds the measured values; each bound is ~1.5-2x the worst
# measured case):
LATENT_TOL = 1.0e-2      # inversion: max |latent error| / max |latent|          (measured 0.36 - 0.47 % on the 8 recordings)
EDIT_Q99_TOL = 1.25e-2   # edit vs the reference recording, 99th percentile      (measured 0.49 - 0.77 %)
EDIT_MAX_TOL = 2.5e-2    # edit vs the reference recording, max, no blend mask   (measured 1.13 - 1.40 %)
MAP_TOL = 2e-2           # captured cross maps, absolute on probabilities in [0,1], END TO END (fp16 q/k/activations upstream of
                         # a peaky softmax; measured 0.85e-2 - 1.37e-2); given identical q/k the kernels store P within 1.6 fp16
                         # ulp (kernel_cases.py)
SELF_MAP_TOL = 4e-3      # captured self-attention maps, absolute                (measured 0.3e-3 - 1.7e-3)
MASK_FLIP_TOL = 9e-3     # fraction of mask elements that may differ from the all-fp32 reference run: the mask thresholds a
                         # normalised score, and the captured fp16 maps carry ~1e-2 of upstream fp16 noise, so pixels sitting
                         # within ~1 % of the threshold flip.  Re-measured in round 4 (profiles/r04_parity_numbers.txt): 0.05 - 0.47 %
                         # on the recorded scenarios, 0.39 % at the judged 8-frame shape with a 67 %-ones mask; the bound is 2x the
                         # worst.  Given IDENTICAL captured maps (oracle edit run on the native inversion maps) the attention-blend
                         # masks are bit-exact: 0 flips, asserted as such.
EDIT_TOL_SAME_MAPS = 1.5e-2     # edit vs the oracle's edit on the natively captured maps, max   (measured 0.53 - 0.93 %)
EDIT_TOL_VS_REFERENCE = 6e-2    # edit vs the all-fp32 reference, MAX, when blend masks are in play: a flipped mask pixel moves
                                # that latent by |x - inverted| -- discrete, of the order of the latent scale (measured 1.7 - 4.7 %)


def check(res):
    assert res["inv_err"] <= LATENT_TOL * res["inv_scale"], res
    has_mask = "attn_mask_flips" in res or "latent_mask_flips" in res
    latent_blend = "latent_mask_flips" in res
    # the bulk of the latents (99th percentile) sits within the same band in every scenario, masks or not
    assert res["edit_err_q99"] <= EDIT_Q99_TOL * res["edit_scale"], res
    # the max: without masks every value is bounded; with masks single flipped pixels (counted and bounded below) may move
    if latent_blend:
        assert res["edit_positions_beyond_band"] <= MASK_FLIP_TOL * res["edit_positions"], res
        if "applied_mask_flips" in res:
            assert res["applied_mask_flips"] <= MASK_FLIP_TOL * res["applied_mask_flips_total"], res
            assert res["edit_err_off_applied_flips"] <= EDIT_TOL_VS_REFERENCE * res["edit_scale"], res
    else:
        assert res["edit_err"] <= (EDIT_TOL_VS_REFERENCE if has_mask else EDIT_MAX_TOL) * res["edit_scale"], res
    if "edit_err_vs_oracle_on_native_maps" in res:
        # the target-prompt half of a latent-blend mask is thresholded from the LIVE cross maps, which differ between the native
        # run and the oracle by fp16 noise even on identical inversion maps: the max may contain such a flip there
        assert res["edit_err_vs_oracle_on_native_maps_q99"] <= EDIT_Q99_TOL * res["edit_scale"], res
        assert res["edit_err_vs_oracle_on_native_maps"] <= (EDIT_TOL_VS_REFERENCE if latent_blend else EDIT_TOL_SAME_MAPS) * res["edit_scale"], res
    if "attn_mask_flips_same_maps" in res:
        assert res["attn_mask_flips_same_maps"] == 0, res
    if "latent_mask_flips_same_inv_maps" in res:
        assert res["latent_mask_flips_same_inv_maps"] <= MASK_FLIP_TOL * res["latent_mask_total"], res
        assert res["applied_mask_flips_same_inv_maps"] <= MASK_FLIP_TOL * res["applied_mask_total"], res
        assert res["edit_err_vs_oracle_off_applied_flips"] <= EDIT_TOL_SAME_MAPS * res["edit_scale"], res
    assert res["map_err"] <= MAP_TOL and res["self_map_err"] <= SELF_MAP_TOL, res
    for k in ("attn_mask", "latent_mask"):
        if k + "_flips" in res:
            assert res[k + "_flips"] <= MASK_FLIP_TOL * res[k + "_total"], res


# ------------------------------------------------------------------------------------------------------------
# Full SD-1.x width (the BASELINE cfg2 architecture: 320/640/1280/1280, 8 heads of d = 40/80/160, lora 160, 64x64 latents)
# ------------------------------------------------------------------------------------------------------------
SD15 = dict(block_out_channels=(320, 640, 1280, 1280), norm_num_groups=32, cross_attention_dim=768, attention_head_dim=8)
FULL_SRC = "a silver jeep driving down a curvy road in the countryside,"
FULL_TGT = "a Porsche car driving down a curvy road in the countryside,"


# Blend threshold of the full-width cases.  With procedural weights the normalised 16^2 blend-word score (spatial_blend.py:24-42:
# sum over words, mean over heads x layers, 3x3 max-pool, / per-frame max) spreads over 0.23 ... 1.0 with its median near 0.57
# (scripts/mask_struct_exp.py on the CPU oracle), so the teaser's th = 0.3 keeps 99 % of the rows live -- a mask test that tests
# nothing (round-3 review).  Scaling the blend words' context rows saturates the softmax (100 % ones); the YAML knob that DOES move the
# split is `blend_th` itself (the reference's configs use 0.3 and 2): 0.55 puts 20-80 % of the rows on either side, asserted below.
FULL_BLEND_TH = 0.55
FULL_MASK_BAND = (0.2, 0.8)

FULL_VARIANTS = {
    # BASELINE cfg2 (config/teaser/jeep_posche.yaml): default model config, Replace + blend-masked self-attention
    "replace_blend": dict(
        model_config={"lora": 160}, prompts=(FULL_SRC, FULL_TGT), is_replace=True, cross_replace={"default_": 0.5},
        self_replace=1.0, blend_words=[["silver", "jeep"], ["Porsche", "car"]], eq_params=None),
    # the JUDGED launch shapes (bench.py's job): F = 8 -> one 8-frame inversion launch and one 16-frame CFG edit launch per layer,
    # i.e. the 320 x 128 ring tile / 320 x 256 ping-pong tile routing, the flash dispatch order with frames 0 and 1 single-source,
    # the one-launch GroupNorms "at 8 frames" -- 1 + 1 steps (cross replacement live at step 0: default_ 1.0), Replace + blend mask,
    # always with the all-fp32 leg (oracle edit on the oracle's own maps): the flip rate against the fp32 reference at this shape
    "cfg2_8f": dict(
        model_config={"lora": 160}, prompts=(FULL_SRC, FULL_TGT), is_replace=True, cross_replace={"default_": 1.0},
        self_replace=1.0, blend_words=[["silver", "jeep"], ["Porsche", "car"]], eq_params=None, F=8, T=1, pure_edit=True),
    # BASELINE cfg1 / cfg3 model config (config/style/sun_flower_van_gogh.yaml:69-73): K/V from the middle frame only, and
    # only where the width reaches 640 (the 320-wide level runs per-frame attention); Refine + Reweight (x10), no mask
    "refine_reweight_mid": dict(
        model_config={"lora": 160, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 640},
        prompts=("a sunflower in a vase on a table", "a sunflower in a vase on a table, van gogh style"), is_replace=False,
        cross_replace={"default_": 0.5}, self_replace=0.5, blend_words=None,
        eq_params={"words": ["van", "gogh"], "values": [10, 10]}),
}


def run_fullwidth_case(device, F=None, T=None, pure_edit=False, seed=11, variant="replace_blend"):
    """Native pipeline vs the fp32 CPU oracle (oracle.OracleUNet / ddim_inversion / ddim_edit) at REAL width with the same
    procedural weights: F frames, T inversion steps with capture + T CFG edit steps (unet_3d_condition.py:307-446 /
    attention_register.py:23-218 end to end).  F = 3 keeps the frame axis non-degenerate: with [-1, 'first'] the two K/V
    slots of frame 2 are frames 1 and 0 (attention.py:374-388), with ['mid'] every frame reads frame 1, and the 5-D
    GroupNorm spans three frames.  This is the only place where the d = 40 log2-folded flash path, the conv / GEMM tile
    routing, the batched time-embedding projection and the level-adapted GroupNorm chunks are compared with the oracle as
    an assembled UNet.  `variant` picks the model config + controller (FULL_VARIANTS)."""
    from oracle import fatezero_oracle as O
    V = FULL_VARIANTS[variant]
    F = V.get("F", 3) if F is None else F
    T = V.get("T", 2) if T is None else T
    pure_edit = pure_edit or V.get("pure_edit", False)
    th = [FULL_BLEND_TH, FULL_BLEND_TH]
    mc = dict(V["model_config"])
    src, tgt = V["prompts"]
    unet = UNetPseudo3DConditionModel(sample_size=64, **SD15, **mc)
    shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    sd = procedural_state_dict(shapes)
    unet.load_state_dict(sd)
    unet = unet.half().to(device).eval()
    ounet = O.OracleUNet(sd, O.UNetConfig(**SD15, model_config=mc))
    tok = ReplayTokenizer()
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=tok, unet=unet, scheduler=DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler.set_timesteps(T)
    g = torch.Generator().manual_seed(seed)
    z0 = torch.randn(1, 4, F, 64, 64, generator=g)
    emb_src = torch.randn(2, 77, 768, generator=g) * 0.5
    emb_tgt = emb_src + 0.25 * torch.randn(2, 77, 768, generator=g)
    res = {"variant": variant, "frames": F}
    # single UNet forward first (inversion mode, no controller side effects on the result)
    lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1,
                                             text_embeddings=emb_src.to(device), store_attention=True, LOW_RESOURCE=True,
                                             latents=z0.to(device))
    ostore = O.StoreController()
    olat = O.ddim_inversion(ounet, O.DDIMSchedule(T), z0, emb_src[1:], ostore)
    res["inv_scale"] = float(olat[-1].abs().max())
    res["inv_err_steps"] = [float((lat[i].float().cpu() - olat[i]).abs().max()) for i in range(1, T + 1)]
    res["inv_err"] = res["inv_err_steps"][-1]
    store = pipe.store_controller
    worst_cross = worst_self = 0.0
    for k, lst in ostore.attention_store_all_step[0].items():
        got = store.attention_store_all_step[0][k]
        assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in lst], k
        for a, b in zip(got, lst):
            e = float((a.float().cpu() - b).abs().max())
            if k.endswith("cross"):
                worst_cross = max(worst_cross, e)
            else:
                worst_self = max(worst_self, e)
    res["map_err"], res["self_map_err"] = worst_cross, worst_self
    kw = dict(prompt=tgt, source_prompt=src, num_inference_steps=T, cross_replace_steps=dict(V["cross_replace"]),
              self_replace_steps=V["self_replace"], use_inversion_attention=True, is_replace_controller=V["is_replace"],
              blend_th=list(th), save_self_attention=False, guidance_scale=7.5)
    if V["blend_words"] is not None:
        kw.update(blend_words=V["blend_words"], blend_self_attention=True)
    if V["eq_params"] is not None:
        kw.update(eq_params=V["eq_params"])
    pipe._encode_prompt = lambda *a, **k: emb_tgt.to(device)
    zT = lat[-1]
    edited = pipe(latents=zT, edit_type="swap", output_type="latent", **kw)["sdimage_output"].images.float().cpu()
    ctrl = pipe.last_edit_controller

    def oracle_edit(ost, z):
        c = O.make_edit_controller(tok, [src, tgt], ost, T, V["is_replace"], dict(V["cross_replace"]), V["self_replace"],
                                   blend_words=V["blend_words"], eq_params=V["eq_params"], blend_th=tuple(th),
                                   blend_self_attention=V["blend_words"] is not None, save_self_attention=False)
        return O.ddim_edit(ounet, O.DDIMSchedule(T), z, emb_tgt, c, guidance_scale=7.5), c
    # oracle edit on the natively captured maps, from the native inverted latent: isolates the edit pass; masks bit-exact
    ost = O.StoreController()
    ost.attention_store_all_step = [{k: [t.float().cpu() for t in v] for k, v in d.items()}
                                    for d in store.attention_store_all_step]
    ost.latents_store = [t.float().cpu() for t in store.latents_store]
    o_edit, o_ctrl = oracle_edit(ost, zT.float().cpu())
    res["edit_scale"] = float(o_edit.abs().max())
    res["edit_err_vs_oracle_on_native_maps"] = float((edited - o_edit).abs().max())
    res["edit_err_vs_oracle_on_native_maps_q99"] = float(torch.quantile((edited - o_edit).abs().flatten(), 0.99))
    if V["blend_words"] is not None:
        res["attn_mask_flips_same_maps"], res["attn_mask_total"] = _mask_flips(ctrl.attention_blend.mask_list,
                                                                                o_ctrl.attention_blend.mask_list)
        res["mask_ones"] = int(sum(int(m.bool().sum()) for m in ctrl.attention_blend.mask_list))
        res["mask_ones_frac"] = res["mask_ones"] / res["attn_mask_total"]
        per_mask = [float(m.float().mean()) for m in ctrl.attention_blend.mask_list]
        res["mask_ones_frac_min_max"] = (min(per_mask), max(per_mask))
    if pure_edit:  # the all-fp32 run: oracle edit on the ORACLE's maps from the oracle's inverted latent
        native2 = pipe(latents=olat[-1].to(device), edit_type="swap", output_type="latent", **kw)["sdimage_output"].images
        p_edit, p_ctrl = oracle_edit(ostore, olat[-1])
        res["edit_err"] = float((native2.float().cpu() - p_edit).abs().max())
        res["edit_err_q99"] = float(torch.quantile((native2.float().cpu() - p_edit).abs().flatten(), 0.99))
        if V["blend_words"] is not None:
            res["attn_mask_flips"], _ = _mask_flips(pipe.last_edit_controller.attention_blend.mask_list,
                                                    p_ctrl.attention_blend.mask_list)
            res["attn_mask_flip_rate"] = res["attn_mask_flips"] / res["attn_mask_total"]
    return res


def run_fullwidth_forward(device, F=16, variant="refine_reweight_mid", seed=13, t=481, oracle_device=None):
    """ONE forward of the full-width UNet on an F-frame clip (no controller) against oracle.OracleUNet: the clip lengths of BASELINE
    cfg3 / cfg4 / cfg5 (16 / 24 / 32 frames) differ from the judged 8 in more than size -- the temporal attention kernel is instantiated
    per clip length, GroupNorm statistics span F frames, the flash dispatch order and the sparse-causal source frames follow clip_len --
    and a whole inversion + edit at 16 frames is minutes of CPU oracle; one forward (~45 s at 16 frames) covers those code paths."""
    from oracle import fatezero_oracle as O
    mc = dict(FULL_VARIANTS[variant]["model_config"])
    unet = UNetPseudo3DConditionModel(sample_size=64, **SD15, **mc)
    shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    sd = procedural_state_dict(shapes)
    unet.load_state_dict(sd)
    unet = unet.half().to(device).eval()
    ounet = O.OracleUNet(sd, O.UNetConfig(**SD15, model_config=mc), device=oracle_device)
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(1, 4, F, 64, 64, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g) * 0.5
    y = unet(z.to(device).half(), t, ctx.to(device).half()).sample.float().cpu()
    fast_before = O.FAST_LARGE_ATTENTION
    O.FAST_LARGE_ATTENTION = oracle_device is not None  # (the GPU-executed oracle: pinned by test_oracle_executed_on_the_gpu_matches_the_cpu_oracle)
    try:
        ref = ounet(z, t, ctx).cpu()
    finally:
        O.FAST_LARGE_ATTENTION = fast_before
    return {"frames": F, "variant": variant, "err": float((y - ref).abs().max()), "scale": float(ref.abs().max()),
            "err_q99": float(torch.quantile((y - ref).abs().flatten()[:: max(1, y.numel() // 1000000)], 0.99))}


# full SD-1.x width, 2 + 2 steps, 3 frames (measured: inversion 0.10 - 0.11 %, cross maps 0.6e-2 - 0.7e-2, self maps 1e-3, edit on the
# natively captured maps 0.72 - 0.74 % max / 0.39 - 0.43 % q99)
FULL_LATENT_TOL = 4e-3
FULL_MAP_TOL = 1.2e-2


def check_fullwidth(res):
    assert res["inv_err"] <= FULL_LATENT_TOL * res["inv_scale"], res
    assert res["map_err"] <= FULL_MAP_TOL and res["self_map_err"] <= SELF_MAP_TOL, res
    assert res["edit_err_vs_oracle_on_native_maps"] <= EDIT_TOL_SAME_MAPS * res["edit_scale"], res
    assert res["edit_err_vs_oracle_on_native_maps_q99"] <= EDIT_Q99_TOL * res["edit_scale"], res
    if "attn_mask_flips_same_maps" in res:
        assert res["attn_mask_flips_same_maps"] == 0, res
        # a (nearly) all-0 / all-1 mask would test nothing: the live-vs-stored row select must see both kinds of rows in bulk
        assert FULL_MASK_BAND[0] <= res["mask_ones_frac"] <= FULL_MASK_BAND[1], res
    if "edit_err" in res:
        assert res["edit_err"] <= EDIT_TOL_VS_REFERENCE * res["edit_scale"], res
        if "attn_mask_flips" in res:
            assert res["attn_mask_flips"] <= MASK_FLIP_TOL * res["attn_mask_total"], res


def run_unet_golden(name, device):
    """Native UNet forward on a golden recorded from the UNMODIFIED reference UNet (oracle/gen_golden.py: gen_unet)."""
    meta = load_json("unet_meta.json")
    m = meta[name]
    g = load_npz(name + ".npz")
    shapes = m["state_dict_shapes"] or meta[m.get("shapes_from", "unet_tiny16_default")]["state_dict_shapes"]
    unet = UNetPseudo3DConditionModel(sample_size=g["x"].shape[-1], **TINY[m["kind"]], **m["model_config"])
    unet.load_state_dict(procedural_state_dict([(n, tuple(s)) for n, s in shapes]))
    unet = unet.half().to(device).eval()
    y = unet(torch.from_numpy(g["x"]).to(device), int(g["t"]), torch.from_numpy(g["ctx"]).to(device)).sample.float().cpu()
    ref = torch.from_numpy(g["y"])
    return {"err": float((y - ref).abs().max()), "scale": float(ref.abs().max())}


def run_drift_case(device, T=50, F=2, L=32, marks=(10, 25, 50), seed=5):
    """T = 50 steps each way at tiny16 width: latent error of the native pipeline vs the fp32 oracle at `marks`, for the
    inversion (vs oracle.ddim_inversion) and for a Replace edit started from the oracle's inverted latent on the oracle's
    own maps (vs oracle.ddim_edit) -- shows the fp16 error does not grow past the stated latent tolerance over a full
    50-step job."""
    from oracle import fatezero_oracle as O
    mc = {"lora": 16}
    unet = build_unet("tiny16", mc, device)
    shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    ounet = O.OracleUNet(procedural_state_dict(shapes), O.UNetConfig(**TINY["tiny16"], model_config=mc))
    tok = ReplayTokenizer()
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=tok, unet=unet, scheduler=DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler.set_timesteps(T)
    g = torch.Generator().manual_seed(seed)
    z0 = torch.randn(1, 4, F, L, L, generator=g)
    emb_src = torch.randn(2, 77, 64, generator=g) * 0.5
    emb_tgt = emb_src + 0.25 * torch.randn(2, 77, 64, generator=g)
    lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1,
                                             text_embeddings=emb_src.to(device), store_attention=True, LOW_RESOURCE=True,
                                             latents=z0.to(device))
    ostore = O.StoreController()
    olat = O.ddim_inversion(ounet, O.DDIMSchedule(T), z0, emb_src[1:], ostore)
    res = {"inv": {m: float((lat[m].float().cpu() - olat[m]).abs().max()) / float(olat[m].abs().max()) for m in marks}}
    octrl = O.make_edit_controller(tok, [FULL_SRC, FULL_TGT], ostore, T, True, {"default_": 0.5}, 0.5,
                                   save_self_attention=False)
    trace = {}

    class _Trace:
        def __init__(self, inner):
            self.inner, self.i = inner, 0

        def __call__(self, *a):
            return self.inner(*a)

        def step_callback(self, x):
            x = self.inner.step_callback(x)
            self.i += 1
            if self.i in marks:
                trace[self.i] = x.clone()
            return x
    O.ddim_edit(ounet, O.DDIMSchedule(T), olat[-1], emb_tgt, _Trace(octrl), guidance_scale=7.5)
    got = {}
    pipe._encode_prompt = lambda *a, **k: emb_tgt.to(device)

    def cb(i, t, x):
        if i + 1 in marks:
            got[i + 1] = x.float().cpu()
    # the native edit reads the NATIVE inversion maps (same weights, same input): upstream fp16 noise included
    pipe(latents=olat[-1].to(device), edit_type="swap", output_type="latent", prompt=FULL_TGT, source_prompt=FULL_SRC,
         num_inference_steps=T, cross_replace_steps={"default_": 0.5}, self_replace_steps=0.5, use_inversion_attention=True,
         is_replace_controller=True, save_self_attention=False, guidance_scale=7.5, callback=cb, callback_steps=1)
    res["edit"] = {m: float((got[m] - trace[m]).abs().max()) / float(trace[m].abs().max()) for m in marks}
    return res


def run_spill_case(device, disk_store, T=20, F=4, L=64, seed=9):
    """A T-step capture inversion + Replace edit with attention blend at tiny16 width, native pipeline only: what a run with the arena's spill
    tier on (disk_store=True) must reproduce bit for bit.  Returns (inverted latent, edited latent, per-step sums of the captured maps,
    a summary of the store's arena)."""
    unet = build_unet("tiny16", {"lora": 16}, device)
    tok = ReplayTokenizer()
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=tok, unet=unet, scheduler=DDIMScheduler(),
                                         disk_store=disk_store)
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler.set_timesteps(T)
    g = torch.Generator().manual_seed(seed)
    z0 = torch.randn(1, 4, F, L, L, generator=g)
    emb_src = torch.randn(2, 77, 64, generator=g) * 0.5
    emb_tgt = emb_src + 0.25 * torch.randn(2, 77, 64, generator=g)
    lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1, text_embeddings=emb_src.to(device),
                                             store_attention=True, LOW_RESOURCE=True, latents=z0.to(device))
    pipe._encode_prompt = lambda *a, **k: emb_tgt.to(device)
    out = pipe(latents=lat[-1], edit_type="swap", output_type="latent", prompt=FULL_TGT, source_prompt=FULL_SRC, num_inference_steps=T,
               cross_replace_steps={"default_": 0.6}, self_replace_steps=0.7, use_inversion_attention=True, is_replace_controller=True,
               blend_words=[["silver", "jeep"], ["Porsche", "car"]], blend_th=[0.55, 0.55], blend_self_attention=True, save_self_attention=True, guidance_scale=7.5,
               disk_store=disk_store)
    store = pipe.store_controller
    sums = [[float(m.cpu().double().sum()) for k in sorted(st) for m in st[k]] for st in store.attention_store_all_step]   # (on one device: same order)
    arena = store.arena   # (a snapshot: the store releases its arena when it is collected)
    info = {"spilled": sorted(arena.spilled), "spilled_bytes": arena.spilled_bytes, "fetch_stats": dict(arena.fetch_stats),
            "pinned": [sp.host.is_pinned() for sp in arena.spilled.values()], "hbm_bytes": arena.total_bytes}
    return lat[-1].float().cpu(), out["sdimage_output"].images.float().cpu(), sums, info


DRIFT_TOL = {"inv": 4e-3, "edit": 1.5e-2}   # measured at steps 10 / 25 / 50: inversion 0.03 / 0.06 / 0.15 %, edit 0.67 / 0.89 / 0.82 %


def check_drift(res):
    for part in ("inv", "edit"):
        for m, e in res[part].items():
            assert e <= DRIFT_TOL[part], (part, m, res)


# ------------------------------------------------------------------------------------------------------------
# BASELINE configs 3 / 4 / 5 (and cfg2 with a latent blend) as WHOLE jobs with every controller window opening AND closing
# ------------------------------------------------------------------------------------------------------------
# The clip lengths of config/style (16), config/attribute (24) and config/shape (32 frames at 576^2 = 72^2 latents) differ from the judged
# 8 frames in code paths, not just in size: the temporal attention kernel is instantiated per clip length, the 5-D GroupNorm statistics
# span F frames, the flash dispatch and the sparse-causal sources follow clip_len, at 72^2 the 36^2 level (1296 tokens) skips capture and
# the blend mask comes from three 18^2 maps.  Each case below is a T-step capture inversion + a T-step CFG edit with T chosen so that the
# cross-replace window [0, int(c (T + 1))), the self-replace window [0, int(s T)) and the latent-blend window (int(.2 T), int(.8 T)) all
# OPEN AND CLOSE inside the run (attention_util.py:129-158,195-197; ptp_utils.py:165-199; spatial_blend.py:117-121), at the TRUE
# geometry (frames, latent size, index lists, list-position blend slicing) and at the true head dims 40 / 80 / 160 (tiny40 width: 2 heads),
# or at full SD-1.x width for the 8-frame case.
#
# The oracle legs of these cases are 3 T forward-equivalents of a 16-32-frame UNet -- tens of minutes on the box's 16 host cores -- so
# the SAME oracle code (oracle/fatezero_oracle.py, fp32) is executed by torch on the GPU (`oracle_device`; torch's own fp32 library
# kernels, nothing of fatezero_amd), with FAST_LARGE_ATTENTION for the levels no controller touches.  tests/test_pipeline_gpu.py first
# pins that execution against the CPU execution of the oracle (test_oracle_executed_on_the_gpu_matches_the_cpu_oracle).
_MID = {"lora": 16, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 160}  # cfg3 / cfg4's model_config at tiny40 width
GEOMETRY_CASES = {
    # config/style/sun_flower_van_gogh.yaml:7,46,62-73 -- 16 frames, ['mid'] / least_sc_channel, Refine + Reweight (x10) and the
    # reference's DEFAULT attention blend: blend_th [2, 2] = the mask is never set, every self-attention row takes the stored map
    "cfg3_style_16f": dict(kind="tiny40", F=16, L=64, T=10, model_config=_MID, prompt_case="style_van_gogh", is_replace=False,
                           cross_replace={"default_": 0.5}, self_replace=0.5, eq_params={"words": ["van", "gogh"], "values": [10, 10]},
                           blend_words=[["sunflower"], ["sunflower"]], blend_th=[2, 2], blend_latents=False, regime="all_stored"),
    # config/attribute/squ_carrot_robot_eggplant.yaml:9-10,58,99 -- 24 frames -- in SURVEY 8(d)'s synthetic variant: Replace + blend
    # words + `blend_latents: True` (config/teaser/jeep_posche_local_latent_blend.yaml:38-39)
    "cfg4_attribute_24f_latentblend": dict(kind="tiny40", F=24, L=64, T=10, model_config=_MID, prompt_case="attribute_rabbit",
                                           is_replace=True, cross_replace={"default_": 0.5}, self_replace=0.5, eq_params=None,
                                           blend_words=[["squirrel"], ["rabbit"]], blend_th=[0.2, 0.2], blend_latents=True, regime="split"),
    # config/shape/swan_duck_flamingo.yaml:7 -- 32 frames at 576^2 (72^2 latents: 5184 / 1296 / 324 / 81 tokens), default index
    # [-1, 'first'], Replace + blend-masked self-attention at the OTHER extreme the reference documents (swan_duck_flamingo.yaml:59:
    # "blend_th : [0.0, 0.0], mask -> 1"): every row keeps the live attention -- the regime bench.py's job is in with its weights
    "cfg5_shape_32f_l72": dict(kind="tiny40", F=32, L=72, T=10, model_config={"lora": 16}, prompt_case="teaser_posche", is_replace=True,
                               cross_replace={"default_": 0.5}, self_replace=0.5, eq_params=None,
                               blend_words=[["silver", "jeep"], ["Porsche", "car"]], blend_th=[0.0, 0.0], blend_latents=False,
                               regime="all_live"),
    # config/teaser/jeep_posche_local_latent_blend.yaml at FULL SD-1.x width, the judged 8 frames, T = 4: cross window [0, 2), self window
    # [0, 2), latent blend live at steps 1-2 of 0-3; Replace + blend-masked self-attention + latent blend
    "cfg2_fullwidth_8f_latentblend": dict(kind="sd15", F=8, L=64, T=4, model_config={"lora": 160}, prompt_case="teaser_posche",
                                          is_replace=True, cross_replace={"default_": 0.5}, self_replace=0.5, eq_params=None,
                                          blend_words=[["silver", "jeep"], ["Porsche", "car"]], blend_th=None, blend_latents=True,
                                          regime="split"),
    # the reference's DEFAULT blend (th [2, 2]: every row takes the stored map -- the inject launches run without a row mask) at FULL width and
    # the judged 8 / 16-frame launch shapes; T = 2: the cross and self windows are live at step 0 and closed at step 1
    "cfg2_fullwidth_8f_all_stored": dict(kind="sd15", F=8, L=64, T=2, model_config={"lora": 160}, prompt_case="teaser_posche", is_replace=True,
                                         cross_replace={"default_": 0.5}, self_replace=0.5, eq_params=None,
                                         blend_words=[["silver", "jeep"], ["Porsche", "car"]], blend_th=[2, 2], blend_latents=False,
                                         regime="all_stored"),
    # THE JUDGED JOB at its own depth (opt-in, FZ_FULL_PARITY=1: minutes of GPU-executed fp32 oracle and ~225 GB of HBM for the two stores):
    # config/teaser/jeep_posche.yaml at FULL SD-1.x width, 8 frames, 64^2 latents, T = 50 + 50 (p2p_ddim_spatial_temporal.py:132-161, 386-421),
    # bench.py's controller: Replace + blend-masked self-attention, cross window [0, 25), self window [0, 25)
    "cfg2_fullwidth_8f_T50": dict(kind="sd15", F=8, L=64, T=50, model_config={"lora": 160}, prompt_case="teaser_posche", is_replace=True,
                                  cross_replace={"default_": 0.5}, self_replace=0.5, eq_params=None,
                                  blend_words=[["silver", "jeep"], ["Porsche", "car"]], blend_th=None, blend_latents=False, regime="split",
                                  deep=True),
    # cfg3's geometry ONCE at full width (opt-in as well): 16 frames, ['mid'] / least_sc_channel 640, Refine + Reweight x10, the default
    # blend_th [2, 2]; T = 4: cross window [0, 2), self window [0, 2)
    "cfg3_fullwidth_16f_mid": dict(kind="sd15", F=16, L=64, T=4,
                                   model_config={"lora": 160, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 640},
                                   prompt_case="style_van_gogh", is_replace=False, cross_replace={"default_": 0.5}, self_replace=0.5,
                                   eq_params={"words": ["van", "gogh"], "values": [10, 10]}, blend_words=[["sunflower"], ["sunflower"]],
                                   blend_th=[2, 2], blend_latents=False, regime="all_stored", deep=True),
    # miniature of the same harness for the GPU-less suite (emulator): all three windows toggle inside T = 6
    "mini_emu": dict(kind="tiny16", F=2, L=64, T=6, model_config={"lora": 16}, prompt_case="teaser_posche", is_replace=True,
                     cross_replace={"default_": 0.5}, self_replace=0.5, eq_params=None,
                     blend_words=[["silver", "jeep"], ["Porsche", "car"]], blend_th=[0.3, 0.3], blend_latents=True, regime=None),
}
GEOMETRY_SPLIT_TH = {"tiny40": 0.55, "sd15": FULL_BLEND_TH}  # thresholds that put 20-80 % of the mask on either side (asserted)


class _LazyF32(list):
    """A list of stored maps that hands out fp32 copies on the oracle's device at the moment of use (the natively captured fp16 maps of a
    50-step full-width job are 75 GB: an eager fp32 copy for the oracle would be 150 GB more)."""

    def __init__(self, items, device):
        super().__init__(items)
        self._dev = device

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [t.float().to(self._dev) for t in list.__getitem__(self, i)]
        return list.__getitem__(self, i).float().to(self._dev)

    def __iter__(self):
        return (list.__getitem__(self, i).float().to(self._dev) for i in range(len(self)))


def controller_windows(T, cross, self_replace, blend_latents):
    """The steps at which each fusion is live, from the reference's own arithmetic."""
    c = cross["default_"] if isinstance(cross, dict) else cross
    c = (0.0, c) if isinstance(c, float) else c
    s = (0.0, self_replace) if isinstance(self_replace, float) else self_replace
    w = {"cross": list(range(int(c[0] * (T + 1)), min(T, int(c[1] * (T + 1))))),          # ptp_utils.py:165-176 on T + 1 rows
         "self": list(range(int(T * s[0]), min(T, int(T * s[1]))))}                       # attention_util.py:195-197
    if blend_latents:  # spatial_blend.py:117-121: counter (1-based, incremented first) strictly between int(.2 T) and int(.8 T)
        w["latent_blend"] = [i for i in range(T) if int(0.2 * T) < i + 1 < int(0.8 * T)]
    return w


def run_geometry_case(name, device, oracle_device=None, seed=21, fp32_leg=True):
    """Native pipeline vs the fp32 oracle on a whole T + T job of GEOMETRY_CASES[name]; see the block comment above."""
    from oracle import fatezero_oracle as O
    G = GEOMETRY_CASES[name]
    odev = torch.device(device if oracle_device is None else oracle_device)
    arch = SD15 if G["kind"] == "sd15" else TINY[G["kind"]]
    F, L, T, mc = G["F"], G["L"], G["T"], dict(G["model_config"])
    consts = load_json("host_constants.json")[G["prompt_case"]]
    src, tgt = consts["prompts"]
    th = list(G["blend_th"]) if G["blend_th"] is not None else [GEOMETRY_SPLIT_TH[G["kind"]]] * 2
    unet = UNetPseudo3DConditionModel(sample_size=64, **arch, **mc)
    shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    sd = procedural_state_dict(shapes)
    unet.load_state_dict(sd)
    unet = unet.half().to(device).eval()
    fast_before = O.FAST_LARGE_ATTENTION
    O.FAST_LARGE_ATTENTION = True
    try:
        ounet = O.OracleUNet(sd, O.UNetConfig(**arch, model_config=mc), device=odev)
        tok = ReplayTokenizer()
        pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=tok, unet=unet, scheduler=DDIMScheduler())
        pipe.set_progress_bar_config(disable=True)
        pipe.scheduler.set_timesteps(T)
        g = torch.Generator().manual_seed(seed)
        cdim = arch["cross_attention_dim"]
        z0 = torch.randn(1, 4, F, L, L, generator=g)
        emb_src = torch.randn(2, 77, cdim, generator=g) * 0.5
        emb_tgt = emb_src + 0.25 * torch.randn(2, 77, cdim, generator=g)
        windows = controller_windows(T, G["cross_replace"], G["self_replace"], G["blend_latents"])
        res = {"case": name, "frames": F, "latent": L, "T": T, "windows": windows, "blend_th": th}
        for wname, steps in windows.items():  # every window opens AND closes inside the run
            assert 0 < len(steps) < T and steps[-1] < T - 1, (wname, steps)
        # ---- inversion with capture -------------------------------------------------------------------------------------
        lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1, text_embeddings=emb_src.to(device),
                                                 store_attention=True, LOW_RESOURCE=True, latents=z0.to(device))
        ostore = O.StoreController()
        olat = O.ddim_inversion(ounet, O.DDIMSchedule(T), z0, emb_src[1:], ostore)
        res["inv_scale"] = float(olat[-1].abs().max())
        res["inv_err_steps"] = [float((lat[i].float().cpu() - olat[i].cpu()).abs().max()) for i in range(1, T + 1)]
        res["inv_err"] = max(res["inv_err_steps"])
        store = pipe.store_controller
        assert len(store.attention_store_all_step) == len(ostore.attention_store_all_step) == T
        worst_cross = worst_self = 0.0
        res["map_list_lengths"] = {k: len(v) for k, v in ostore.attention_store_all_step[0].items()}
        for step in (0, T - 1):
            for k, lst in ostore.attention_store_all_step[step].items():
                got = store.attention_store_all_step[step][k]
                assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in lst], (k, step)
                for a, b in zip(got, lst):
                    e = float((a.float().cpu() - b.cpu()).abs().max())
                    if k.endswith("cross"):
                        worst_cross = max(worst_cross, e)
                    else:
                        worst_self = max(worst_self, e)
        res["map_err"], res["self_map_err"] = worst_cross, worst_self
        # ---- edit -------------------------------------------------------------------------------------------------------
        kw = dict(prompt=tgt, source_prompt=src, num_inference_steps=T, cross_replace_steps=dict(G["cross_replace"]),
                  self_replace_steps=G["self_replace"], use_inversion_attention=True, is_replace_controller=G["is_replace"],
                  blend_th=list(th), save_self_attention=False, guidance_scale=7.5, blend_words=G["blend_words"], blend_self_attention=True,
                  blend_latents=G["blend_latents"])
        if G["eq_params"] is not None:
            kw.update(eq_params=G["eq_params"])
        pipe._encode_prompt = lambda *a, **k: emb_tgt.to(device)

        def native_edit(z):
            per_step = {}

            def cb(i, t, x):
                per_step[i] = x.float().cpu()
            out = pipe(latents=z.to(device), edit_type="swap", output_type="latent", callback=cb, callback_steps=1, **kw)
            return out["sdimage_output"].images.float().cpu(), per_step, pipe.last_edit_controller

        def oracle_edit(ost, z, forced_applied=None):
            c = O.make_edit_controller(tok, [src, tgt], ost, T, G["is_replace"], dict(G["cross_replace"]), G["self_replace"],
                                       blend_words=G["blend_words"], eq_params=G["eq_params"], blend_th=tuple(th),
                                       blend_self_attention=True, blend_latents=G["blend_latents"], save_self_attention=False)
            if forced_applied is not None and c.latent_blend is not None:
                # teacher forcing: the oracle computes (and records) its OWN masks, but blends the edited latents with the mask the
                # native run applied at that step.  The target-prompt half of an applied mask is thresholded from the LIVE cross maps
                # (fp16 natively, fp32 here): a pixel within that noise of the threshold flips, moves its latent by |x - inverted| and
                # the following UNet steps spread the jump over the frame -- with the masks forced, what is compared is the arithmetic;
                # the flips themselves are counted and bounded separately.
                lb, queue = c.latent_blend, list(forced_applied)

                def forced_call(attention_store, target_h=None, target_w=None, x_t=None, _orig=lb.__call__):
                    n_before = len(lb.applied_mask_list)
                    x_own = _orig(attention_store, target_h, target_w, x_t=x_t)
                    if len(lb.applied_mask_list) == n_before:  # outside the window: nothing was blended
                        return x_own
                    m = queue.pop(0).to(x_t).reshape(1, 1, *x_t.shape[2:])
                    return torch.cat([x_t[:1], x_t[:1] + m * (x_t[1:] - x_t[:1])], dim=0)
                c.latent_blend = type("ForcedBlender", (), {"__call__": staticmethod(forced_call), "mask_list": lb.mask_list,
                                                            "applied_mask_list": lb.applied_mask_list})()
            per_step = {}
            inner_cb = c.step_callback

            def traced(x):
                x = inner_cb(x)
                per_step[len(per_step)] = x.float().cpu()
                return x
            c.step_callback = traced
            return O.ddim_edit(ounet, O.DDIMSchedule(T), z, emb_tgt, c, guidance_scale=7.5).cpu(), per_step, c

        def masks_report(nc, oc, tag):
            fl, tot = _mask_flips(nc.attention_blend.mask_list, oc.attention_blend.mask_list)
            res[f"attn_mask_flips_{tag}"], res["attn_mask_total"] = fl, tot
            if G["blend_latents"]:
                res[f"latent_mask_flips_{tag}"], res["latent_mask_total"] = _mask_flips(nc.latent_blend.mask_list, oc.latent_blend.mask_list)
                na, oa = nc.latent_blend.applied_mask_list, oc.latent_blend.applied_mask_list
                assert len(na) == len(oa) == len(windows["latent_blend"]), (len(na), len(oa), windows)
                res[f"applied_mask_flips_{tag}"], res["applied_mask_total"] = _mask_flips(na, oa)
                res[f"applied_mask_flips_{tag}_first_step"], res["applied_mask_step_total"] = _mask_flips(na[:1], oa[:1])
                fl = torch.stack([(a.bool().cpu() != b.bool().cpu()).reshape(-1, *a.shape[-2:]) for a, b in zip(na, oa)]).any(0)
                return torch.nn.functional.max_pool2d(fl[None].float(), 3, 1, 1)[0].bool()  # flipped pixels + their 3x3 neighbourhood
            return None

        # (1) same maps: oracle edit on the NATIVELY captured maps from the native inverted latent -- isolates the edit pass
        zT = lat[-1].float().cpu()
        edited, nsteps, nctrl = native_edit(zT)
        ost = O.StoreController()
        if G.get("deep"):
            ost.attention_store_all_step = [{k: _LazyF32(v, odev) for k, v in d.items()} for d in store.attention_store_all_step]
        else:
            ost.attention_store_all_step = [{k: [t.float().to(odev) for t in v] for k, v in d.items()} for d in store.attention_store_all_step]
        ost.latents_store = [t.float().to(odev) for t in store.latents_store]
        forced = list(nctrl.latent_blend.applied_mask_list) if G["blend_latents"] else None
        o_edit, osteps, octrl = oracle_edit(ost, zT, forced_applied=forced)
        del ost
        res["edit_scale"] = float(o_edit.abs().max())
        near = masks_report(nctrl, octrl, "same_maps")
        per_step = [float((nsteps[i] - osteps[i]).abs().max()) for i in range(T)]
        res["edit_err_steps_same_maps"] = per_step
        em = (edited - o_edit).abs().amax(dim=(0, 1))
        res["edit_err_same_maps"] = float(em.max())
        res["edit_err_same_maps_q99"] = float(torch.quantile((edited - o_edit).abs().flatten()[:: max(1, edited.numel() // 1000000)], 0.99))
        ml = nctrl.attention_blend.mask_list
        res["attn_mask_calls"] = len(ml)
        res["mask_ones_frac"] = float(sum(float(m.float().sum()) for m in ml) / max(1, sum(m.numel() for m in ml)))
        if G["blend_latents"]:
            al = nctrl.latent_blend.applied_mask_list
            res["applied_mask_ones_frac"] = float(sum(float(m.float().sum()) for m in al) / max(1, sum(m.numel() for m in al)))
        res["outputs_finite"] = bool(torch.isfinite(edited).all())
        if not fp32_leg:
            return res
        # (2) the all-fp32 leg: oracle edit on the ORACLE's maps from the oracle's inverted latent vs the native edit from that latent
        zo = olat[-1].cpu()
        edited2, nsteps2, nctrl2 = native_edit(zo)
        p_edit, psteps, pctrl = oracle_edit(ostore, zo)
        res["edit_err_steps_vs_fp32"] = [float((nsteps2[i] - psteps[i]).abs().max()) for i in range(T)]
        near2 = masks_report(nctrl2, pctrl, "vs_fp32")
        em2 = (edited2 - p_edit).abs().amax(dim=(0, 1))
        res["edit_err_vs_fp32"] = float(em2.max())
        d2 = (edited2 - p_edit).abs()
        if near2 is not None:  # a flipped pixel of the applied latent mask moves that latent by |x - inverted|: the bulk is judged away from them
            d2 = d2[:, :, ~near2]
        res["edit_err_vs_fp32_q99"] = float(torch.quantile(d2.flatten()[:: max(1, d2.numel() // 1000000)], 0.99))
        if near2 is not None:
            res["edit_err_vs_fp32_off_applied_flips"] = float(em2[~near2].max())
            res["edit_positions_beyond_band"] = int((em2 > EDIT_TOL_VS_REFERENCE * res["edit_scale"]).sum())
            res["edit_positions"] = em2.numel()
        res["outputs_finite"] = bool(torch.isfinite(edited).all() and torch.isfinite(edited2).all())
        return res
    finally:
        O.FAST_LARGE_ATTENTION = fast_before


# Bounds of the geometry cases: <= 2x the worst measured on MI355X (profiles/r05_parity_numbers.txt; the native path is bit-deterministic).
# Measured worst over the four cases -> bound:
GEO_LATENT_TOL = 4e-3             # inversion, worst of the T steps, / max |latent|                         0.21 %
GEO_MAP_TOL = 1.8e-2              # captured cross maps (first and last step), absolute                     1.00e-2
GEO_SELF_MAP_TOL = 2.8e-3         # captured self maps                                                      1.42e-3
GEO_EDIT_TOL_SAME_MAPS = 2.6e-2   # edit vs the oracle on the native maps (+ native applied masks), max      1.39 %  (grows ~0.14 % per step)
GEO_EDIT_Q99_TOL = 1.1e-2         # the same, 99th percentile                                               0.58 %
GEO_EDIT_TOL_VS_FP32 = 2.4e-2     # edit vs the all-fp32 run, max, when no mask pixel can flip               1.17 %
GEO_EDIT_Q99_VS_FP32 = 2e-2       # edit vs the all-fp32 run, 99th percentile (away from applied-mask flips)  1.04 %
GEO_ATTN_FLIP_TOL = 4.4e-3        # attention-blend mask elements that differ from the all-fp32 run          0.22 %
GEO_SRC_MASK_FLIP_TOL = 3e-3      # latent blender's source-prompt masks vs the all-fp32 run                 0.15 %
GEO_APPLIED_FLIP_TOL = 6.6e-3     # applied latent masks (source OR live target mask) vs the all-fp32 run, FIRST blend step   0.33 %
GEO_APPLIED_FLIP_ALL_TOL = 4e-2   # ... over all blend steps: a flip changes the latents, the next step's LIVE cross maps and with them the
                                  # next mask -- the count compounds over the window and moves with the fp32 library kernels the GPU-executed
                                  # oracle happens to get (0.63 % and 1.33 % on two boxes for the same 24-frame case): a sanity bound only
GEO_APPLIED_FLIP_SAME_TOL = 7e-3  # the same on identical stored maps (only the live target half differs)    0.34 %
GEO_BEYOND_BAND_TOL = 4e-2        # latent positions whose error leaves the 6 % band (flips + what they spread): 0.92 % / 1.49 % on two boxes
                                  # (compounds like the count above)


# The judged job at its own depth (cfg2_fullwidth_8f_T50) and cfg3's geometry at full width: bounds <= 2x what MI355X measured
# (profiles/r06_parity_numbers.txt); the rest of check_geometry applies unchanged.
GEO_DEEP = {
    # measured (r06f): inversion 0.161 %, cross maps 9.0e-3, self maps 9.3e-4, edit on the native maps 0.646 % max / 0.306 % q99 after 50 steps,
    # all-fp32 leg 0.693 % max / 0.310 % q99, attention-mask flips 0.151 %
    "cfg2_fullwidth_8f_T50": dict(inv=3.2e-3, cross_map=1.8e-2, self_map=1.9e-3, edit_same_maps=1.3e-2, edit_same_maps_q99=6.1e-3,
                                  attn_flips=3e-3, edit_vs_fp32=1.4e-2, edit_vs_fp32_q99=6.2e-3),
    # measured (r06f): inversion 0.110 %, cross maps 8.2e-3, self maps 1.41e-3, edit 0.824 % max / 0.427 % q99, all-fp32 0.949 % / 0.432 %, no flips
    "cfg3_fullwidth_16f_mid": dict(inv=2.2e-3, cross_map=1.7e-2, self_map=2.8e-3, edit_same_maps=1.65e-2, edit_same_maps_q99=8.6e-3,
                                   attn_flips=1e-3, edit_vs_fp32=1.9e-2, edit_vs_fp32_q99=8.7e-3),
}


def check_geometry(res):
    G = GEOMETRY_CASES[res["case"]]
    if G.get("deep") and res["case"] in GEO_DEEP:
        return _check_deep(G, res, GEO_DEEP[res["case"]])
    assert res["outputs_finite"], res
    assert res["inv_err"] <= GEO_LATENT_TOL * res["inv_scale"], res
    assert res["map_err"] <= GEO_MAP_TOL and res["self_map_err"] <= GEO_SELF_MAP_TOL, res
    # identical maps on both sides: the attention-blend masks are bit-exact
    assert res["attn_mask_flips_same_maps"] == 0, res
    assert res["edit_err_same_maps_q99"] <= GEO_EDIT_Q99_TOL * res["edit_scale"], res
    if G["blend_latents"]:
        # the source-prompt half of a latent mask comes from the stored maps (identical): exact; the target-prompt half of the APPLIED
        # mask is thresholded from the LIVE cross maps (fp16 here, fp32 in the oracle)
        assert res["latent_mask_flips_same_maps"] == 0, res
        assert res["applied_mask_flips_same_maps"] <= GEO_APPLIED_FLIP_SAME_TOL * res["applied_mask_total"], res
    # (latent blend: the oracle blends with the natively applied masks in this leg -- the arithmetic is compared, the flips are counted above)
    assert res["edit_err_same_maps"] <= GEO_EDIT_TOL_SAME_MAPS * res["edit_scale"], res
    _check_regime(G, res)
    if "edit_err_vs_fp32" not in res:
        return
    if G["blend_latents"]:
        assert res["applied_mask_flips_vs_fp32_first_step"] <= GEO_APPLIED_FLIP_TOL * res["applied_mask_step_total"], res
        assert res["applied_mask_flips_vs_fp32"] <= GEO_APPLIED_FLIP_ALL_TOL * res["applied_mask_total"], res
        assert res["latent_mask_flips_vs_fp32"] <= GEO_SRC_MASK_FLIP_TOL * res["latent_mask_total"], res
        # a flipped pixel moves its latent by |x - inverted| and the UNet steps that follow spread the jump over the frame (global
        # attention): over several blend steps the max is not boundable, away from the flips either -- the COUNT of positions that leave
        # the band is, and so is the bulk (q99 below, taken away from the flips)
        assert res["edit_positions_beyond_band"] <= GEO_BEYOND_BAND_TOL * res["edit_positions"], res
    elif res["attn_mask_flips_vs_fp32"] == 0:
        assert res["edit_err_vs_fp32"] <= GEO_EDIT_TOL_VS_FP32 * res["edit_scale"], res
    else:
        assert res["edit_err_vs_fp32"] <= EDIT_TOL_VS_REFERENCE * res["edit_scale"], res
    assert res["attn_mask_flips_vs_fp32"] <= GEO_ATTN_FLIP_TOL * res["attn_mask_total"], res
    assert res["edit_err_vs_fp32_q99"] <= GEO_EDIT_Q99_VS_FP32 * res["edit_scale"], res


def _check_deep(G, res, B):
    assert res["outputs_finite"], res
    assert res["inv_err"] <= B["inv"] * res["inv_scale"], res
    assert res["map_err"] <= B["cross_map"] and res["self_map_err"] <= B["self_map"], res
    assert res["attn_mask_flips_same_maps"] == 0, res                      # identical maps on both sides: bit-exact masks
    assert res["edit_err_same_maps"] <= B["edit_same_maps"] * res["edit_scale"], res
    assert res["edit_err_same_maps_q99"] <= B["edit_same_maps_q99"] * res["edit_scale"], res
    _check_regime(G, res)
    if "edit_err_vs_fp32" in res:
        assert res["attn_mask_flips_vs_fp32"] <= B["attn_flips"] * res["attn_mask_total"], res
        assert res["edit_err_vs_fp32_q99"] <= B["edit_vs_fp32_q99"] * res["edit_scale"], res
        assert res["edit_err_vs_fp32"] <= B["edit_vs_fp32"] * res["edit_scale"], res


def _check_regime(G, res):
    if G["regime"] == "all_stored":
        assert res["mask_ones_frac"] == 0.0, res          # blend_th [2, 2]: no row keeps the live attention
    elif G["regime"] == "all_live":
        assert res["mask_ones_frac"] >= 0.9, res
    elif G["regime"] == "split":
        assert FULL_MASK_BAND[0] <= res["mask_ones_frac"] <= FULL_MASK_BAND[1], res
