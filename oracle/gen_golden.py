#!/usr/bin/env python3
"""Generate tests/golden/* by running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE.  Runs only in the authoring container (the reference does not exist on the
GPU box); its outputs are committed as small fixtures and pin `oracle/fatezero_oracle.py`
(tests/test_oracle_golden.py) and, through it or directly, the HIP path (tests -m gpu).

    python oracle/gen_golden.py            # all scenarios
    python oracle/gen_golden.py pipeline:pipe_f4_prev_first,pipe_f3_mid_next   # (re)record only these pipeline runs
    python oracle/gen_golden.py const unet # a subset

How the reference is made importable: `oracle/refshim` restates the
diffusers==0.11.1 classes it imports and stubs the visualisation-only modules; the tokenizer is the
CLIP BPE vendored at /root/reference/CLIP/clip (the same vocabulary as SD's CLIP-L tokenizer).
"""
import gzip
import hashlib
import html
import importlib.util
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

sys.path.insert(0, os.path.join(HERE, "refshim"))
from stubs import install  # noqa: E402

install()
sys.path.insert(0, REF)
# NOTE: the repo root must NOT be on sys.path here: the reference's `video_diffusion` is a namespace package (no
# __init__.py) and this repo's alias package of the same name would shadow it.
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]
import torch  # noqa: E402

torch.cuda.get_device_name = lambda *a, **k: "cpu"  # attention.py:229 calls it unconditionally

_wspec = importlib.util.spec_from_file_location("_oracle_weights", os.path.join(HERE, "weights.py"))
_wmod = importlib.util.module_from_spec(_wspec)
_wspec.loader.exec_module(_wmod)
procedural_state_dict = _wmod.procedural_state_dict
from video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel  # noqa: E402
import video_diffusion as _vd  # noqa: E402
assert list(_vd.__path__)[0].startswith(REF), "golden vectors must come from the unmodified reference"
from video_diffusion.pipelines.p2p_ddim_spatial_temporal import P2pDDIMSpatioTemporalPipeline  # noqa: E402
from video_diffusion.prompt_attention import attention_util, ptp_utils, seq_aligner  # noqa: E402
from video_diffusion.prompt_attention.spatial_blend import SpatialBlender  # noqa: E402
from diffusers.schedulers import DDIMScheduler  # noqa: E402  (refshim restatement)


# ---------------------------------------------------------------------------------------------
# tokenizer adapter over the vendored CLIP BPE (HF CLIPTokenizer semantics for encode / decode)
# ---------------------------------------------------------------------------------------------
def load_bpe_tokenizer():
    spec = importlib.util.spec_from_file_location("_ref_simple_tokenizer", f"{REF}/CLIP/clip/simple_tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    st = mod.SimpleTokenizer(f"{REF}/CLIP/clip/bpe_simple_vocab_16e6.txt.gz")

    class Tok:
        bos, eos = st.encoder["<|startoftext|>"], st.encoder["<|endoftext|>"]
        model_max_length = 77

        def encode(self, text):
            return [self.bos] + st.encode(text) + [self.eos]

        def decode(self, ids):
            ids = [int(i) for i in ids]
            return st.decode(ids).strip()  # HF convert_tokens_to_string: replace('</w>',' ').strip()

    return Tok()


class RecordingTokenizer:
    """Wraps a tokenizer and records every encode/decode so tests can replay it without the BPE vocab."""

    def __init__(self, tok):
        self.tok = tok
        self.enc, self.dec = {}, {}

    def encode(self, text):
        ids = self.tok.encode(text)
        self.enc[text] = [int(i) for i in ids]
        return ids

    def decode(self, ids):
        s = self.tok.decode(ids)
        self.dec[",".join(str(int(i)) for i in ids)] = s
        return s


# ---------------------------------------------------------------------------------------------
def save_npz(name, **arrays):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def tensor_digest(t: torch.Tensor):
    """Order-sensitive fp64 checksums + a strided sample: enough to pin a big tensor in a few bytes."""
    t = t.detach().double().flatten()
    n = t.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37)
    idx = torch.linspace(0, n - 1, steps=min(n, 16)).long()
    return {"n": n, "sum": float(t.sum()), "wsum": float((t * w).sum()), "abs": float(t.abs().sum()),
            "sample": [float(x) for x in t[idx]]}


# ---------------------------------------------------------------------------------------------
# scenario: host-side constants
# ---------------------------------------------------------------------------------------------
PROMPT_CASES = [
    # (name, source, target, is_replace, cross_replace_steps, blend_words, eq_params, T)
    ("teaser_posche", "a silver jeep driving down a curvy road in the countryside,",
     "a Porsche car driving down a curvy road in the countryside,", True, {"default_": 0.5},
     [["silver", "jeep"], ["Porsche", "car"]], None, 50),
    ("teaser_watercolor", "a silver jeep driving down a curvy road in the countryside,",
     "watercolor painting of a silver jeep driving down a curvy road in the countryside,", False,
     {"default_": 0.8}, [["jeep"], ["jeep"]], {"words": ["watercolor", "painting"], "values": [10, 10]}, 50),
    ("low_resource_watercolor", "a silver jeep driving down a curvy road in the countryside",
     "watercolor painting of a silver jeep driving down a curvy road in the countryside", False,
     {"default_": 0.8}, None, {"words": ["watercolor"], "values": [10]}, 10),
    ("style_van_gogh", "a sunflower in a vase on a table", "a sunflower in a vase on a table, van gogh style",
     False, {"default_": 0.5, "gogh": (0.0, 0.9)}, None, {"words": ["van", "gogh"], "values": [10, 10]}, 50),
    ("attribute_rabbit", "A squirrel is eating a carrot", "A rabbit is eating a carrot", True,
     {"default_": 0.5, "rabbit": 0.4}, [["squirrel"], ["rabbit"]], None, 50),
    ("shape_swan", "a black swan with a red beak swimming in a river near a wall and bushes,",
     "a Swarovski crystal swan with a red beak swimming in a river near a wall and bushes,", False,
     {"default_": 0.8}, [["black", "swan"], ["Swarovski", "crystal", "swan"]],
     {"words": ["Swarovski", "crystal"], "values": [5, 5]}, 50),
]


def gen_const(tok):
    out = {}
    for name, src, tgt, is_rep, crs, bw, eq, T in PROMPT_CASES:
        prompts = [src, tgt]
        case = {"prompts": prompts, "T": T, "is_replace": is_rep, "cross_replace_steps": crs,
                "blend_words": bw, "eq_params": eq}
        case["word_inds"] = {}
        for p in prompts:
            for w in sorted(set(p.split(" "))):
                case["word_inds"][f"{p}|{w}"] = ptp_utils.get_word_inds(p, w, tok).tolist()
        alpha = ptp_utils.get_time_words_attention_alpha(prompts, T, dict(crs), tok)
        case["cross_replace_alpha"] = alpha.reshape(T + 1, 77).to(torch.uint8).tolist()
        if is_rep:
            case["replacement_mapper"] = seq_aligner.get_replacement_mapper(prompts, tok)[0].tolist()
        else:
            mp, al = seq_aligner.get_refinement_mapper(prompts, tok)
            case["refinement_mapper"] = mp[0].tolist()
            case["refinement_alphas"] = al[0].tolist()
        if eq is not None:
            case["equalizer"] = attention_util.get_equalizer(tgt, eq["words"], eq["values"], tokenizer=tok)[0].tolist()
        if bw is not None:
            sb = SpatialBlender(prompts, bw, tokenizer=tok, NUM_DDIM_STEPS=T, save_path=None)
            case["alpha_layers"] = sb.alpha_layers.reshape(2, 77).tolist()
        out[name] = case
    with open(os.path.join(GOLD, "host_constants.json"), "w") as f:
        json.dump(out, f)
    print("  wrote host_constants.json")


# ---------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------
TINY = {
    # heads=2 => head dims 16/32/64/64
    "tiny16": dict(sample_size=64, block_out_channels=(32, 64, 128, 128), norm_num_groups=8,
                   cross_attention_dim=64, attention_head_dim=2),
    # heads=2 => head dims 40/80/160/160: the true SD-1.x head dims (d=40 needs MFMA K padding)
    "tiny40": dict(sample_size=64, block_out_channels=(80, 160, 320, 320), norm_num_groups=16,
                   cross_attention_dim=64, attention_head_dim=2),
}


def build_ref_unet(kind, model_config, seed=0):
    torch.manual_seed(0)
    unet = UNetPseudo3DConditionModel(**TINY[kind], **model_config)
    shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    unet.load_state_dict(procedural_state_dict(shapes, seed))
    unet.eval().requires_grad_(False)
    return unet, shapes


def gen_unet(only=None):
    meta = {}
    meta_path = os.path.join(GOLD, "unet_meta.json")
    if only and os.path.exists(meta_path):  # partial run: keep the other cases' records
        meta = json.load(open(meta_path))
    cases = [
        ("unet_tiny16_default", "tiny16", {"lora": 16}, 2, 16),
        ("unet_tiny16_mid", "tiny16", {"lora": 16, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 64}, 3, 16),
        ("unet_tiny16_conv1d", "tiny16", {}, 2, 8),  # no 'lora' key: plain temporal Conv1d with bias
        ("unet_tiny40_default", "tiny40", {"lora": 160}, 2, 16),
        # 576^2 frames (BASELINE cfg5): 72^2 latents -> 5184 / 1296 / 324 / 81 tokens, none of them a multiple of 64 below the top level
        ("unet_tiny40_l72", "tiny40", {"lora": 160}, 2, 72),
    ]
    for name, kind, mc, F_, L in cases:
        if only and name not in only:
            continue
        unet, shapes = build_ref_unet(kind, mc)
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(2, 4, F_, L, L, generator=g)
        ctx = torch.randn(2, 77, 64, generator=g)
        store = attention_util.AttentionStore()
        store.LOW_RESOURCE = True

        class _P:  # register_attention_control walks `model.unet`
            pass
        p = _P()
        p.unet = unet
        attention_util.register_attention_control(p, store)
        with torch.no_grad():
            t0 = time.time()
            y = unet(x, torch.tensor(481), encoder_hidden_states=ctx).sample
            dt = time.time() - t0
        maps = {k: [tensor_digest(m) for m in v] for k, v in store.step_store.items()}
        shapes_maps = {k: [list(m.shape) for m in v] for k, v in store.step_store.items()}
        save_npz(name + ".npz", x=x, ctx=ctx, y=y, t=np.int64(481))
        meta[name] = {"kind": kind, "model_config": mc, "F": F_, "L": L, "seconds": dt,
                      "state_dict_shapes": shapes if name.endswith("default") or "conv1d" in name else None,
                      "shapes_from": "unet_tiny40_default" if kind == "tiny40" else "unet_tiny16_default",
                      "map_shapes": shapes_maps, "map_digests": maps}
        print(f"  {name}: {dt:.1f}s  |y|={float(y.abs().mean()):.4f}")
    with open(meta_path, "w") as f:
        json.dump(meta, f)


# ---------------------------------------------------------------------------------------------
# scenario: controllers on synthetic maps (no UNet) -- fast
# ---------------------------------------------------------------------------------------------
class _SB(SpatialBlender):
    """The reference blender with only the PNG dump disabled (tvu.save_image is stubbed out)."""

    def __init__(self, *a, **k):
        k["save_path"] = None
        super().__init__(*a, **k)
        self._last_full_mask = None
        self.applied_mask_list = []  # what the reference never keeps: the rows that blend the EDITED latents (mask[1:]) when they do

    def get_mask(self, *a, **k):
        m = super().get_mask(*a, **k)
        self._last_full_mask = m.float().cpu().detach().clone()
        return m

    def __call__(self, *a, **k):
        out = super().__call__(*a, **k)
        x_t = k.get("x_t")
        if x_t is not None and self.start_blend < self.counter < self.end_blend and self.substruct_layers is None:  # spatial_blend.py:117-120
            self.applied_mask_list.append(self._last_full_mask[1:])
        return out


def synthetic_layer_calls(F_, heads, n_kv, g, batch, peaky=4.0):
    """Yield (attn, is_cross, place) in the 512^2 call order at reduced F/heads."""
    order = [("down", 4096, 4), ("down", 1024, 4), ("down", 256, 4), ("mid", 64, 2),
             ("up", 256, 6), ("up", 1024, 6), ("up", 4096, 6)]
    for place, lq, n in order:
        for i in range(n):
            is_cross = (i % 2 == 1)
            if lq > 1024:
                # not captured / not edited: pass a tiny stand-in with Lq > 32**2 is too big; use a view trick
                attn = torch.zeros(batch * F_, heads, lq, 1).expand(batch * F_, heads, lq, 2)
                yield attn, is_cross, place
                continue
            lk = 77 if is_cross else n_kv * lq
            logits = torch.randn(batch * F_, heads, lq, lk, generator=g) * peaky
            if is_cross:  # spatial blobs per token so masks are structured
                r = int(lq ** 0.5)
                yy, xx = torch.meshgrid(torch.arange(r), torch.arange(r), indexing="ij")
                cx = torch.rand(batch * F_, 1, 1, lk, generator=g) * r
                cy = torch.rand(batch * F_, 1, 1, lk, generator=g) * r
                d2 = (xx.reshape(1, 1, lq, 1) - cx) ** 2 + (yy.reshape(1, 1, lq, 1) - cy) ** 2
                logits = logits * 0.3 - d2 / (2 * (r / 4) ** 2)
            yield logits.softmax(-1), is_cross, place


def gen_controller(tok):
    res = {}
    F_, heads, T = 2, 2, 3
    for name, src, tgt, is_rep, crs, bw, eq, _ in PROMPT_CASES[:2]:
        for variant in ("attn_blend", "latent_blend"):
            g = torch.Generator().manual_seed(7)
            store = attention_util.AttentionStore()
            store.LOW_RESOURCE = True
            inv_latents = []
            for s in range(T):  # inversion: capture
                for attn, is_cross, place in synthetic_layer_calls(F_, heads, 2, g, 1):
                    store(attn.clone() if attn.shape[-1] != 2 else attn, is_cross, place)
                lat = torch.randn(1, 4, F_, 64, 64, generator=g)
                store.step_callback(lat)
                inv_latents.append(lat)
            store.LOW_RESOURCE = False
            attention_util.SpatialBlender = _SB
            ctrl = attention_util.make_controller(
                tok, [src, tgt], is_rep, dict(crs), self_replace_steps=0.7, blend_words=bw,
                equilizer_params=eq, additional_attention_store=store, use_inversion_attention=True,
                blend_th=(0.3, 0.3), NUM_DDIM_STEPS=T, blend_latents=(variant == "latent_blend"),
                blend_self_attention=(variant == "attn_blend"), save_path="/tmp/unused",
                save_self_attention=False)
            attention_util.SpatialBlender = SpatialBlender
            digests, lat_out = [], []
            for s in range(T):
                step_d = []
                for attn, is_cross, place in synthetic_layer_calls(F_, heads, 2, g, 2):
                    a = attn.clone() if attn.shape[-1] != 2 else attn.clone()
                    out = ctrl(a, is_cross, place)
                    if attn.shape[-2] <= 1024:
                        step_d.append(tensor_digest(out[F_:]))
                lat = torch.randn(1, 4, F_, 64, 64, generator=g)
                lat2 = ctrl.step_callback(lat)
                lat_out.append(lat2)
                digests.append(step_d)
            key = f"{name}_{variant}"
            blender = ctrl.attention_blend if variant == "attn_blend" else ctrl.latent_blend
            masks = torch.stack(blender.mask_list[:12]).to(torch.uint8) if variant == "latent_blend" else None
            mask_small = [m.to(torch.uint8) for m in blender.mask_list]
            # attention-blend masks come in three resolutions; pack per resolution
            packed = {}
            for i, m in enumerate(mask_small):
                packed.setdefault(m.shape[-1], []).append(m)
            arrays = {f"mask_r{r}": np.packbits(torch.stack(v).numpy().astype(bool), axis=None)
                      for r, v in packed.items()}
            arrays.update({f"mask_r{r}_shape": np.array(torch.stack(v).shape) for r, v in packed.items()})
            arrays["latents_out_digest"] = np.array([[d["sum"], d["wsum"], d["abs"]] for d in map(tensor_digest, lat_out)])
            save_npz(f"controller_{key}.npz", **arrays)
            res[key] = {"digests": digests, "mask_mean": float(torch.cat([m.float().flatten() for m in mask_small]).mean())}
            print(f"  controller {key}: mask ones fraction {res[key]['mask_mean']:.3f}")
    with open(os.path.join(GOLD, "controller_meta.json"), "w") as f:
        json.dump({"F": F_, "heads": heads, "T": T, "seed": 7, "cases": res}, f)


# ---------------------------------------------------------------------------------------------
# scenario: the reference pipeline end to end in latent space
# ---------------------------------------------------------------------------------------------
class _FakeVAE(torch.nn.Module):
    class config:
        block_out_channels = (1, 1, 1, 1)

    def __init__(self, z_raw):
        super().__init__()
        self.z_raw = z_raw

    def encode(self, image):
        z = self.z_raw

        class D:
            def sample(self, generator=None):
                return z

        class O:
            latent_dist = D()
        return O()


def gen_pipeline(tok, only=None):
    """only: set of scenario names to (re)record; the others keep their committed vectors and pipeline_meta.json entries."""
    meta = {}
    meta_path = os.path.join(GOLD, "pipeline_meta.json")
    if only and os.path.exists(meta_path):
        meta = json.load(open(meta_path))
    T = 4
    scen = [
        # small-latent scenarios (every level <= 32x32 tokens is captured; no blend words): fast enough for the CPU suite
        ("pipe_small_replace", 0, {"lora": 16}, dict(self_replace_steps=0.5, L=32, no_blend=True)),
        ("pipe_small_refine_reweight", 2, {"lora": 16, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 64},
         dict(self_replace_steps=0.8, L=16, no_blend=True)),
        # name, prompt case idx, model_config, overrides
        ("pipe_replace_blend", 0, {"lora": 16}, dict(self_replace_steps=0.5, blend_self_attention=True)),
        ("pipe_refine_reweight_latentblend", 1, {"lora": 16, "SparseCausalAttention_index": ["mid"]},
         dict(self_replace_steps=0.75, blend_self_attention=True, blend_latents=True)),
        ("pipe_refine_noblend", 3, {"lora": 16, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 64},
         dict(self_replace_steps=0.5)),
        # more than two frames, so that the 'previous frame', 'first', 'mid' and 'next frame' key/value sources are all
        # different frames (with F = 2 the default [-1, 'first'] degenerates to frame 0 twice)
        ("pipe_f4_prev_first", 0, {"lora": 16}, dict(self_replace_steps=0.5, L=16, no_blend=True, F=4)),
        ("pipe_f3_mid_next", 2, {"lora": 16, "SparseCausalAttention_index": ["mid", 1]},
         dict(self_replace_steps=0.8, L=16, no_blend=True, F=3)),
        # 576^2 frames (BASELINE cfg5): 72^2 latents.  Only the 18^2 and 9^2 maps are <= 1024 tokens, so `down_cross[2:4]` is EMPTY and
        # the blend mask comes from three 18^2 maps (spatial_blend.py:78) instead of five 16^2 ones; 36^2 self maps are not replaced
        ("pipe_l72_replace_blend", 0, {"lora": 16}, dict(self_replace_steps=0.5, blend_self_attention=True, L=72)),
    ]
    for name, ci, mc, ov in scen:
        if only and name not in only:
            continue
        _, src, tgt, is_rep, crs, bw, eq, _ = PROMPT_CASES[ci]
        ov = dict(ov)
        L = ov.pop("L", 64)
        F_ = ov.pop("F", 2)
        if ov.pop("no_blend", False):
            bw = None
        unet, _ = build_ref_unet("tiny16", mc)
        g = torch.Generator().manual_seed(99)
        z_raw = torch.randn(F_, 4, L, L, generator=g)
        emb_src = torch.randn(2, 77, 64, generator=g)   # [uncond; cond(source)]
        emb_tgt = torch.cat([emb_src[:1], torch.randn(1, 77, 64, generator=g)])  # [uncond; cond(target)]
        sched = DDIMScheduler()
        pipe = P2pDDIMSpatioTemporalPipeline(vae=_FakeVAE(z_raw), text_encoder=torch.nn.Identity(), tokenizer=tok,
                                             unet=unet, scheduler=sched, disk_store=False)
        pipe.scheduler.set_timesteps(T)
        pipe.set_progress_bar_config(disable=True)
        t0 = time.time()
        lat_all = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1,
                                                     text_embeddings=emb_src, store_attention=True,
                                                     LOW_RESOURCE=True, save_path=None)
        t_inv = time.time() - t0
        pipe._encode_prompt = lambda *a, **k: emb_tgt
        pipe.decode_latents = lambda lat: lat
        attention_util.SpatialBlender = _SB
        attention_util.show_cross_attention = lambda *a, **k: None
        kwargs = dict(prompt=tgt, source_prompt=src, edit_type="swap", image=None, strength=None,
                      num_inference_steps=T, guidance_scale=7.5, num_images_per_prompt=1,
                      latents=lat_all[-1], save_path="/tmp/unused", is_replace_controller=is_rep,
                      cross_replace_steps=dict(crs), use_inversion_attention=True, blend_th=[0.3, 0.3],
                      save_self_attention=False)
        if bw is not None:
            kwargs["blend_words"] = bw
        if eq is not None:
            kwargs["eq_params"] = eq
        kwargs.update(ov)
        if bw is None:
            kwargs.pop("blend_self_attention", None)
            kwargs.pop("blend_latents", None)
        stash = {}
        _mk = attention_util.make_controller

        def _mk_spy(*a, **k):
            stash["ctrl"] = _mk(*a, **k)
            return stash["ctrl"]
        attention_util.make_controller = _mk_spy
        t0 = time.time()
        out = pipe(**kwargs)
        t_edit = time.time() - t0
        attention_util.make_controller = _mk
        attention_util.SpatialBlender = SpatialBlender
        edited = out["sdimage_output"].images
        if isinstance(edited, list):  # the reference's numpy_to_pil wraps a 5-D batch into a list of sequences
            edited = torch.stack(edited)
        store = pipe.store_controller
        arrays = {"z0": lat_all[0], "zT": lat_all[-1], "emb_src": emb_src, "emb_tgt": emb_tgt, "edited": edited,
                  "inv_latents_digest": np.array([[d["sum"], d["wsum"], d["abs"]] for d in map(tensor_digest, lat_all)])}
        if out["mask_list"] is not None:
            ml = torch.stack(out["mask_list"]).bool()
            arrays["latent_mask_bits"] = np.packbits(ml.numpy(), axis=None)
            arrays["latent_mask_shape"] = np.array(ml.shape)
        lb = getattr(stash["ctrl"], "latent_blend", None)
        if lb is not None and getattr(lb, "applied_mask_list", None):
            am = torch.stack(lb.applied_mask_list).bool()  # [blending steps, P - 1, F, h, w]
            arrays["latent_applied_mask_bits"] = np.packbits(am.numpy(), axis=None)
            arrays["latent_applied_mask_shape"] = np.array(am.shape)
        ab = stash["ctrl"].attention_blend
        ab_frac = None
        if ab is not None:  # attention-blend masks, in call order, grouped by resolution
            packed = {}
            for m in ab.mask_list:
                packed.setdefault(m.shape[-1], []).append(m.bool())
            for r, v in packed.items():
                st = torch.stack(v)
                arrays[f"attn_mask_r{r}_bits"] = np.packbits(st.numpy(), axis=None)
                arrays[f"attn_mask_r{r}_shape"] = np.array(st.shape)
            ab_frac = float(torch.cat([m.float().flatten() for m in ab.mask_list]).mean())
        # a few captured inversion maps, exactly as stored (step 0 and last), in fp16 to stay small
        m0 = store.attention_store_all_step[0]
        arrays["inv_step0_down_cross2"] = m0["down_cross"][min(2, len(m0["down_cross"]) - 1)].half()  # (72^2 latents capture only two)
        arrays["inv_step0_mid_self0"] = m0["mid_self"][0].half()
        save_npz(name + ".npz", **arrays)
        meta[name] = {"F": F_, "L": L, "T": T, "prompt_case": PROMPT_CASES[ci][0], "model_config": mc,
                      "kwargs": {k: v for k, v in kwargs.items() if k not in ("latents", "image")},
                      "seconds_inversion": t_inv, "seconds_edit": t_edit,
                      "map_digests_step0": {k: [tensor_digest(m) for m in v] for k, v in m0.items()},
                      "map_shapes": {k: [list(m.shape) for m in v] for k, v in m0.items()},
                      "timesteps": [int(t) for t in pipe.scheduler.timesteps]}
        mm = float(torch.stack(out["mask_list"]).float().mean()) if out["mask_list"] is not None else None
        print(f"  {name}: inv {t_inv:.1f}s edit {t_edit:.1f}s  attn-mask ones={ab_frac} latent-mask ones={mm}  |edited|={float(edited.abs().mean()):.4f}")
    with open(meta_path, "w") as f:
        json.dump(meta, f)


def main():
    os.makedirs(GOLD, exist_ok=True)
    args = sys.argv[1:]
    only, only_unet = set(), set()
    for a in list(args):  # "pipeline:name1,name2" / "unet:name" record just those scenarios
        if a.startswith("pipeline:"):
            only |= set(a.split(":", 1)[1].split(","))
            args[args.index(a)] = "pipeline"
        if a.startswith("unet:"):
            only_unet |= set(a.split(":", 1)[1].split(","))
            args[args.index(a)] = "unet"
    which = set(args) or {"const", "unet", "controller", "pipeline"}
    rec = RecordingTokenizer(load_bpe_tokenizer())
    torch.set_grad_enabled(False)
    if "const" in which:
        print("[const]"); gen_const(rec)
    if "unet" in which:
        print("[unet]"); gen_unet(only_unet or None)
    if "controller" in which:
        print("[controller]"); gen_controller(rec)
    if "pipeline" in which:
        print("[pipeline]"); gen_pipeline(rec, only or None)
    # tokenizer replay table (merge with an existing one so partial runs do not drop entries)
    path = os.path.join(GOLD, "tokenizer_replay.json")
    table = {"encode": {}, "decode": {}}
    if os.path.exists(path):
        table = json.load(open(path))
    table["encode"].update(rec.enc)
    table["decode"].update(rec.dec)
    json.dump(table, open(path, "w"))
    src = open(os.path.abspath(__file__), "rb").read()
    print("generator sha1", hashlib.sha1(src).hexdigest()[:12])


if __name__ == "__main__":
    main()
