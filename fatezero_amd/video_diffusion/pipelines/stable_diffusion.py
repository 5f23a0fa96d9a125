"""Base pipeline (reference: video_diffusion/pipelines/stable_diffusion.py, the parts P2pDDIMSpatioTemporalPipeline
inherits: module registry, scheduler fix-ups, `_encode_prompt`, `decode_latents`, `prepare_extra_step_kwargs`,
`numpy_to_pil`, progress-bar plumbing).  No diffusers dependency: `vae`, `text_encoder` and `tokenizer` are
duck-typed (diffusers / transformers objects work; the VAE and CLIP encoders themselves are SURVEY §8f "next")."""
import inspect
from dataclasses import dataclass
from typing import Any, List, Optional, Union

import numpy as np
import torch


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Any = None


class _NullBar:
    def __init__(self, iterable=None, total=None, **kw):
        self.iterable = iterable

    def __iter__(self):
        return iter(self.iterable)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        pass


class SpatioTemporalStableDiffusionPipeline:
    _optional_components = []

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):
        cfg = getattr(scheduler, "config", None)
        if cfg is not None:  # stable_diffusion.py:56-81
            if getattr(cfg, "steps_offset", 1) != 1:
                cfg.steps_offset = 1
            if getattr(cfg, "clip_sample", False) is True:
                cfg.clip_sample = False
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        vcfg = getattr(vae, "config", None)
        boc = getattr(vcfg, "block_out_channels", None) if vcfg is not None else None
        self.vae_scale_factor = 2 ** (len(boc) - 1) if boc is not None else 8
        self._progress_bar_config = {}

    # -- plumbing ----------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.unet.device

    def to(self, device):
        for m in (self.vae, self.text_encoder, self.unet):
            if hasattr(m, "to"):
                m.to(device)
        return self

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable=None, total=None):
        if self._progress_bar_config.get("disable", False):
            return _NullBar(iterable, total)
        try:
            from tqdm.auto import tqdm
        except Exception:
            return _NullBar(iterable, total)
        return tqdm(iterable, **self._progress_bar_config) if iterable is not None else tqdm(total=total, **self._progress_bar_config)

    def prepare_before_train_loop(self, params_to_optimize=None):
        for m in (self.vae, self.unet, self.text_encoder):
            if isinstance(m, torch.nn.Module):
                m.requires_grad_(False)
                m.eval()

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        """No-op: flash attention (no materialised score matrix) is the only attention path of this build."""

    def disable_xformers_memory_efficient_attention(self, *a, **k):
        pass

    def enable_vae_slicing(self):
        if hasattr(self.vae, "enable_slicing"):
            self.vae.enable_slicing()

    def disable_vae_slicing(self):
        if hasattr(self.vae, "disable_slicing"):
            self.vae.disable_slicing()

    # -- text ---------------------------------------------------------------------------------------------
    def _embed(self, prompts: List[str], device):
        tok = self.tokenizer(prompts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt")
        ids = tok.input_ids if hasattr(tok, "input_ids") else tok["input_ids"]
        return self.text_encoder(ids.to(device))[0]

    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None):
        """stable_diffusion.py:180-295: [uncond ; text] embeddings, each repeated num_images_per_prompt times."""
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        bs = len(prompts)
        emb = self._embed(prompts, device)
        emb = emb.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, emb.shape[1], -1)
        if do_classifier_free_guidance:
            if negative_prompt is None:
                uncond = [""] * bs
            elif type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond = [negative_prompt]
            elif bs != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {bs}. Please make sure that passed `negative_prompt` matches"
                                 " the batch size of `prompt`.")
            else:
                uncond = list(negative_prompt)
            un = self._embed(uncond, device)
            un = un.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, un.shape[1], -1)
            emb = torch.cat([un, emb])
        return emb

    # -- images -------------------------------------------------------------------------------------------
    def decode_latents(self, latents):
        """stable_diffusion.py:297-319: VAE decode in chunks of 16 frames -> float numpy [b, f, h, w, c] in [0, 1]."""
        if self.vae is None:
            raise RuntimeError("decode_latents needs a VAE; pass output_type='latent' to keep the latents")
        is_video = latents.dim() == 5
        b = latents.shape[0]
        latents = 1 / 0.18215 * latents
        if is_video:
            f = latents.shape[2]
            latents = latents.permute(0, 2, 1, 3, 4).reshape(b * f, *latents.shape[1:2], *latents.shape[3:])
        image = torch.cat([self.vae.decode(l).sample for l in torch.split(latents, 16, dim=0)], dim=0)
        image = (image / 2 + 0.5).clamp(0, 1).cpu().float()
        image = image.permute(0, 2, 3, 1).numpy()
        return image.reshape(b, -1, *image.shape[1:]) if is_video else image

    @staticmethod
    def numpy_to_pil(images):
        """stable_diffusion.py:566-576: a list (per batch element) of lists of PIL frames."""
        from PIL import Image

        def seq(arr):
            arr = (arr * 255).round().astype("uint8")
            return [Image.fromarray(a.squeeze()) for a in arr]
        if len(images.shape) == 5:
            return [seq(s) for s in images]
        return [seq(images)]

    def prepare_extra_step_kwargs(self, generator, eta):
        keys = set(inspect.signature(self.scheduler.step).parameters.keys())
        extra = {}
        if "eta" in keys:
            extra["eta"] = eta
        if "generator" in keys:
            extra["generator"] = generator
        return extra

    def print_pipeline(self, logger=None):
        import sys
        from ... import kernels as K
        lines = [f"{self.__class__}", f"python {sys.version}", f"torch {torch.__version__}", f"native: {K.version()}"]
        if torch.cuda.is_available():
            lines.append(f"device: {torch.cuda.get_device_name(0)}")
        for l in lines:
            (logger.info if logger is not None else print)(l)
