"""Per-prompt / per-seed sample driver (reference: video_diffusion/pipelines/p2p_validation_loop.py:18-166).

`P2pSampleLogger` keeps the reference's constructor and `log_sample_images` call: for every editing prompt it picks the
edit type (`save` for the first prompt when no inversion attention is used, `swap` otherwise, None without prompt-to-prompt
editing), merges `p2p_config[idx]` into the pipeline keyword arguments, runs the pipeline once per seed and writes
`step_{step}_{idx}_{seed}.gif` (+ PNG folder, + mp4 when imageio exists), the attention strips next to it and the grids
`step_{step}.gif` / `step_{step}atten.gif`.  The call plan itself is `config_driver.plan_edits` (one implementation of the
reference's branching, shared with the YAML driver)."""
import os
from typing import List, Optional, Union

import numpy as np
import torch

from ... import config_driver
from ..common.image_util import annotate_image, make_grid, save_gif_mp4_folder_type


def tensor_to_numpy(image, b=1):
    """[(b f), c, h, w] in [-1, 1] -> [b, f, h, w, c] in [0, 1] (p2p_validation_loop.py:169-177)."""
    image = (image / 2 + 0.5).clamp(0, 1).cpu().float().numpy()
    bf, c, h, w = image.shape
    return image.reshape(b, bf // b, c, h, w).transpose(0, 1, 3, 4, 2)


class P2pSampleLogger:
    def __init__(self, editing_prompts: List[str], clip_length: int, logdir: str, subdir: str = "sample",
                 num_samples_per_prompt: int = 1, sample_seeds: List[int] = None, num_inference_steps: int = 20,
                 guidance_scale: float = 7, strength: float = None, annotate: bool = False, annotate_size: int = 15,
                 use_make_grid: bool = True, grid_column_size: int = 2, prompt2prompt_edit: bool = False,
                 p2p_config: dict = None, use_inversion_attention: bool = True, source_prompt: str = None,
                 traverse_p2p_config: bool = False, **args) -> None:
        self.editing_prompts = list(editing_prompts)
        self.clip_length = clip_length
        self.guidance_scale = guidance_scale
        self.num_inference_steps = num_inference_steps
        self.strength = strength
        if sample_seeds is None:
            limit = int(1e5)
            if num_samples_per_prompt > limit:
                raise ValueError
            sample_seeds = sorted(torch.randint(0, limit, (num_samples_per_prompt,)).numpy().tolist())
        self.sample_seeds = list(sample_seeds)
        self.logdir = os.path.join(logdir, subdir)
        os.makedirs(self.logdir)  # like the reference: an existing sample directory is an error
        self.annotate, self.annotate_size = annotate, annotate_size
        self.make_grid, self.grid_column_size = use_make_grid, grid_column_size
        self.prompt2prompt_edit = prompt2prompt_edit
        self.p2p_config = p2p_config
        self.use_inversion_attention = use_inversion_attention
        self.source_prompt = source_prompt
        self.traverse_p2p_config = traverse_p2p_config

    def _plan(self):
        return config_driver.plan_edits(
            dict(editing_prompts=self.editing_prompts, p2p_config=self.p2p_config, sample_seeds=self.sample_seeds,
                 use_inversion_attention=self.use_inversion_attention, prompt2prompt_edit=self.prompt2prompt_edit,
                 strength=self.strength, num_inference_steps=self.num_inference_steps, clip_length=self.clip_length,
                 guidance_scale=self.guidance_scale), self.source_prompt)

    def log_sample_images(self, pipeline, device: torch.device, step: int, image=None, latents: torch.FloatTensor = None,
                          uncond_embeddings_list: List[torch.FloatTensor] = None, save_dir=None):
        samples_all, attention_all = [], []
        if image is not None:
            frames = pipeline.numpy_to_pil(tensor_to_numpy(image))[0]
            if self.annotate:
                frames = [annotate_image(f, "input sequence", font_size=self.annotate_size) for f in frames]
            samples_all.append(frames)
        for call in self._plan():
            idx, seed, kw = call["prompt_index"], call["seed"], dict(call["kwargs"])
            # the reference seeds a generator ON the pipeline's device (p2p_validation_loop.py:108-109): noise drawn from it (strength-based
            # img2img / use_invertion_latents False) is the device generator's stream, so per-seed results match upstream's
            try:
                generator = torch.Generator(device=device).manual_seed(seed)
            except RuntimeError:  # a device without generator support (the CPU-emulation tests pass 'cpu' anyway)
                generator = torch.Generator(device="cpu").manual_seed(seed)
            ret = pipeline(image=image, generator=generator, latents=latents, uncond_embeddings_list=uncond_embeddings_list,
                           save_path=save_dir, **kw)
            attention_output = None
            if self.prompt2prompt_edit:
                sequence = ret["sdimage_output"].images[0]
                attention_output = ret["attention_output"]
            else:
                sequence = ret.images[0]
            images = [annotate_image(f, kw["prompt"], font_size=self.annotate_size) for f in sequence] if self.annotate else sequence
            if self.make_grid:
                samples_all.append(images)
                if attention_output is not None and len(attention_output) > 0:
                    attention_all.append(list(attention_output))
            save_path = os.path.join(self.logdir, f"step_{step}_{idx}_{seed}.gif")
            save_gif_mp4_folder_type(images, save_path)
            if attention_output is not None and len(attention_output) > 0:
                save_gif_mp4_folder_type(list(attention_output), save_path.replace(".gif", "atten.gif"))
        if self.make_grid:
            samples_all = [make_grid(images, cols=int(np.ceil(np.sqrt(len(samples_all))))) for images in zip(*samples_all)]
            save_path = os.path.join(self.logdir, f"step_{step}.gif")
            save_gif_mp4_folder_type(samples_all, save_path)
            if attention_all:
                grids = [make_grid(images, cols=1) for images in zip(*attention_all)]
                save_gif_mp4_folder_type(grids, save_path.replace(".gif", "atten.gif"))
        return samples_all
