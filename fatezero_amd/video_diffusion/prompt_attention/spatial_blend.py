"""Blend masks from cross-attention (reference: video_diffusion/prompt_attention/spatial_blend.py).

`SpatialBlender` keeps the reference's constructor, attributes (`alpha_layers`, `th`, `start_blend`, `end_blend`,
`counter`, `mask_list`, `prompt_choose`) and call signature, but the reduction
(sum over blend words -> mean over heads x layers -> 3x3 max-pool -> nearest resize -> per-(prompt, frame) max
normalisation -> threshold) is one HIP kernel (`fz_blend_mask`) reading the fp16 maps straight out of the HBM
arena, and the result is cached per (inversion step, resolution): the reference recomputes the very same mask for
each of the 11 self-attention layers of a step.  PNG dumps of the masks (`save_path`) are host I/O
inside the hot loop in the reference and are not reproduced.
"""
from typing import List

import numpy as np
import torch

from ... import kernels as K
from . import ptp_utils
from .attention_store import CapturedMap


class SpatialBlender:
    def __init__(self, prompts: List[str], words, substruct_words=None, start_blend=0.2, end_blend=0.8,
                 th=(0.9, 0.9), tokenizer=None, NUM_DDIM_STEPS=None, save_path=None, prompt_choose="source"):
        self.count = 0
        self.MAX_NUM_WORDS = 77
        self.NUM_DDIM_STEPS = NUM_DDIM_STEPS
        self.save_path = None  # mask PNG dumps are not reproduced (see module docstring)
        assert prompt_choose in ["source", "both"], \
            "choose to generate the mask by only source prompt or both the source and target"
        if substruct_words is not None:
            raise NotImplementedError("substruct_words is never set by make_controller (attention_util.py:336-351)")
        self.substruct_layers = None
        self.prompt_choose = prompt_choose
        alpha_layers = torch.zeros(len(prompts), 1, 1, 1, 1, self.MAX_NUM_WORDS)
        for i, (prompt, words_) in enumerate(zip(prompts, words)):
            if isinstance(words_, str):
                words_ = [words_]
            for word in words_:
                ind = ptp_utils.get_word_inds(prompt, word, tokenizer)
                alpha_layers[i, :, :, :, :, ind] = 1
        self.alpha_layers = alpha_layers
        self.start_blend = int(start_blend * self.NUM_DDIM_STEPS)
        self.end_blend = int(end_blend * self.NUM_DDIM_STEPS)
        self.counter = 0
        self.th = th
        self.mask_list = []
        self._alpha_dev = {}
        self._cache = {}

    def _alpha80(self, n_prompts, device):
        key = (n_prompts, str(device))
        if key not in self._alpha_dev:
            a = torch.zeros(n_prompts, 80, dtype=torch.float32)
            a[:, : self.MAX_NUM_WORDS] = self.alpha_layers.reshape(-1, self.MAX_NUM_WORDS)[:n_prompts]
            self._alpha_dev[key] = a.to(device)
        return self._alpha_dev[key]

    @staticmethod
    def select_maps(store_dict):
        """spatial_blend.py:78: `down_cross[2:4] + up_cross[:3]` -- by list position, not by resolution."""
        return list(store_dict["down_cross"][2:4]) + list(store_dict["up_cross"][:3])

    def mask_from_storage(self, maps5: List[torch.Tensor], target_h, target_w):
        """maps5: fp16 storages [P, F, heads, r*r, 80]. Returns float mask [P, F, h, w] of 0/1."""
        n_prompts = maps5[0].shape[0]
        res = {m.shape[3] for m in maps5}
        if len(res) != 1:  # the reference's torch.cat(dim=1) raises on this (SURVEY App. A, 256^2 inputs)
            raise RuntimeError(f"blend-mask maps have different resolutions {sorted(res)}: blend_words needs the "
                               "512^2 list layout (five 16x16 cross maps)")
        alpha = self._alpha80(n_prompts, maps5[0].device)
        return K.blend_mask(maps5, alpha, float(self.th[0]), (target_h, target_w),
                            or_with_first=(self.prompt_choose == "both"))

    def __call__(self, attention_store, step_in_store: int = None, target_h=None, target_w=None, x_t=None):
        """attention_store: dict of lists of maps ([F,heads,r*r,77] or [P,F,heads,r*r,77] tensors, or CapturedMap)."""
        if target_h is None and target_w is None and x_t is not None:
            target_h, target_w = x_t.shape[-2:]
        self.counter += 1
        cache_key = (step_in_store, target_h, target_w) if (x_t is None and step_in_store is not None) else None
        mask = self._cache.get(cache_key) if cache_key is not None else None
        if mask is None:
            storages = []
            for item in self.select_maps(attention_store):
                st = item.storage if isinstance(item, CapturedMap) else _as_storage(item)
                storages.append(st[None] if st.dim() == 4 else st)
            mask = self.mask_from_storage(storages, target_h, target_w)
            if cache_key is not None:
                if len(self._cache) > 8:
                    self._cache.clear()
                self._cache[cache_key] = mask
        # mask is one: use generated information; zero: use inverted information (spatial_blend.py:113-115)
        self.mask_list.append(mask[0][:, None, :, :])
        if x_t is not None:
            m = mask[:, None, ...] if x_t.dim() == 5 else mask
            if (self.counter > self.start_blend) and (self.counter < self.end_blend):
                x_t = x_t[:1] + m * (x_t - x_t[:1])
            return x_t
        return mask


def _as_storage(t: torch.Tensor) -> torch.Tensor:
    """A user-supplied map [..., 77] -> fp16 storage with 80-half rows."""
    if t.shape[-1] == K.CROSS_P_STRIDE and t.dtype == torch.float16 and t.is_contiguous():
        return t
    out = torch.zeros(*t.shape[:-1], K.CROSS_P_STRIDE, dtype=torch.float16, device=t.device)
    out[..., : t.shape[-1]] = t
    return out
