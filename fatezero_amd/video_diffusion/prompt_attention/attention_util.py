"""Edit controllers and their factory (reference: video_diffusion/prompt_attention/attention_util.py).

`make_controller(...)` and the classes `EmptyControl`, `AttentionControlEdit`, `AttentionReplace`,
`AttentionRefine`, `AttentionReweight` keep the reference's names, constructor arguments and attributes.  What
changes is where the arithmetic happens: for the fused kernels a controller call is a *plan*
(`plan_controlled`): the inversion-time map of the matching (step, layer) in the HBM arena plus a handful of
per-step constants --

  cross-attention   new = (base @ M) * A + cur * B              (fz_attn_cross, INJECT)
      Replace (attention_util.py:213-223):  M = replacement mapper,        A0 = 1,      B0 = 0
      Refine  (:243-253):                   M = one-hot gather of mapper,  A0 = alphas, B0 = 1 - alphas
      Reweight(:282-286):                   (A0, B0) of the wrapped controller times the equalizer
      time gate (:129-132):                 A = alpha_t A0,  B = alpha_t B0 + (1 - alpha_t)
  self-attention    rows with blend-mask 0 take the stored map, rows with mask 1 keep the live attention
                    (:80-92, :136-151); without a blender every row takes the stored map and QK^T is skipped.

The reference's tensor protocol (`forward(attn, is_cross, place)`, `replace_cross_attention`,
`replace_self_attention`) is kept as plain torch code for drop-in use; the UNet of this package never calls it.
"""
import abc
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from ... import kernels as K
from ..models.attention import AttnPlan
from . import ptp_utils, seq_aligner
from .attention_register import register_attention_control  # noqa: F401  (re-exported like the reference)
from .attention_store import KEYS, MAX_CONTROLLED_TOKENS, AttentionControl, AttentionStore, CapturedMap
from .spatial_blend import SpatialBlender

MAX_WORDS = 77


class EmptyControl:
    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        return attn

    def attention_plan(self, is_cross, place, n_frames, clip_len, heads, lq, lk, device):
        return AttnPlan(n_frames)


from .visualization import show_cross_attention  # noqa: E402,F401  (attention_util.py:18 of the reference re-exports it)


class AttentionControlEdit(AttentionStore, abc.ABC):
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, latent_blend: Optional[SpatialBlender],
                 tokenizer=None, additional_attention_store: AttentionStore = None, use_inversion_attention: bool = False,
                 attention_blend: SpatialBlender = None, save_self_attention: bool = True, disk_store=False):
        super().__init__(save_self_attention=save_self_attention, disk_store=disk_store)
        self.additional_attention_store = additional_attention_store
        self.batch_size = len(prompts)
        self.attention_blend = attention_blend
        if self.additional_attention_store is not None:
            self.batch_size = len(prompts) // 2
            assert self.batch_size == 1, "Only support single video editing with additional attention_store"
        self.cross_replace_alpha = ptp_utils.get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer)
        if isinstance(self_replace_steps, float):
            self_replace_steps = 0, self_replace_steps
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.latent_blend = latent_blend
        self.prev_attention_key_name = 0
        self.use_inversion_attention = use_inversion_attention
        self.attention_position_counter_dict = {k: 0 for k in KEYS}
        # The reference stores EVERY live (un-edited) map of the edit pass (`super().forward`, attention_util.py:103) although
        # only two things ever read them: the latent blend (all cross maps <= 32x32) and `show_cross_attention(..., 16,
        # ["up", "down"])` at the end of the edit (p2p_ddim_spatial_temporal.py:211-215: the 16x16 cross maps).  Here a live
        # cross map is written (and summed) only when one of them will consume it.
        self.track_cross_attention = latent_blend is not None
        self.visualize_res = 16  # None: no `attention_output` (nothing tracked for it)
        self._coef_cache = {}
        self._mapper_t_dev = None

    # ---------------------------------------------------------------------------------------------------
    # constants of the fused cross-attention edit
    # ---------------------------------------------------------------------------------------------------
    @abc.abstractmethod
    def mapper_matrix(self) -> torch.Tensor:
        """[77, 77] float M with base' = base @ M."""

    @abc.abstractmethod
    def inner_coef(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(A0, B0), each [77]: replace_cross_attention(base, cur) == (base @ M) * A0 + cur * B0."""

    def cross_constants(self, step: int, device):
        if self._mapper_t_dev is None or self._mapper_t_dev.device != device:
            mt = torch.zeros(K.CROSS_KEYS, K.CROSS_KEYS)
            mt[:MAX_WORDS, :MAX_WORDS] = self.mapper_matrix().t()
            self._mapper_t_dev = mt.to(device=device, dtype=torch.float16).contiguous()
        key = str(device)
        if key not in self._coef_cache:
            # every step's (A, B) row pair in ONE device tensor, moved once: a host-to-device copy per step would be a stream
            # synchronisation per step (the first cross-attention layer of every edit forward would drain the GPU)
            a0, b0 = self.inner_coef()
            n = self.cross_replace_alpha.shape[0]
            alpha = self.cross_replace_alpha.reshape(n, -1)[:, :MAX_WORDS].float()
            coef = torch.zeros(n, 2, K.CROSS_KEYS)
            coef[:, 0, :MAX_WORDS] = alpha * a0
            coef[:, 1, :MAX_WORDS] = alpha * b0 + (1 - alpha)
            self._coef_cache[key] = coef.to(device)
        return self._mapper_t_dev, self._coef_cache[key][step]

    # ---------------------------------------------------------------------------------------------------
    def _step_in_store(self):
        n = len(self.additional_attention_store.attention_store_all_step)
        return n - self.cur_step - 1 if self.use_inversion_attention else self.cur_step  # attention_util.py:108-111

    def update_attention_position_dict(self, current_attention_key):
        self.attention_position_counter_dict[current_attention_key] += 1

    @property
    def issue_events_first(self):
        """plan_controlled hands out the step's stored maps, per-step constants, blend masks thresholded from the STORED cross maps: nothing of the live forward is read."""
        return type(self).plan_controlled is AttentionControlEdit.plan_controlled

    def issue_signature(self):
        if type(self).plan_controlled is not AttentionControlEdit.plan_controlled:
            return None
        in_self_window = self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]
        return ("edit", bool(self.LOW_RESOURCE), bool(self.save_self_attention), bool(in_self_window), self.attention_blend is not None,
                bool(self.track_cross_attention), self.visualize_res)

    def plan_controlled(self, is_cross, place, n_ctrl, clip_len, heads, lq, lk, device) -> AttnPlan:
        if lq > MAX_CONTROLLED_TOKENS:
            return AttnPlan(0)
        assert self.additional_attention_store is not None, "the edit needs the inversion-time AttentionStore"
        key = f"{place}_{'cross' if is_cross else 'self'}"
        pos = self.attention_position_counter_dict[key]
        sis = self._step_in_store()
        step_maps = self.additional_attention_store.maps_of_step(sis)
        base: CapturedMap = step_maps[key][pos]
        self.update_attention_position_dict(key)
        plan = AttnPlan(0)
        if is_cross:
            mapper_t, coef = self.cross_constants(self.cur_step, device)
            plan.mode, plan.p, plan.mapper_t, plan.coef = K.FZ_ATTN_INJECT, base.storage, mapper_t, coef
            if self.track_cross_attention or (self.visualize_res is not None and lq == self.visualize_res ** 2):
                plan.cur_out = self.new_slot(key, n_ctrl, heads, lq, lk, True, device).storage
            return plan
        if self.save_self_attention:  # only outside the 'swap' flow of the reference's validation loop
            plan.capture_first = self.new_slot(key, n_ctrl, heads, lq, lk, False, device).storage
        if self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]:
            plan.mode, plan.p = K.FZ_ATTN_INJECT, base.storage
            if self.attention_blend is not None:
                h = int(np.sqrt(lq))
                mask = self.attention_blend(target_h=h, target_w=h, attention_store=step_maps, step_in_store=sis)
                plan.row_mask = mask[0].reshape(mask.shape[1], h * h).contiguous()  # [F, Lq]: 1 keeps the live attention
        return plan

    # ---------------------------------------------------------------------------------------------------
    # reference tensor protocol (attention_util.py:80-158), plain torch
    # ---------------------------------------------------------------------------------------------------
    def replace_self_attention(self, attn_base, att_replace, reshaped_mask=None):
        if att_replace.shape[-2] <= MAX_CONTROLLED_TOKENS:
            attn_base = attn_base.to(att_replace.device, dtype=att_replace.dtype)
            attn_base = attn_base.unsqueeze(0).expand(att_replace.shape[0], *attn_base.shape)
            if reshaped_mask is not None:
                return reshaped_mask * att_replace + (1 - reshaped_mask) * attn_base
            return attn_base
        return att_replace

    def replace_cross_attention(self, attn_base, att_replace):
        a0, b0 = self.inner_coef()
        m = self.mapper_matrix().to(att_replace.device, att_replace.dtype)
        base = attn_base.to(att_replace.device, att_replace.dtype)
        return (base @ m)[None] * a0.to(att_replace) + att_replace * b0.to(att_replace)

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        super().forward(attn, is_cross, place_in_unet)
        if attn.shape[-2] <= MAX_CONTROLLED_TOKENS:
            key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
            pos = self.attention_position_counter_dict[key]
            sis = self._step_in_store()
            step_dict = self.additional_attention_store.attention_store_all_step[sis]
            attn_base = step_dict[key][pos]
            self.update_attention_position_dict(key)
            if is_cross or (self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]):
                clip_length = attn.shape[0] // self.batch_size
                attn = attn.reshape(self.batch_size, clip_length, *attn.shape[1:])
                if is_cross:
                    aw = self.cross_replace_alpha[self.cur_step].to(attn.device, attn.dtype)
                    attn = self.replace_cross_attention(attn_base, attn) * aw + (1 - aw) * attn
                else:
                    reshaped_mask = None
                    if self.attention_blend is not None:
                        h = int(np.sqrt(attn.shape[-2]))
                        mask = self.attention_blend(target_h=h, target_w=h,
                                                    attention_store=self.additional_attention_store.maps_of_step(sis),
                                                    step_in_store=sis)
                        reshaped_mask = mask.permute(1, 0, 2, 3).reshape(mask.shape[1], mask.shape[0], h * h)[..., None].to(attn.dtype)
                    attn = self.replace_self_attention(attn_base, attn, reshaped_mask)
                attn = attn.reshape(self.batch_size * clip_length, *attn.shape[2:])
        return attn

    def between_steps(self):
        super().between_steps()
        self.attention_position_counter_dict = {k: 0 for k in KEYS}

    def step_callback(self, x_t):
        x_t = super().step_callback(x_t)
        if self.latent_blend is None:
            return x_t
        store = self.additional_attention_store
        sis = len(store.latents_store) - self.cur_step if self.use_inversion_attention else self.cur_step  # :52-55
        inverted = store.latents_store[sis].to(device=x_t.device, dtype=x_t.dtype)
        base_maps = store.maps_of_step(sis)
        blend = self.get_empty_cross_store()
        inv_steps = 1.0 / float(self.cur_step)
        for key in blend:
            for i, cm in enumerate(base_maps[key]):
                # [source inversion map ; accumulated live map] (attention_util.py:68-76).  The accumulated map is
                # scaled by 1/steps before the fp16 cast: get_mask normalises per (prompt, frame), so the scale is free.
                acc = self._sum_storage[key][i]
                pair = torch.stack([cm.storage, (acc * inv_steps).to(torch.float16)], dim=0)
                blend[key].append(pair)
        x_t = self.latent_blend(x_t=torch.cat([inverted, x_t], dim=0), attention_store=blend)
        return x_t[1:, ...]


class AttentionReplace(AttentionControlEdit):
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, latent_blend=None, tokenizer=None,
                 additional_attention_store=None, use_inversion_attention=False, attention_blend=None,
                 save_self_attention: bool = True, disk_store=False):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend, tokenizer=tokenizer,
                         additional_attention_store=additional_attention_store, use_inversion_attention=use_inversion_attention,
                         attention_blend=attention_blend, save_self_attention=save_self_attention, disk_store=disk_store)
        self.mapper = seq_aligner.get_replacement_mapper(prompts, tokenizer)  # [1, 77, 77]

    def mapper_matrix(self):
        return self.mapper[0].float()

    def inner_coef(self):
        return torch.ones(MAX_WORDS), torch.zeros(MAX_WORDS)


class AttentionRefine(AttentionControlEdit):
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, latent_blend=None, tokenizer=None,
                 additional_attention_store=None, use_inversion_attention=False, attention_blend=None,
                 save_self_attention: bool = True, disk_store=False):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend, tokenizer=tokenizer,
                         additional_attention_store=additional_attention_store, use_inversion_attention=use_inversion_attention,
                         attention_blend=attention_blend, save_self_attention=save_self_attention, disk_store=disk_store)
        self.mapper, alphas = seq_aligner.get_refinement_mapper(prompts, tokenizer)
        self.alphas = alphas.reshape(alphas.shape[0], 1, 1, alphas.shape[1])

    def mapper_matrix(self):
        m = torch.zeros(MAX_WORDS, MAX_WORDS)
        idx = self.mapper[0]
        cols = torch.arange(MAX_WORDS)
        ok = idx >= 0  # -1 marks an inserted target token: its alpha is 0, the column stays empty
        m[idx[ok], cols[ok]] = 1.0
        return m

    def inner_coef(self):
        al = self.alphas.reshape(-1)[:MAX_WORDS].float()
        return al, 1 - al


class AttentionReweight(AttentionControlEdit):
    """First replace / refine, then scale the attention of the selected words (attention_util.py:275-304)."""

    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, equalizer, latent_blend=None,
                 controller: Optional[AttentionControlEdit] = None, tokenizer=None, additional_attention_store=None,
                 use_inversion_attention=False, attention_blend=None, save_self_attention: bool = True, disk_store=False):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend, tokenizer=tokenizer,
                         additional_attention_store=additional_attention_store, use_inversion_attention=use_inversion_attention,
                         attention_blend=attention_blend, save_self_attention=save_self_attention, disk_store=disk_store)
        self.equalizer = equalizer
        self.prev_controller = controller

    def mapper_matrix(self):
        return self.prev_controller.mapper_matrix() if self.prev_controller is not None else torch.eye(MAX_WORDS)

    def inner_coef(self):
        eq = self.equalizer.reshape(-1)[:MAX_WORDS].float()
        if self.prev_controller is None:
            return eq, torch.zeros(MAX_WORDS)
        a0, b0 = self.prev_controller.inner_coef()
        return a0 * eq, b0 * eq


def get_equalizer(text: str, word_select, values, tokenizer=None):
    """attention_util.py:307-316."""
    if isinstance(word_select, (int, str)):
        word_select = (word_select,)
    equalizer = torch.ones(1, MAX_WORDS)
    for word, val in zip(word_select, values):
        inds = ptp_utils.get_word_inds(text, word, tokenizer)
        equalizer[:, inds] = val
    return equalizer


def make_controller(tokenizer, prompts: List[str], is_replace_controller: bool, cross_replace_steps: Dict[str, float],
                    self_replace_steps: float = 0.0, blend_words=None, equilizer_params=None,
                    additional_attention_store=None, use_inversion_attention=False, blend_th=(0.3, 0.3),
                    NUM_DDIM_STEPS=None, blend_latents=False, blend_self_attention=False, save_path=None,
                    save_self_attention=True, disk_store=False) -> AttentionControlEdit:
    """attention_util.py:320-387.  `save_path` feeds the blend-mask PNG dumps (`<save_path>/latent_blend_mask`,
    `<save_path>/attention_blend_mask`, written off the hot loop: spatial_blend.py); unlike the reference, `save_path=None`
    together with `blend_words` is accepted and simply dumps nothing."""
    latent_blend = attention_blend = None
    if not ((blend_words is None) or (blend_words == "None")):
        if blend_latents:
            latent_blend = SpatialBlender(prompts, blend_words, start_blend=0.2, end_blend=0.8, tokenizer=tokenizer,
                                          th=blend_th, NUM_DDIM_STEPS=NUM_DDIM_STEPS, prompt_choose="both",
                                          save_path=None if save_path is None else save_path + "/latent_blend_mask")
        if blend_self_attention:
            attention_blend = SpatialBlender(prompts, blend_words, start_blend=0.0, end_blend=2, tokenizer=tokenizer,
                                             th=blend_th, NUM_DDIM_STEPS=NUM_DDIM_STEPS, prompt_choose="source",
                                             save_path=None if save_path is None else save_path + "/attention_blend_mask")
    common = dict(cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps, latent_blend=latent_blend,
                  tokenizer=tokenizer, additional_attention_store=additional_attention_store,
                  use_inversion_attention=use_inversion_attention, attention_blend=attention_blend,
                  save_self_attention=save_self_attention, disk_store=disk_store)
    cls = AttentionReplace if is_replace_controller else AttentionRefine
    controller = cls(prompts, NUM_DDIM_STEPS, **common)
    if equilizer_params is not None:
        eq = get_equalizer(prompts[1], equilizer_params["words"], equilizer_params["values"], tokenizer=tokenizer)
        controller = AttentionReweight(prompts, NUM_DDIM_STEPS, equalizer=eq, controller=controller, **common)
    return controller
