"""Attention controllers, base + capture store (reference: video_diffusion/prompt_attention/attention_store.py).

Same public attributes and call protocol as the reference (`__call__(attn, is_cross, place_in_unet)`,
`step_callback`, `between_steps`, `step_store`, `attention_store`, `attention_store_all_step`, `latents_store`,
`LOW_RESOURCE`, `cur_step`, `cur_att_layer`, `num_att_layers`, `get_average_attention`, `reset`) but the maps are
never copied to the host: every captured probability map is written by the attention kernel itself, coalesced,
into an fp16 slab of HBM (`MapArena`), and `attention_store_all_step[step][key][i]` is a *view* of that slab with
the reference's shape [F, heads, Lq, Lk].  Cross maps are stored with a row stride of 80 halves (16-byte aligned
rows) and exposed as `[..., :77]` views.

Two entry points:
  * `attention_plan(...)`  -- what the fused kernels use (fatezero_amd/video_diffusion/models/attention.py);
  * `__call__(attn, ...)`   -- the reference's tensor protocol, kept for drop-in use with foreign attention code.
Differences, by design: the running sum `attention_store` (attention_store.py:95-101) is kept in fp32 and only for
cross maps unless `accumulate_self=True` (its self-attention entries have no live consumer in the reference);
`latents_store` stays on the device; `disk_store=True` is accepted and ignored (288 GB of HBM replace the disk).
"""
import abc
from typing import Dict, List, Optional

import torch

from ... import kernels as K
from ..models.attention import AttnPlan

KEYS = ("down_cross", "mid_cross", "up_cross", "down_self", "mid_self", "up_self")
MAX_CONTROLLED_TOKENS = 32 ** 2  # attention_store.py:83, attention_util.py:104


class AttentionControl(abc.ABC):
    def __init__(self):
        self.LOW_RESOURCE = False  # assume the edit has classifier-free guidance (attention_store.py:56)
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    # -- reference protocol ------------------------------------------------------------------------------
    def step_callback(self, x_t):
        self.cur_att_layer = 0
        self.cur_step += 1
        self.between_steps()
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return 0

    @abc.abstractmethod
    def forward(self, attn, is_cross: bool, place_in_unet: str):
        raise NotImplementedError

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            if self.LOW_RESOURCE:
                attn = self.forward(attn, is_cross, place_in_unet)
            else:
                h = attn.shape[0]
                attn[h // 2:] = self.forward(attn[h // 2:], is_cross, place_in_unet)
        self.cur_att_layer += 1
        return attn

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    # -- kernel protocol ---------------------------------------------------------------------------------
    def attention_plan(self, is_cross, place, n_frames, clip_len, heads, lq, lk, device) -> AttnPlan:
        n_plain = 0 if self.LOW_RESOURCE else n_frames // 2
        plan = self.plan_controlled(is_cross, place, n_frames - n_plain, clip_len, heads, lq, lk, device)
        plan.n_plain = n_plain
        self.cur_att_layer += 1
        return plan

    def plan_controlled(self, is_cross, place, n_ctrl, clip_len, heads, lq, lk, device) -> AttnPlan:
        return AttnPlan(0)


class MapArena:
    """HBM slabs for captured maps. The first step sizes the slab; later steps take one allocation each (or a
    single pre-reserved block when the number of steps is known).  Reserved blocks (tens of GB) are recycled
    through a process-wide pool when their store dies, so a second job never pays hipMalloc/hipFree for them."""
    _pool = []  # released reserved blocks (uint8 tensors), largest reuse wins

    @classmethod
    def _take(cls, nbytes, device):
        best = None
        for i, t in enumerate(cls._pool):
            if t.device == torch.device(device) and t.numel() >= nbytes and (best is None or t.numel() < cls._pool[best].numel()):
                best = i
        if best is not None:
            return cls._pool.pop(best)
        cls._pool.clear()  # nothing fits: let the allocator have the old blocks back before asking for a bigger one
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=device)
        except torch.OutOfMemoryError as e:  # e.g. 32 frames x 512^2 x 50 steps with index [-1, 'first']: 299 GB
            raise RuntimeError(
                f"the attention-map arena of this job needs {nbytes / 1e9:.1f} GB of HBM in one block and does not fit on {device}: "
                "the maps are kept on the device by design (the reference's host / disk store is not reproduced) -- use fewer frames or "
                "DDIM steps, a one-frame SparseCausalAttention_index (e.g. ['mid'] halves the self-attention maps), "
                "save_self_attention=False, or split the clip's frames over GPUs (fatezero_amd.dist.FrameShard)") from e

    def release(self):
        if self.reserved is not None:
            MapArena._pool.append(self.reserved)
            self.reserved = None

    def __init__(self):
        self.step_bytes = 0
        self.cur = None
        self.cur_off = 0
        self.reserved = None
        self.reserved_off = 0
        self.first_step_done = False
        self.total_bytes = 0

    def reserve(self, n_steps, device):
        if self.first_step_done and self.step_bytes and n_steps > 0 and self.reserved is None:
            self.reserved = MapArena._take(self.step_bytes * n_steps, device)
            self.reserved_off = 0
            self.total_bytes += self.step_bytes * n_steps

    def alloc(self, shape, device):
        nbytes = 2
        for s in shape:
            nbytes *= s
        nbytes = (nbytes + 255) // 256 * 256
        if not self.first_step_done:
            self.step_bytes += nbytes
            self.total_bytes += nbytes
            return torch.empty(shape, dtype=torch.float16, device=device)
        if self.cur is None or self.cur_off + nbytes > self.cur.numel():
            if self.reserved is not None and self.reserved_off + self.step_bytes <= self.reserved.numel():
                self.cur = self.reserved[self.reserved_off: self.reserved_off + self.step_bytes]
                self.reserved_off += self.step_bytes
            else:
                self.cur = torch.empty(max(self.step_bytes, nbytes), dtype=torch.uint8, device=device)
                self.total_bytes += self.cur.numel()
            self.cur_off = 0
        out = self.cur[self.cur_off: self.cur_off + 2 * _numel(shape)].view(torch.float16).view(shape)
        self.cur_off += nbytes
        return out

    def end_step(self):
        self.first_step_done = True
        self.cur = None
        self.cur_off = 0


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class CapturedMap:
    """One captured map: `storage` is what the kernels address ([F, heads, Lq, 80] for cross maps), `view` is the
    reference-shaped tensor handed to users ([F, heads, Lq, 77])."""
    __slots__ = ("storage", "view")

    def __init__(self, storage, lk):
        self.storage = storage
        self.view = storage[..., :lk] if storage.shape[-1] != lk else storage


class AttentionStore(AttentionControl):
    def __init__(self, save_self_attention: bool = True, disk_store=False, accumulate_self: bool = False):
        super().__init__()
        self.disk_store = disk_store  # accepted for API compatibility; maps stay in HBM
        if disk_store:
            import warnings
            warnings.warn("disk_store=True: this build keeps the attention maps in the HBM arena (74.7 GB for 8 frames x 50 steps "
                          "of the 288 GB); nothing is written to `store_dir`", stacklevel=2)
        self.store_dir = None
        self.save_self_attention = save_self_attention
        self.accumulate_self = accumulate_self
        self.expected_steps = None  # optional hint (pipeline sets it) so the arena is one allocation
        self.latents_store: List[torch.Tensor] = []
        self._init_state()

    def _init_state(self):
        self.step_store = self.get_empty_store()
        self.attention_store: Dict[str, List[torch.Tensor]] = {}
        self.attention_store_all_step: List[Dict[str, List[torch.Tensor]]] = []
        self._step_maps = {k: [] for k in KEYS}       # CapturedMap objects of the current step
        self._all_step_maps: List[Dict[str, List[CapturedMap]]] = []
        self._sum_storage: Dict[str, List[torch.Tensor]] = {}
        self.arena = MapArena()

    @staticmethod
    def get_empty_store():
        return {k: [] for k in KEYS}

    @staticmethod
    def get_empty_cross_store():
        return {"down_cross": [], "mid_cross": [], "up_cross": []}

    # -- capture -------------------------------------------------------------------------------------------
    def _wants(self, is_cross, lq):
        return lq <= MAX_CONTROLLED_TOKENS and (is_cross or self.save_self_attention)

    def new_slot(self, key, frames, heads, lq, lk, is_cross, device) -> CapturedMap:
        width = K.CROSS_P_STRIDE if is_cross else lk
        cm = CapturedMap(self.arena.alloc((frames, heads, lq, width), device), lk)
        self._step_maps[key].append(cm)
        self.step_store[key].append(cm.view)
        return cm

    def plan_controlled(self, is_cross, place, n_ctrl, clip_len, heads, lq, lk, device) -> AttnPlan:
        if not self._wants(is_cross, lq):
            return AttnPlan(0)
        key = f"{place}_{'cross' if is_cross else 'self'}"
        cm = self.new_slot(key, n_ctrl, heads, lq, lk, is_cross, device)
        return AttnPlan(0, K.FZ_ATTN_CAPTURE, p=cm.storage)

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        """Reference tensor protocol (attention_store.py:81-93): keep a copy of the map."""
        if self._wants(is_cross, attn.shape[-2]):
            key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
            f, heads, lq, lk = attn.shape
            cm = self.new_slot(key, f, heads, lq, lk, is_cross, attn.device)
            if cm.storage.shape[-1] != lk:
                cm.storage[..., lk:] = 0
            cm.view.copy_(attn)
        return attn

    def between_steps(self):
        # running sum (attention_store.py:95-101), fp32, cross maps (+ self maps on request)
        keys = [k for k in KEYS if k.endswith("cross") or self.accumulate_self]
        if len(self.attention_store) == 0:
            self.attention_store = self.get_empty_store()
            self._sum_storage = {k: [] for k in KEYS}
            for k in keys:
                for cm in self._step_maps[k]:
                    s = cm.storage.float()
                    self._sum_storage[k].append(s)
                    self.attention_store[k].append(s[..., : cm.view.shape[-1]] if s.shape[-1] != cm.view.shape[-1] else s)
        else:
            for k in keys:
                for acc, cm in zip(self._sum_storage[k], self._step_maps[k]):
                    K.accumulate(acc, cm.storage)
        self.attention_store_all_step.append(self.step_store)
        self._all_step_maps.append(self._step_maps)
        self.step_store = self.get_empty_store()
        self._step_maps = {k: [] for k in KEYS}
        first = not self.arena.first_step_done
        self.arena.end_step()
        if first and self.expected_steps and self.latents_store is not None:
            dev = None
            for k in KEYS:
                if self._all_step_maps[0][k]:
                    dev = self._all_step_maps[0][k][0].storage.device
                    break
            if dev is not None:
                self.arena.reserve(self.expected_steps - 1, dev)

    def step_callback(self, x_t):
        x_t = super().step_callback(x_t)
        self.latents_store.append(x_t.detach().clone())
        return x_t

    def maps_of_step(self, step_in_store) -> Dict[str, List[CapturedMap]]:
        return self._all_step_maps[step_in_store]

    def get_average_attention(self):
        return {key: [item / self.cur_step for item in self.attention_store[key]] for key in self.attention_store}

    def reset(self):
        super().reset()
        self.release_arena()
        self._init_state()

    def release_arena(self):
        """Drop every captured map and hand the reserved HBM block back to the pool."""
        arena = getattr(self, "arena", None)
        if arena is not None:
            self.step_store = self.get_empty_store()
            self.attention_store_all_step = []
            self._all_step_maps = []
            self._step_maps = {k: [] for k in KEYS}
            self.attention_store = {}
            self.latents_store = []
            if hasattr(self, "_sum_storage"):
                self._sum_storage = None
            arena.cur = None
            arena.release()
            # tensors handed out earlier (attention_store_all_step views, CapturedMap.storage) alias the released block and are
            # invalid from here on; a store that is used again reserves a fresh arena instead of allocating per step
            self.arena = type(arena)()

    def __del__(self):
        try:
            self.release_arena()
        except Exception:
            pass

    @property
    def arena_bytes(self):
        return self.arena.total_bytes
