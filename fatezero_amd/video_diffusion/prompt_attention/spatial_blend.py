"""Blend masks from cross-attention (reference: video_diffusion/prompt_attention/spatial_blend.py).

`SpatialBlender` keeps the reference's constructor, attributes (`alpha_layers`, `th`, `start_blend`, `end_blend`,
`counter`, `mask_list`, `prompt_choose`) and call signature, but the reduction
(sum over blend words -> mean over heads x layers -> 3x3 max-pool -> nearest resize -> per-(prompt, frame) max
normalisation -> threshold) is one HIP kernel (`fz_blend_mask`) reading the fp16 maps straight out of the HBM
arena, and the result is cached per (inversion step, resolution): the reference recomputes the very same mask for
each of the 11 self-attention layers of a step.

PNG dumps of the masks (`save_path`, spatial_blend.py:43-55): same directory layout, same file-name pattern
(`{save_path}/{prompt_choose}/step_in_store_{step:04d}/mask_{timestamp}_{count:02d}.png`), same picture
(`torchvision.utils.save_image(..., normalize=True)` of the frames as a grid: 8 per row, 2 pixels of padding) -- but
moved OFF the hot loop: the mask of a (step, resolution) is copied device -> pinned host memory once, asynchronously
on a side stream, and a writer thread encodes the files after the copy's event; the denoise loop never waits for it.
`flush_mask_dumps()` (called by the pipeline after the edit) joins the outstanding writes.
"""
import datetime
import os
import queue
import threading
from typing import List

import numpy as np
import torch

from ... import kernels as K
from . import ptp_utils
from .attention_store import CapturedMap


def mask_grid_image(mask: np.ndarray, nrow: int = 8, padding: int = 2) -> np.ndarray:
    """What `tvu.save_image(rearrange(mask, "c p h w -> p c h w"), path, normalize=True)` writes for a 0/1 mask [F, h, w]
    (torchvision.utils.make_grid + save_image [3P torchvision]): min-max normalisation over the whole batch
    ((x - min) / (max - min + 1e-5): a constant mask comes out black), grey -> RGB, frames on a grid of `nrow` columns with
    `padding` black pixels around each (one frame: no grid, no padding), x 255 + 0.5 truncated to uint8.  Returns uint8
    [H, W, 3]."""
    m = mask.astype(np.float32)
    lo, hi = float(m.min()), float(m.max())
    m = (np.clip(m, lo, hi) - lo) / max(hi - lo, 1e-5)
    f, h, w = m.shape
    if f == 1:
        grid = m[0]
    else:
        xmaps = min(nrow, f)
        ymaps = -(-f // xmaps)
        hh, ww = h + padding, w + padding
        grid = np.zeros((hh * ymaps + padding, ww * xmaps + padding), np.float32)
        for k in range(f):
            y, x = divmod(k, xmaps)
            grid[y * hh + padding: y * hh + padding + h, x * ww + padding: x * ww + padding + w] = m[k]
    img = np.clip(grid * 255.0 + 0.5, 0, 255).astype(np.uint8)
    return np.repeat(img[:, :, None], 3, axis=2)


class _MaskDumper:
    """Writer thread for the blend-mask PNGs: jobs are (event or None, host uint8 tensor [F, h, w], list of paths)."""

    def __init__(self):
        self.q = queue.Queue()
        self.thread = None
        self.lock = threading.Lock()
        self.streams = {}
        self.errors = []

    def _run(self):
        while True:
            job = self.q.get()
            try:  # EVERYTHING a job can raise sits inside: each get() is answered by exactly one task_done(), or flush() would hang
                if job is None:
                    return
                from PIL import Image
                ev, host, paths = job
                if ev is not None:
                    ev.synchronize()
                img = Image.fromarray(mask_grid_image(host.numpy()))
                for path in paths:
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    img.save(path)
            except Exception as e:  # reported by flush(); a failed dump must not kill the edit
                self.errors.append(repr(e))
            finally:
                self.q.task_done()

    def _ensure_thread(self):
        with self.lock:
            if self.thread is None or not self.thread.is_alive():
                self.thread = threading.Thread(target=self._run, name="fz-mask-dump", daemon=True)
                self.thread.start()

    def stage(self, mask: torch.Tensor):
        """Start the device -> host copy of a 0/1 mask [F, h, w]; returns (event, host tensor) for `write`."""
        m8 = mask.to(torch.uint8)
        if not m8.is_cuda:
            return None, m8.contiguous().clone()
        dev = m8.device
        side = self.streams.get(dev)
        if side is None:
            side = self.streams[dev] = torch.cuda.Stream(device=dev)
        host = torch.empty(m8.shape, dtype=torch.uint8, pin_memory=True)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            host.copy_(m8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        m8.record_stream(side)
        return ev, host

    def write(self, staged, paths):
        self._ensure_thread()
        self.q.put((staged[0], staged[1], list(paths)))

    def flush(self, raise_on_error=False):
        """Wait for the queued dumps.  A failed dump (disk full, permissions, no PIL) must not cost the caller the finished edit:
        the errors come back as a list (and as a warning) unless `raise_on_error`."""
        if self.thread is not None:
            import queue
            import time
            while self.q.unfinished_tasks:  # (q.join() would wait forever for jobs nobody will take if the writer died)
                if not self.thread.is_alive():
                    dropped = 0
                    while True:  # answer every job nobody will take any more, through the queue's own protocol
                        try:
                            self.q.get_nowait()
                        except queue.Empty:
                            break
                        self.q.task_done()
                        dropped += 1
                    self.errors.append(f"writer thread died; {dropped} queued dump(s) dropped")
                    break
                time.sleep(0.01)
        errs, self.errors = self.errors, []
        if errs:
            if raise_on_error:
                raise RuntimeError("blend-mask PNG dump failed: " + "; ".join(errs))
            import warnings
            warnings.warn("blend-mask PNG dump failed (the edit itself is unaffected): " + "; ".join(errs))
        return errs


_DUMPER = _MaskDumper()


def flush_mask_dumps(raise_on_error=False):
    """Wait until every blend-mask PNG queued so far is on disk (the pipeline calls it once after the denoise loop).  Returns the list
    of dump errors (empty when all files were written); warns instead of raising unless `raise_on_error`."""
    return _DUMPER.flush(raise_on_error)


class SpatialBlender:
    def __init__(self, prompts: List[str], words, substruct_words=None, start_blend=0.2, end_blend=0.8,
                 th=(0.9, 0.9), tokenizer=None, NUM_DDIM_STEPS=None, save_path=None, prompt_choose="source"):
        self.count = 0
        self.MAX_NUM_WORDS = 77
        self.NUM_DDIM_STEPS = NUM_DDIM_STEPS
        self.save_path = save_path  # mask PNG dumps: written off-loop (module docstring)
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
        self.applied_mask_list = []  # (extension) what blends the EDITED latents each step: mask[1:], the source mask OR-ed with the target one
        self.dumped_mask_list = []   # (extension, only filled with save_path) the very rows each PNG was drawn from: mask[-1]
        self._alpha_dev = {}
        self._cache = {}
        self._staged = {}

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

    def _dump(self, mask, step_in_store, cache_key):
        """spatial_blend.py:43-55: one PNG per get_mask call (the last prompt's mask when there are two), numbered by
        `self.count`.  The D2H copy of a cached mask is staged once; every call only queues a file name."""
        now = datetime.datetime.now().strftime("%Y-%m-%dT%H-%M-%S")
        path = f"{self.save_path}/{self.prompt_choose}/"
        if step_in_store is not None:
            path += f"step_in_store_{step_in_store:04d}"
        path += f"/mask_{now}_{self.count:02d}.png"
        self.count += 1
        self.dumped_mask_list.append(mask[-1])
        staged = self._staged.get(cache_key) if cache_key is not None else None
        if staged is None:
            staged = _DUMPER.stage(mask[-1])
            if cache_key is not None:
                self._staged[cache_key] = staged
        _DUMPER.write(staged, [path])

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
                    self._staged.clear()
                self._cache[cache_key] = mask
        if self.save_path is not None:
            self._dump(mask, step_in_store, cache_key)
        # mask is one: use generated information; zero: use inverted information (spatial_blend.py:113-115)
        self.mask_list.append(mask[0][:, None, :, :])
        if x_t is not None:
            m = mask[:, None, ...] if x_t.dim() == 5 else mask
            if (self.counter > self.start_blend) and (self.counter < self.end_blend):
                self.applied_mask_list.append(mask[1:])
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
