"""Dataset front-end (reference: video_diffusion/data/dataset.py:15-160): a folder of frames ->
one clip `images [c, f, h, w]` in [-1, 1] plus the tokenised prompt, which test_fatezero.py:141-196 feeds to the VAE."""
import os
from pathlib import Path
from typing import Dict, Iterable, Optional

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from ..common.image_util import IMAGE_EXTENSION
from . import transform as T

_NO_OFFSET = {"left": 0, "right": 0, "top": 0, "bottom": 0}


class ImageSequenceDataset(Dataset):
    """Item i is the clip that starts at frame `start_sample_frame + stride * i` and takes every `sampling_rate`-th frame,
    `n_sample_frame` of them (all frames of the folder when negative).  `stride <= 0` means one clip per folder."""

    def __init__(self, path: str, prompt_ids: torch.Tensor, prompt: str, start_sample_frame: int = 0, n_sample_frame: int = 8,
                 sampling_rate: int = 1, stride: int = -1, image_mode: str = "RGB", image_size: int = 512, crop: str = "center",
                 class_data_root: Optional[str] = None, class_prompt_ids: Optional[torch.Tensor] = None,
                 offset: Optional[Dict[str, int]] = None, **args):
        self.path = path
        self.images = self.get_image_list(path)
        self.n_images = len(self.images)
        self.offset = dict(_NO_OFFSET if offset is None else offset)
        self.start_sample_frame = start_sample_frame
        self.n_sample_frame = self.n_images if n_sample_frame < 0 else n_sample_frame
        self.sampling_rate = sampling_rate
        self.sequence_length = (self.n_sample_frame - 1) * sampling_rate + 1  # span of one clip in source frames
        if self.n_images < self.sequence_length:
            raise ValueError(f"a clip spans {self.sequence_length} frames but {path} only holds {self.n_images}")
        self.stride = stride if stride > 0 else self.n_images + 1
        self.video_len = (self.n_images - self.sequence_length) // self.stride + 1
        self.image_mode, self.image_size = image_mode, image_size
        try:
            self.crop = {"center": T.center_crop, "random": T.random_crop}[crop]
        except KeyError:
            raise ValueError(f"crop must be 'center' or 'random', got {crop!r}")
        self.prompt, self.prompt_ids = prompt, prompt_ids
        if class_data_root is not None:  # regularisation images of one-shot tuning: part of the item format, unused by editing
            self.class_data_root = Path(class_data_root)
            self.class_images_path = sorted(self.class_data_root.iterdir())
            self.num_class_images = len(self.class_images_path)
            self.class_prompt_ids = class_prompt_ids

    def __len__(self) -> int:
        return max(self.video_len, getattr(self, "num_class_images", 0))

    def __getitem__(self, index: int) -> dict:
        frames = [self.load_frame(i) for i in self.get_frame_indices(index % self.video_len)]
        item = {"images": self.transform(frames), "prompt_ids": self.prompt_ids}
        if hasattr(self, "class_data_root"):
            first = index % (self.num_class_images - self.n_sample_frame)
            item["class_images"] = self.tensorize_frames([self.load_class_frame(i) for i in self.get_class_indices(first)])
            item["class_prompt_ids"] = self.class_prompt_ids
        return item

    def transform(self, frames) -> torch.Tensor:
        x = self.tensorize_frames(frames)
        x = T.offset_crop(x, **self.offset)
        x = T.short_size_scale(x, size=self.image_size)
        return self.crop(x, height=self.image_size, width=self.image_size)

    @staticmethod
    def tensorize_frames(frames) -> torch.Tensor:
        """list of PIL / HxWxC uint8 frames -> float [c, f, h, w] in [-1, 1]."""
        stack = np.stack([np.asarray(f) for f in frames])  # f h w c
        return torch.from_numpy(np.ascontiguousarray(stack.transpose(3, 0, 1, 2))).div(255) * 2 - 1

    def load_frame(self, index: int) -> Image.Image:
        return Image.open(os.path.join(self.path, self.images[index])).convert(self.image_mode)

    def load_class_frame(self, index: int) -> Image.Image:
        return Image.open(self.class_images_path[index]).convert(self.image_mode)

    def get_frame_indices(self, index: int) -> Iterable[int]:
        start = (self.start_sample_frame or 0) + self.stride * index
        return (start + i * self.sampling_rate for i in range(self.n_sample_frame))

    def get_class_indices(self, index: int) -> Iterable[int]:
        return (index + i for i in range(self.n_sample_frame))

    @staticmethod
    def get_image_list(path: str):
        return [f for f in sorted(os.listdir(path)) if f.endswith(IMAGE_EXTENSION)]
