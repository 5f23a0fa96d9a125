"""Frame geometry of the dataset front-end (reference: video_diffusion/data/transform.py): crop a fixed border, scale the
short side with an anti-aliased bilinear filter, crop to a square.  All functions take a 4-D float tensor whose last two
dimensions are (height, width) -- the dataset hands over [c, f, h, w] -- and return a view / a resized copy."""
import random

import torch
import torch.nn.functional as F


def offset_crop(image: torch.Tensor, left: int = 0, right: int = 0, top: int = 200, bottom: int = 0) -> torch.Tensor:
    """Cut `left/right/top/bottom` pixels off the borders; each margin is clipped so that at least one pixel survives
    (transform.py:47-57 -- note the reference's default of top=200, kept because callers rely on passing all four)."""
    h, w = image.shape[-2:]
    left = min(left, w - 1)
    right = min(right, w - left - 1)
    top = min(top, h - 1)
    bottom = min(bottom, h - top - 1)
    return image[..., top:h - bottom, left:w - right]


def short_size_scale(images: torch.Tensor, size: int) -> torch.Tensor:
    """Resize so that the shorter side becomes `size`; the longer side is int(size / short * long) (transform.py:6-18)."""
    h, w = images.shape[-2:]
    if h < w:
        target = (size, int(size / h * w))
    else:
        target = (int(size / w * h), size)
    return F.interpolate(images, size=target, mode="bilinear", antialias=True)


def random_short_side_scale(images: torch.Tensor, size_min: int, size_max: int) -> torch.Tensor:
    return short_size_scale(images, random.randint(size_min, size_max))


def center_crop(images: torch.Tensor, height: int, width: int) -> torch.Tensor:
    h, w = images.shape[-2:]
    y0, x0 = (h - height) // 2, (w - width) // 2
    return images[..., y0:y0 + height, x0:x0 + width]


def random_crop(images: torch.Tensor, height: int, width: int) -> torch.Tensor:
    h, w = images.shape[-2:]
    y0, x0 = random.randint(0, h - height), random.randint(0, w - width)
    return images[..., y0:y0 + height, x0:x0 + width]
