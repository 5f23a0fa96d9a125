"""Image / clip writers of the sample logger (reference: video_diffusion/common/image_util.py:58-210), PIL only.

The reference writes every clip three times (`save_gif_mp4_folder_type`, image_util.py:159-168): an animated gif, an mp4
(imageio + ffmpeg) and a folder of numbered PNGs (cv2).  Neither imageio nor cv2 exists in this environment, so the gif
and the PNG folder are written with PIL and the mp4 only when imageio happens to be importable -- same paths, same names
(`x.gif`, `x.mp4`, `x/00000.png`).  Annotation uses PIL's built-in bitmap font instead of downloading OpenSans
(image_util.py:29-54): there is no network."""
import math
import os
import textwrap
from typing import List, Sequence, Union

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageFont

IMAGE_EXTENSION = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp", ".JPEG")


def to_pil(x) -> Image.Image:
    """PIL image from a PIL image, an HxW / HxWx{1,3} uint8 or [0,1] float array, or a CxHxW tensor in [0,1]."""
    if isinstance(x, Image.Image):
        return x
    if isinstance(x, torch.Tensor):
        t = x.detach().float().cpu()
        if t.dim() == 4:
            t = t[0]
        if t.dim() == 3 and t.shape[0] in (1, 3):
            t = t.permute(1, 2, 0)
        x = t.numpy()
    a = np.asarray(x)
    if a.dtype != np.uint8:
        a = (np.clip(a, 0.0, 1.0) * 255.0).round().astype(np.uint8)
    if a.ndim == 3 and a.shape[-1] == 1:
        a = a[..., 0]
    return Image.fromarray(a)


def pad(image: Image.Image, top=0, right=0, bottom=0, left=0, color=(255, 255, 255)) -> Image.Image:
    out = Image.new(image.mode, (image.width + right + left, image.height + top + bottom), color)
    out.paste(image, (left, top))
    return out


def annotate_image(image: Image.Image, text: str, font_size: int = 15) -> Image.Image:
    """The prompt wrapped above the frame on a white band (image_util.py:36-54)."""
    try:
        font = ImageFont.load_default(size=font_size)
    except TypeError:  # older Pillow: fixed-size bitmap font
        font = ImageFont.load_default()
    probe = ImageDraw.Draw(image)
    x0, y0, x1, y1 = probe.textbbox((0, 0), text, font=font)
    text_w, text_h = max(x1 - x0, 1), max(y1 - y0, 1)
    per_line = max(1, math.floor(len(text) * image.width / text_w))
    lines = textwrap.wrap(text, width=per_line) or [""]
    image = pad(image.convert("RGB"), top=(text_h + 2) * len(lines) + 3)
    ImageDraw.Draw(image).text((0, 0), "\n".join(lines), fill=(0, 0, 0), font=font)
    return image


def make_grid(images: Sequence, rows=None, cols=None) -> Image.Image:
    """Row-major grid of equally sized tiles (image_util.py:57-74)."""
    images = [to_pil(i) for i in images]
    if rows is None:
        assert cols is not None
        rows = math.ceil(len(images) / cols)
    else:
        cols = math.ceil(len(images) / rows)
    w, h = images[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, image in enumerate(images):
        if image.size != (w, h):
            image = image.resize((w, h))
        grid.paste(image, box=(i % cols * w, i // cols * h))
    return grid


def save_images_as_gif(images: Sequence[Image.Image], save_path: str, loop=0, duration=100, optimize=False) -> None:
    images[0].save(save_path, save_all=True, append_images=list(images[1:]), optimize=optimize, loop=loop, duration=duration)


def save_images_as_mp4(images: Sequence[Image.Image], save_path: str) -> bool:
    try:
        import imageio
    except ImportError:
        return False
    writer = imageio.get_writer(save_path, fps=10)
    for i in images:
        writer.append_data(np.array(i.convert("RGB")))
    writer.close()
    return True


def save_images_as_folder(images: Sequence[Image.Image], save_path: str) -> None:
    os.makedirs(save_path, exist_ok=True)
    for index, image in enumerate(images):
        image.save(os.path.join(save_path, f"{index:05d}.png"))


def save_gif_mp4_folder_type(images, save_path: str, save_gif=True) -> List[str]:
    """`x.gif` + `x.mp4` (if imageio is available) + `x/%05d.png`; returns what was written."""
    images = [to_pil(i) for i in images]
    written = []
    if save_gif:
        save_images_as_gif(images, save_path)
        written.append(save_path)
    mp4 = save_path.replace("gif", "mp4")
    if save_images_as_mp4(images, mp4):
        written.append(mp4)
    folder = save_path.replace(".gif", "")
    save_images_as_folder(images, folder)
    written.append(folder)
    return written


def numpy_seq_to_pil(images) -> List[Image.Image]:
    images = np.asarray(images)
    if images.ndim == 3:
        images = images[None, ...]
    images = (images * 255).round().astype("uint8")
    if images.shape[-1] == 1:
        return [Image.fromarray(image.squeeze(), mode="L") for image in images]
    return [Image.fromarray(image) for image in images]


def numpy_batch_seq_to_pil(images) -> List[List[Image.Image]]:
    return [numpy_seq_to_pil(sequence) for sequence in images]


def log_train_samples(train_dataloader, save_path, num_batch: int = 4):
    """Grid gif of the first input clips (image_util.py:120-137; test_fatezero.py:151)."""
    samples = []
    for idx, batch in enumerate(train_dataloader):
        if idx >= num_batch:
            break
        samples.append(batch["images"])
    x = torch.cat(samples).float().cpu().numpy()            # b c f h w
    x = np.transpose(x, (0, 2, 3, 4, 1))                    # b f h w c
    x = (x * 0.5 + 0.5).clip(0, 1)
    seqs = numpy_batch_seq_to_pil(x)
    frames = [make_grid(images, cols=int(np.ceil(np.sqrt(len(seqs))))) for images in zip(*seqs)]
    return save_gif_mp4_folder_type(frames, save_path)
