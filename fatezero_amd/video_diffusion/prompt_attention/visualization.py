"""Cross-attention heat-map strips (reference: video_diffusion/prompt_attention/visualization.py:14-72).

`aggregate_attention` averages the stored running-sum cross maps of one resolution over layers and heads;
`show_cross_attention` renders, per frame, one 256x256 grey heat map per prompt token with the token text underneath,
and (with `save_path`) writes the strips as gif / PNG folder.  This is post-loop work on a few hundred KB: plain torch /
PIL on the host, never on the hot path (the maps come out of the HBM arena's fp32 running sums).  Text is drawn with
PIL's built-in font (the reference uses cv2.putText, ptp_utils.py:32-44; cv2 is not available here)."""
import datetime
import os
from typing import List

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageFont


def aggregate_attention(prompts, attention_store, res: int, from_where: List[str], is_cross: bool, select: int, to_cpu=True):
    """Mean over (layers of `from_where` with res*res query tokens) x heads of the step-averaged maps
    -> [frames, res, res, tokens] (visualization.py:14-32)."""
    out = []
    maps = attention_store.get_average_attention()
    num_pixels = res ** 2
    kind = "cross" if is_cross else "self"
    for location in from_where:
        for item in maps[f"{location}_{kind}"]:
            item = item.float()
            if item.dim() == 3 and item.shape[1] == num_pixels:
                out.append(item.reshape(len(prompts), -1, res, res, item.shape[-1])[select])
            elif item.dim() == 4 and item.shape[2] == num_pixels:
                t = item.shape[0]
                out.append(item.reshape(len(prompts), t, -1, res, res, item.shape[-1])[select])
    if not out:
        raise ValueError(f"no stored {kind} map with {res}x{res} query tokens under {from_where}")
    out = torch.cat(out, dim=-4)
    out = out.sum(-4) / out.shape[-4]
    return out.cpu() if to_cpu else out


def text_under_image(image: np.ndarray, text: str, text_color=(0, 0, 0)) -> np.ndarray:
    """White band of 20 % of the height under the image with the text centred in it (ptp_utils.py:32-44)."""
    h, w, c = image.shape
    offset = int(h * 0.2)
    canvas = Image.fromarray(np.concatenate([image, np.full((offset, w, c), 255, dtype=np.uint8)], axis=0))
    draw = ImageDraw.Draw(canvas)
    try:
        font = ImageFont.load_default(size=max(10, offset // 2))
    except TypeError:
        font = ImageFont.load_default()
    x0, y0, x1, y1 = draw.textbbox((0, 0), text, font=font)
    draw.text(((w - (x1 - x0)) // 2, h + (offset - (y1 - y0)) // 2 - y0), text, fill=tuple(text_color), font=font)
    return np.array(canvas)


def view_images(images, num_rows=1, offset_ratio=0.02, save_path=None):
    """One white-separated contact sheet of equally sized images; saved as <save_path>/<time>.png when asked
    (ptp_utils.py:47-79).  Returns the PIL image."""
    if isinstance(images, np.ndarray) and images.ndim == 4:
        images = list(images)
    elif not isinstance(images, list):
        images = [images]
    num_empty = len(images) % num_rows
    images = [np.asarray(i).astype(np.uint8) for i in images] + [np.full(images[0].shape, 255, np.uint8)] * num_empty
    h, w, c = images[0].shape
    off = int(h * offset_ratio)
    cols = len(images) // num_rows
    sheet = np.full((h * num_rows + off * (num_rows - 1), w * cols + off * (cols - 1), 3), 255, dtype=np.uint8)
    for i in range(num_rows):
        for j in range(cols):
            sheet[i * (h + off): i * (h + off) + h, j * (w + off): j * (w + off) + w] = images[i * cols + j]
    pil = Image.fromarray(sheet)
    if save_path is not None:
        os.makedirs(save_path, exist_ok=True)
        pil.save(os.path.join(save_path, datetime.datetime.now().strftime("%Y-%m-%dT%H-%M-%S-%f") + ".png"))
    return pil


class LazyStrips(list):
    """The list `show_cross_attention` returns, rendered on first use.  The aggregation over layers / heads runs right away on
    the device (a few small tensors, detached from the HBM arena, so the arena can be recycled); the PIL work -- one 256x256
    tile per token per frame -- and the device-to-host copy only happen when somebody looks at the list (the sample logger
    does, a latent-space job never does: nothing of it sits on the editing loop's critical path)."""

    def __init__(self, render):
        super().__init__()
        self._render = render

    def _fill(self):
        if self._render is not None:
            render, self._render = self._render, None
            super().extend(render())

    def __len__(self):
        self._fill()
        return super().__len__()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)

    def __bool__(self):
        return len(self) > 0


def _decode_one(tokenizer, tok: int) -> str:
    try:
        return tokenizer.decode(int(tok))            # HF tokenizers (the reference's call, visualization.py:64)
    except (TypeError, KeyError, AttributeError):
        return tokenizer.decode([int(tok)])          # the light tokenizers of this repo take a list of ids


def show_cross_attention(tokenizer, prompts, attention_store, res: int, from_where: List[str], select: int = 0,
                         save_path=None):
    """-> list (one entry per frame) of uint8 arrays [256 * 1.2, 256 * n_tokens, 3] (visualization.py:35-72); with
    `save_path` the per-frame contact sheets and the gif / PNG folder of the strips are written immediately."""
    from ..common.image_util import save_gif_mp4_folder_type
    if isinstance(prompts, str):
        prompts = [prompts]
    tokens = tokenizer.encode(prompts[select])
    maps_dev = aggregate_attention(prompts, attention_store, res, from_where, True, select, to_cpu=False)

    def render():
        maps = maps_dev.cpu()
        if maps.dim() == 3:
            maps = maps[None]
        strips = []
        for j in range(maps.shape[0]):
            tiles = []
            for i in range(len(tokens)):
                m = maps[j, :, :, i]
                m = 255 * m / m.max().clamp_min(1e-20)
                img = m.unsqueeze(-1).expand(*m.shape, 3).numpy().astype(np.uint8)
                img = np.array(Image.fromarray(img).resize((256, 256)))
                tiles.append(text_under_image(img, _decode_one(tokenizer, tokens[i])))
            if save_path is not None:
                view_images(np.stack(tiles, axis=0), save_path=save_path)
            strips.append(np.concatenate(tiles, axis=1))
        if save_path is not None:
            now = datetime.datetime.now().strftime("%Y-%m-%dT%H-%M-%S")
            save_gif_mp4_folder_type(strips, f"{save_path}/{now}.gif")
        return strips

    out = LazyStrips(render)
    if save_path is not None:
        out._fill()
    return out
