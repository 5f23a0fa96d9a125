"""Synthetic frame folders + the case list shared by oracle/gen_golden_dataset.py (records the reference's output) and
tests/test_dataset.py (checks the product against it).  No product imports here."""
import os

import numpy as np
from PIL import Image


def write_frames(folder, n=10, h=40, w=56, seed=0, ext=".png"):
    """n smooth-ish RGB frames (low-frequency pattern + noise + a moving block) as lossless PNGs; a distractor text file."""
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for i in range(n):
        img = np.stack([127 + 100 * np.sin(xx / (5 + c) + i * 0.3) * np.cos(yy / (7 - c)) for c in range(3)], -1)
        img += rng.randint(-12, 13, size=img.shape)
        y0, x0 = (3 * i) % (h - 8), (5 * i) % (w - 8)
        img[y0:y0 + 8, x0:x0 + 8] = rng.randint(0, 256, size=3)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(folder, f"{i:05d}{ext}"))
    with open(os.path.join(folder, "notes.txt"), "w") as f:
        f.write("not an image")


# (name, write_frames kwargs, dataset kwargs, item index)
CASES = [
    ("landscape_center", dict(n=10, h=40, w=56), dict(n_sample_frame=4, image_size=32), 0),
    ("portrait_offset", dict(n=9, h=60, w=36, seed=1), dict(n_sample_frame=3, sampling_rate=2, start_sample_frame=1, image_size=24,
                                                            offset={"left": 2, "right": 3, "top": 5, "bottom": 1}), 0),
    ("strided_second_clip", dict(n=12, h=32, w=32, seed=2), dict(n_sample_frame=2, stride=3, image_size=16), 2),
    ("all_frames_upscale", dict(n=5, h=20, w=28, seed=3), dict(n_sample_frame=-1, image_size=48), 0),
]
