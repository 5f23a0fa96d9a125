"""Frame accuracy / temporal consistency (CLIP/frame_acc_tem_con.py) with a stand-in encoder: the arithmetic is what is tested."""
import os

import numpy as np
import torch
from PIL import Image

from fatezero_amd import metrics


class FakeEncoder:
    """Embeds an image as (mean red, mean green, mean blue, 1) and a prompt by which colour word it names."""
    logit_scale = 100.0

    def encode_image(self, images):
        return torch.tensor([[*np.asarray(i.convert("RGB"), dtype=np.float32).reshape(-1, 3).mean(0) / 255.0, 1.0] for i in images])

    def encode_text(self, texts):
        table = {"red": [1.0, 0, 0, 1], "green": [0, 1.0, 0, 1], "blue": [0, 0, 1.0, 1]}
        return torch.tensor([next(v for k, v in table.items() if k in t) for t in texts])


def _frames(folder, colours, size=(8, 12)):
    os.makedirs(folder)
    for i, c in enumerate(colours):
        Image.fromarray(np.full((size[1], size[0], 3), c, dtype=np.uint8)).save(os.path.join(folder, f"{i:05d}.png"))


def test_frame_metrics_arithmetic():
    img = torch.tensor([[1.0, 0.0], [0.0, 2.0], [3.0, 3.0]])
    txt = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    acc, con = metrics.frame_metrics(img, txt)
    assert abs(acc - 2 / 3) < 1e-6                       # frames 1 (target) and 2 (tie counts as success, >=)
    assert abs(con - (0.0 + 2 ** -0.5) / 2) < 1e-6       # cos(f0,f1) = 0, cos(f1,f2) = 1/sqrt 2
    assert np.isnan(metrics.frame_metrics(img[:1], txt)[1])


def test_folder_and_dataset(tmp_path):
    _frames(str(tmp_path / "car_red"), [(250, 10, 10), (240, 30, 20), (10, 10, 250)])
    _frames(str(tmp_path / "car_green"), [(10, 250, 10), (20, 240, 30)])
    enc = FakeEncoder()
    acc, con = metrics.folder_success(str(tmp_path / "car_red"), "a blue car", "a red car", enc)
    assert abs(acc - 2 / 3) < 1e-6 and 0.0 < con < 1.0
    out = metrics.dataset_metrics([str(tmp_path / "car_red"), str(tmp_path / "car_green")],
                                  {"car_red": {"source": "a blue car", "target": "a red car"},
                                   "car_green": {"source": "a red car", "target": "a green car"}}, enc)
    assert abs(out["dataset_average_rate"] - (2 / 3 + 1.0) / 2) < 1e-6


def test_portrait_frames_keep_the_bottom_square(tmp_path):
    arr = np.zeros((20, 10, 3), dtype=np.uint8)
    arr[10:] = 255
    Image.fromarray(arr).save(str(tmp_path / "p.png"))
    img = metrics.crop_read_image_path(str(tmp_path / "p.png"))
    assert img.size == (10, 10) and np.asarray(img).min() == 255
