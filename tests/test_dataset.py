"""Dataset front-end against tensors recorded from the unmodified reference
(oracle/gen_golden_dataset.py -> tests/golden/dataset_frontend.npz) on identical procedurally generated PNG frames."""
import os

import numpy as np
import pytest
import torch

import dataset_cases as DC
from fatezero_amd.video_diffusion.data import transform as T
from fatezero_amd.video_diffusion.data.dataset import ImageSequenceDataset

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_frontend.npz"))
IDS = torch.zeros(1, 77, dtype=torch.long)


@pytest.mark.parametrize("case", DC.CASES, ids=[c[0] for c in DC.CASES])
def test_matches_reference_recording(tmp_path, case):
    name, frames_kw, ds_kw, index = case
    folder = str(tmp_path / name)
    DC.write_frames(folder, **frames_kw)
    ds = ImageSequenceDataset(path=folder, prompt_ids=IDS, prompt="a clip", **ds_kw)
    assert len(ds) == int(GOLD[name + "__len"][0])
    item = ds[index]
    assert item["prompt_ids"] is IDS
    got, want = item["images"].numpy(), GOLD[name]
    assert got.shape == want.shape and got.dtype == np.float32
    # same torch build on both sides: the antialiased bilinear resize is bit-reproducible
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert got.min() >= -1.0 and got.max() <= 1.0


def test_only_image_files_are_listed_and_short_folders_are_refused(tmp_path):
    folder = str(tmp_path / "f")
    DC.write_frames(folder, n=3, h=16, w=16)
    assert ImageSequenceDataset.get_image_list(folder) == ["00000.png", "00001.png", "00002.png"]
    with pytest.raises(ValueError):
        ImageSequenceDataset(path=folder, prompt_ids=IDS, prompt="x", n_sample_frame=4)
    with pytest.raises(ValueError):
        ImageSequenceDataset(path=folder, prompt_ids=IDS, prompt="x", n_sample_frame=2, crop="middle")
    ds = ImageSequenceDataset(path=folder, prompt_ids=IDS, prompt="x", n_sample_frame=2, sampling_rate=2, image_size=8)
    assert list(ds.get_frame_indices(0)) == [0, 2] and ds.sequence_length == 3 and len(ds) == 1


def test_transform_primitives():
    x = torch.arange(2 * 3 * 10 * 14, dtype=torch.float32).view(2, 3, 10, 14)
    assert T.offset_crop(x, left=1, right=2, top=3, bottom=4).shape == (2, 3, 3, 11)
    assert T.offset_crop(x, left=100, right=100, top=100, bottom=100).shape == (2, 3, 1, 1)  # margins are clipped
    assert T.short_size_scale(x, 5).shape == (2, 3, 5, 7)
    assert T.short_size_scale(x.transpose(2, 3), 5).shape == (2, 3, 7, 5)
    c = T.center_crop(x, 4, 6)
    assert c.shape == (2, 3, 4, 6) and torch.equal(c, x[:, :, 3:7, 4:10])
    assert T.random_crop(x, 4, 6).shape == (2, 3, 4, 6)


def test_class_images_item_format(tmp_path):
    folder, cls = str(tmp_path / "f"), str(tmp_path / "cls")
    DC.write_frames(folder, n=4, h=16, w=16)
    DC.write_frames(cls, n=6, h=16, w=16, seed=9)
    os.remove(os.path.join(cls, "notes.txt"))
    cids = torch.ones(1, 77, dtype=torch.long)
    ds = ImageSequenceDataset(path=folder, prompt_ids=IDS, prompt="x", n_sample_frame=2, image_size=16, class_data_root=cls,
                              class_prompt_ids=cids)
    assert len(ds) == 6
    item = ds[5]
    assert item["class_images"].shape == (3, 2, 16, 16) and item["class_prompt_ids"] is cids
