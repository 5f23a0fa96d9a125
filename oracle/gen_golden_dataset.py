#!/usr/bin/env python3
"""Record tests/golden/dataset_frontend.npz by running the UNMODIFIED reference dataset front-end
(/root/reference/video_diffusion/data/dataset.py + transform.py) on synthetic PNG frames.

TEST INFRASTRUCTURE; runs only in the authoring container.  The frames are generated procedurally (tests/dataset_cases.py
regenerates the identical PNGs on the GPU box), so the fixture holds only the reference's OUTPUT tensors.

    python oracle/gen_golden_dataset.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "refshim"))
from stubs import install  # noqa: E402

install()
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dataset_cases as DC  # noqa: E402  (frame generator + the case list; imports nothing of the product)

sys.path.insert(0, REF)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]
import torch  # noqa: E402
import video_diffusion as _vd  # noqa: E402

assert list(_vd.__path__)[0].startswith(REF), "golden vectors must come from the unmodified reference"
from video_diffusion.data.dataset import ImageSequenceDataset  # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, frames_kw, ds_kw, index in DC.CASES:
            folder = os.path.join(tmp, name)
            DC.write_frames(folder, **frames_kw)
            ds = ImageSequenceDataset(path=folder, prompt_ids=torch.zeros(1, 77, dtype=torch.long), prompt="a clip", **ds_kw)
            item = ds[index]
            out[name] = item["images"].numpy().astype(np.float32)
            out[name + "__len"] = np.array([len(ds)])
            print(name, out[name].shape, len(ds))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dataset_frontend.npz"), **out)


if __name__ == "__main__":
    main()
