"""from_2d_model / load_2d_state_dict (reference: unet_3d_condition.py:449-501) on a synthetic diffusers-layout folder."""
import json
import os

import pytest
import torch

from fatezero_amd.video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel

CFG_2D = {"_class_name": "UNet2DConditionModel", "_diffusers_version": "0.11.1", "sample_size": 64, "in_channels": 4,
          "out_channels": 4, "block_out_channels": [32, 64, 64, 64], "layers_per_block": 2, "attention_head_dim": 2,
          "cross_attention_dim": 64, "norm_num_groups": 8,
          "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
          "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}


def _two_d_state(model):
    """What a 2-D SD checkpoint holds: every key of the 3-D model that is not a temporal addition."""
    g = torch.Generator().manual_seed(3)
    return {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in model.state_dict().items() if "_temporal" not in k}


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_from_2d_model_roundtrip(tmp_path, fmt):
    folder = tmp_path / "unet"
    os.makedirs(folder)
    json.dump(CFG_2D, open(folder / "config.json", "w"))
    blank = UNetPseudo3DConditionModel.from_2d_model(str(folder), {"lora": 16})   # no weights file yet: construct only
    assert type(blank.down_blocks[0]).__name__ == "CrossAttnDownBlockPseudo3D"
    assert type(blank.up_blocks[0]).__name__ == "UpBlockPseudo3D"
    sd2 = _two_d_state(blank)
    if fmt == "bin":
        torch.save(sd2, folder / "diffusion_pytorch_model.bin")
    else:
        from safetensors.torch import save_file
        save_file(sd2, str(folder / "diffusion_pytorch_model.safetensors"))
    model = UNetPseudo3DConditionModel.from_2d_model(str(folder), {"lora": 16})
    sd3 = model.state_dict()
    for k, v in sd2.items():
        assert torch.equal(sd3[k], v), k
    temporal = [k for k in sd3 if "_temporal" in k]
    assert temporal and all(k not in sd2 for k in temporal)
    # un-tuned checkpoints leave the temporal LoRA `up` at zero: an exact no-op
    assert all(float(sd3[k].abs().max()) == 0.0 for k in temporal if k.endswith("conv_temporal.up.weight"))


def test_load_2d_state_dict_errors(tmp_path):
    folder = tmp_path / "unet"
    os.makedirs(folder)
    json.dump(CFG_2D, open(folder / "config.json", "w"))
    model = UNetPseudo3DConditionModel.from_2d_model(str(folder), {"lora": 16})
    sd2 = _two_d_state(model)
    with pytest.raises(KeyError):   # a 2-D key the 3-D model does not have
        model.load_2d_state_dict({**sd2, "not.a.key": torch.zeros(1)})
    k0 = next(iter(sd2))
    with pytest.raises(ValueError):  # shape mismatch
        model.load_2d_state_dict({**sd2, k0: torch.zeros(3, 3, 3)})
    with pytest.raises(KeyError):   # a non-temporal 3-D key missing from the checkpoint
        model.load_2d_state_dict({k: v for k, v in sd2.items() if k != k0})
    with pytest.raises(RuntimeError):
        UNetPseudo3DConditionModel.from_2d_model(str(tmp_path / "nowhere"), None)
