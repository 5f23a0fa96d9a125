"""fatezero_amd/clip.py (OpenAI CLIP on the native kernels) against vectors recorded from the reference's
own vendored CLIP model (oracle/gen_golden_clip.py: CLIP/clip/model.py in fp32 with the same procedural weights)."""
import json
import os

import numpy as np
import torch

from oracle.gen_golden_clip import clip_inputs, clip_weights  # TEST INFRASTRUCTURE: the generator's own input / weight recipe

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_clip_case(name, device):
    from fatezero_amd import clip
    meta = json.load(open(os.path.join(GOLD, "clip_meta.json")))[name]
    cfg = meta["config"]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    sd = clip_weights([(k, tuple(s)) for k, s in meta["state_dict_shapes"]])
    model = clip.build_model(sd)  # architecture from the shapes, exactly as for an OpenAI checkpoint
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    assert model.visual.input_resolution == cfg["image_resolution"] and model.context_length == cfg["context_length"]
    model = model.to(device)
    image, text = clip_inputs(cfg)
    fi = model.encode_image(image.to(device)).float().cpu()
    ft = model.encode_text(text.to(device)).float().cpu()
    li, lt = model(image.to(device), text.to(device))
    res = {}
    for key, got in (("image_features", fi), ("text_features", ft), ("logits_per_image", li.float().cpu())):
        want = torch.from_numpy(gold[key])
        assert got.shape == want.shape, key
        res[key] = float((got - want).abs().max() / want.abs().max())
    assert torch.equal(lt, li.t())
    return res


def check(res):
    # fp16 storage / fp32 accumulation through 12 + 12 layers against the fp32 reference
    assert res["image_features"] <= 1e-2 and res["text_features"] <= 1e-2 and res["logits_per_image"] <= 2e-2, res
