"""Golden vectors for fatezero_amd/clip.py from the reference's OWN vendored CLIP model (TEST INFRASTRUCTURE; runs only in the
authoring container, where /root/reference exists).

    python oracle/gen_golden_clip.py

CLIP/clip/model.py is imported by file path (the `clip` package __init__ pulls torchvision, which is absent here; model.py itself
needs only torch + numpy), instantiated at two sizes -- a tiny ViT for the CPU-emulation test and the real ViT-B/32 dimensions for
the MI355X test -- filled with oracle/weights.py's name-keyed procedural weights (the tests regenerate the very same weights, so
only inputs and outputs are stored), and run in fp32 on CPU: `encode_image`, `encode_text`, `forward` (CLIP/clip/model.py:340-372).
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.weights import procedural_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {
    "clip_tiny": dict(embed_dim=64, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8, context_length=77,
                      vocab_size=600, transformer_width=128, transformer_heads=2, transformer_layers=2),
    "clip_vitb32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32, context_length=77,
                        vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12),
}


def clip_inputs(cfg, n_img=4, n_txt=2, seed=7):
    """Shared with the tests (tests/clip_cases.py re-creates the same tensors): images in the normalised range, token rows
    [SOT] words [EOT] 0 ... with EOT = the largest id (what `text.argmax(-1)` relies on, model.py:352-354)."""
    g = torch.Generator().manual_seed(seed)
    r, v = cfg["image_resolution"], cfg["vocab_size"]
    image = torch.randn(n_img, 3, r, r, generator=g)
    text = torch.zeros(n_txt, cfg["context_length"], dtype=torch.long)
    for i in range(n_txt):
        n = 5 + 3 * i
        text[i, 0] = v - 2
        text[i, 1:1 + n] = torch.randint(1, v - 2, (n,), generator=g)
        text[i, 1 + n] = v - 1
    return image, text


def clip_weights(shapes, seed=0):
    sd = procedural_state_dict(shapes, seed)
    sd["logit_scale"] = torch.tensor(4.6052)  # ln 100: the trained models sit at the clamp
    return sd


def main():
    spec = importlib.util.spec_from_file_location("ref_clip_model", "/root/reference/CLIP/clip/model.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    meta = {}
    for name, cfg in CASES.items():
        torch.manual_seed(0)
        model = ref.CLIP(**cfg).eval().float()
        shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict(clip_weights(shapes))
        image, text = clip_inputs(cfg)
        with torch.no_grad():
            fi = model.encode_image(image)
            ft = model.encode_text(text)
            li, lt = model(image, text)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), image_features=fi.numpy(), text_features=ft.numpy(),
                            logits_per_image=li.numpy())
        meta[name] = {"config": cfg, "state_dict_shapes": [[k, list(s)] for k, s in shapes]}
        print(name, "image", tuple(fi.shape), float(fi.abs().max()), "text", tuple(ft.shape), float(ft.abs().max()), "logits", li.tolist()[0])
    json.dump(meta, open(os.path.join(GOLD, "clip_meta.json"), "w"))


if __name__ == "__main__":
    main()
