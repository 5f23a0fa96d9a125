"""Editing-quality metrics of the reference's evaluation script (CLIP/frame_acc_tem_con.py:11-57):

* frame accuracy     -- share of edited frames whose CLIP image embedding is closer (softmax over the two logits) to the target
                        prompt than to the source prompt;
* temporal consistency -- mean cosine similarity of the CLIP image embeddings of consecutive frames.

The arithmetic lives here; the encoder is a parameter: anything with `encode_image(list of PIL) -> [F, D]`,
`encode_text(list of str) -> [P, D]` and a `logit_scale` (the reference uses OpenAI CLIP ViT-B/32, whose weights are not part
of this repository: `TransformersClipEncoder` wraps a local transformers CLIPModel folder when one is available)."""
from glob import glob
from typing import Dict, List, Sequence, Tuple

import torch


def crop_read_image_path(image_path: str):
    """Portrait frames keep their bottom square (frame_acc_tem_con.py:11-16)."""
    from PIL import Image
    img = Image.open(image_path)
    w, h = img.size
    return img.crop((0, h - w, w, h)) if h > w else img


def frame_metrics(image_features: torch.Tensor, text_features: torch.Tensor, logit_scale: float = 100.0) -> Tuple[float, float]:
    """image_features [F, D], text_features [2, D] = (source, target).  Returns (frame accuracy, temporal consistency)."""
    img = image_features.float()
    txt = text_features.float()
    img_n = img / img.norm(dim=1, keepdim=True)
    txt_n = txt / txt.norm(dim=1, keepdim=True)
    probs = (logit_scale * img_n @ txt_n.t()).softmax(dim=-1)           # CLIP.forward: logits_per_image
    accuracy = float((probs[:, 1] >= probs[:, 0]).float().mean())
    if img.shape[0] < 2:
        return accuracy, float("nan")
    consistency = float((img_n[:-1] * img_n[1:]).sum(dim=1).mean())
    return accuracy, consistency


def folder_success(folder: str, source_prompt: str, target_prompt: str, encoder) -> Tuple[float, float]:
    """All `*png` frames of one result folder (frame_acc_tem_con.py:36-57)."""
    files = sorted(glob(folder + "/*png"))
    if not files:
        raise FileNotFoundError(f"no png frames in {folder}")
    with torch.no_grad():
        img = encoder.encode_image([crop_read_image_path(f) for f in files])
        txt = encoder.encode_text([source_prompt, target_prompt])
    return frame_metrics(img, txt, getattr(encoder, "logit_scale", 100.0))


def dataset_metrics(folders: Sequence[str], prompts: Dict[str, Dict[str, str]], encoder) -> Dict[str, float]:
    """Average over result folders; `prompts[basename] = {"source": ..., "target": ...}` (CLIP/bench_clean_prompt.yaml)."""
    import os
    acc, con = [], []
    for folder in folders:
        p = prompts[os.path.basename(folder)]
        a, c = folder_success(folder, p["source"], p["target"], encoder)
        acc.append(a)
        con.append(c)
    n = max(len(acc), 1)
    return {"dataset_average_rate": sum(acc) / n, "dataset_average_tempconst": sum(con) / n}


class TransformersClipEncoder:
    """CLIP image / text towers from a local transformers checkpoint folder (e.g. openai/clip-vit-base-patch32)."""

    def __init__(self, path: str, device: str = "cpu"):
        from transformers import CLIPModel, CLIPProcessor
        self.model = CLIPModel.from_pretrained(path).eval().to(device)
        self.processor = CLIPProcessor.from_pretrained(path)
        self.device = device
        self.logit_scale = float(self.model.logit_scale.exp())

    def encode_image(self, images: List) -> torch.Tensor:
        px = self.processor(images=images, return_tensors="pt")["pixel_values"].to(self.device)
        return self.model.get_image_features(pixel_values=px)

    def encode_text(self, texts: List[str]) -> torch.Tensor:
        tok = self.processor(text=texts, return_tensors="pt", padding=True)
        return self.model.get_text_features(input_ids=tok["input_ids"].to(self.device),
                                            attention_mask=tok["attention_mask"].to(self.device))
