"""Editing-quality metrics of the reference's evaluation script (CLIP/frame_acc_tem_con.py:11-57):

* frame accuracy     -- share of edited frames whose CLIP image embedding is closer (softmax over the two logits) to the target
                        prompt than to the source prompt;
* temporal consistency -- mean cosine similarity of the CLIP image embeddings of consecutive frames.

The arithmetic lives here; the encoder is a parameter: anything with `encode_image(list of PIL) -> [F, D]`,
`encode_text(list of str) -> [P, D]` and a `logit_scale`.  `NativeClipEncoder` is the reference's own choice -- OpenAI CLIP
ViT-B/32 from its `.pt` checkpoint -- on the hand-written kernels (fatezero_amd/clip.py); `TransformersClipEncoder` wraps a local
transformers CLIPModel folder.  Neither checkpoint is part of this repository (no network here).

    python -m fatezero_amd.metrics --results ./baselines_results/ours --prompts CLIP/bench_clean_prompt.yaml --clip ViT-B-32.pt
prints what CLIP/frame_acc_tem_con.py:62-94 prints (per-folder rate / consistency, then the dataset averages)."""
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
        self.logit_scale = float(self.model.logit_scale.detach().exp())

    def encode_image(self, images: List) -> torch.Tensor:
        px = self.processor(images=images, return_tensors="pt")["pixel_values"].to(self.device)
        return self.model.get_image_features(pixel_values=px)

    def encode_text(self, texts: List[str]) -> torch.Tensor:
        tok = self.processor(text=texts, return_tensors="pt", padding=True)
        return self.model.get_text_features(input_ids=tok["input_ids"].to(self.device),
                                            attention_mask=tok["attention_mask"].to(self.device))


class NativeClipEncoder:
    """OpenAI CLIP on the native kernels (fatezero_amd/clip.py): `clip.load(path)` + `preprocess` + `clip.tokenize`, exactly the
    calls of CLIP/frame_acc_tem_con.py:8,19-25."""

    def __init__(self, checkpoint: str, device: str = "cuda", bpe_path: str = None):
        from . import clip
        self._clip = clip
        self.model, self.preprocess = clip.load(checkpoint, device=device)
        self.device, self.bpe_path = device, bpe_path
        self.logit_scale = float(self.model.logit_scale.detach().exp())

    def encode_image(self, images: List) -> torch.Tensor:
        return self.model.encode_image(torch.stack([self.preprocess(i) for i in images]).to(self.device))

    def encode_text(self, texts: List[str]) -> torch.Tensor:
        return self.model.encode_text(self._clip.tokenize(texts, bpe_path=self.bpe_path).to(self.device))


def main(argv=None):
    import argparse
    import os
    from .config_driver import load_config
    ap = argparse.ArgumentParser(description="frame accuracy / temporal consistency of edited clips (CLIP/frame_acc_tem_con.py)")
    ap.add_argument("--results", default="./baselines_results/ours", help="folder of result folders, one per entry of --prompts")
    ap.add_argument("--prompts", default="CLIP/bench_clean_prompt.yaml", help="YAML: <folder name>: {source: ..., target: ...}")
    ap.add_argument("--clip", default="ViT-B/32", help="OpenAI CLIP checkpoint file (or model name under ~/.cache/clip)")
    ap.add_argument("--bpe", default=None, help="bpe_simple_vocab_16e6.txt.gz (default: FZ_CLIP_BPE or CLIP/clip/ in the working directory)")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    prompts = load_config(args.prompts)
    enc = NativeClipEncoder(args.clip, args.device, args.bpe)
    folders = sorted(glob(f"{args.results}/*"))
    rates, cons = [], []
    for folder in folders:
        p = prompts[os.path.basename(folder)]
        rate, con = folder_success(folder, p["source"], p["target"], enc)
        print(folder)
        print(f"folder_success_rate {rate}")
        print(f"folder_temporal_consistency {con}")
        rates.append(rate)
        cons.append(con)
    print("folder_success_rate list :")
    print(rates)
    print("folder_temporal_consistency list :")
    print(cons)
    n = max(len(rates), 1)
    print(f"dataset_average_rate {sum(rates) / n}")
    print(f"dataset_average_tempconst {sum(cons) / n}")
    return {"dataset_average_rate": sum(rates) / n, "dataset_average_tempconst": sum(cons) / n}


if __name__ == "__main__":
    main()
