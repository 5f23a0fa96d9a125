"""OpenAI CLIP (ViT image tower + text tower) on the native kernels -- the encoder of the reference's editing-quality metric
(reference: CLIP/frame_acc_tem_con.py:7-33 `clip.load("ViT-B/32")`, `preprocess`, `clip.tokenize`,
`model.encode_image / encode_text / model(image, text)`; arithmetic: CLIP/clip/model.py:171-372 ResidualAttentionBlock,
VisionTransformer, CLIP.encode_text / forward; CLIP/clip/clip.py:79-86 `_transform`, :195-235 `tokenize`).

Same surface as the vendored package -- `load(path) -> (model, preprocess)`, `tokenize(texts)`, `model.encode_image`,
`model.encode_text`, `model(image, text) -> (logits_per_image, logits_per_text)`, `model.logit_scale`, `model.visual.*` -- and the
same parameter names, so the state dict of an OpenAI checkpoint (`ViT-B-32.pt`, TorchScript archive or plain state dict)
loads unchanged.  (ResNet image towers -- RN50 ... -- are not built: the metric uses ViT-B/32.)

Engine: token-major fp16.  LayerNorm (csrc/norms.hip), every Linear with its bias / residual epilogue and the patch
embedding (a stride-P convolution = one GEMM over unfolded patches; csrc/igemm.hip), and the image tower's attention (the
flash kernel of the UNet, csrc/attn_flash.hip: d = 64, 50 keys) are the hand-written kernels.  What stays in PyTorch: the
embedding lookups, the causal 77-token attention of the text tower (run once per prompt pair) and the quick-GELU between the two
MLP GEMMs.  There is no CPU fallback: the kernels need libfatezero_hip.so.
"""
import gzip
import html
import os
from typing import List, Union

import torch
from torch import nn

from . import kernels as K
from .video_diffusion.models.clip_text import CLIPTokenizer, _bytes_to_unicode
from .video_diffusion.models.resnet import _LinearParams, _NormParams

__all__ = ["CLIP", "load", "tokenize", "build_model", "available_models"]


def _ln(norm: _NormParams, x):
    g, b = norm.packed(x.device)
    return K.layernorm(x, g, b, eps=norm.eps)


class _InProjAttention(nn.Module):
    """nn.MultiheadAttention's parameters (`in_proj_weight` [3W, W], `in_proj_bias`, `out_proj`), model.py:176,185-187."""

    def __init__(self, width, heads):
        super().__init__()
        self.heads, self.dim = heads, width // heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = _LinearParams(width, width)
        nn.init.normal_(self.in_proj_weight, std=width ** -0.5)
        self._packed = None

    def packed(self, device):
        if self._packed is None or self._packed[0].device != device:
            self._packed = (self.in_proj_weight.detach().to(device=device, dtype=torch.float16).contiguous(),
                            self.in_proj_bias.detach().to(device=device, dtype=torch.float16).contiguous())
        return self._packed

    def forward(self, x, residual, causal):
        """x: LayerNorm'ed [N, L, W] fp16 -> residual + out_proj(attention)."""
        n, l, w = x.shape
        wq, bq = self.packed(x.device)
        qkv = K.gemm(x, wq, bq)  # [N, L, 3W]
        if causal or self.dim not in K.SUPPORTED_HEAD_DIMS:
            q, k, v = (qkv[..., i * w:(i + 1) * w].reshape(n, l, self.heads, self.dim).permute(0, 2, 1, 3).float() for i in range(3))
            s = (q * self.dim ** -0.5) @ k.transpose(-1, -2)
            if causal:  # model.py:328-335: additive -inf above the diagonal
                s = s + torch.full((l, l), float("-inf"), device=x.device).triu_(1)
            o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(n, l, w).to(torch.float16)
        else:
            # every image is its own "clip" of one frame attending to its own tokens: the UNet's flash kernel with index_list = []
            o = torch.empty(n, l, w, dtype=torch.float16, device=x.device)
            vt = K.transpose_pad(qkv[..., 2 * w:], K.pad64(l))
            K.attn_self(qkv[..., :w], qkv[..., w:2 * w], vt, o, clip_len=1, heads=self.heads, index_list=[], mode=K.FZ_ATTN_FLASH)
        return self.out_proj.apply(o, res=residual)


class _MLP(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.c_fc = _LinearParams(width, 4 * width)
        self.c_proj = _LinearParams(4 * width, width)

    def forward(self, x, residual):
        h = self.c_fc.apply(x).float()
        h = h * torch.sigmoid(1.702 * h)  # QuickGELU, model.py:166-168
        return self.c_proj.apply(h.to(torch.float16), res=residual)


class _Block(nn.Module):
    """ResidualAttentionBlock (model.py:171-192): x + attn(ln_1(x)); x + mlp(ln_2(x))."""

    def __init__(self, width, heads):
        super().__init__()
        self.attn = _InProjAttention(width, heads)
        self.ln_1 = _NormParams(width)
        self.mlp = _MLP(width)
        self.ln_2 = _NormParams(width)

    def forward(self, x, causal):
        x = self.attn(_ln(self.ln_1, x), x, causal)
        return self.mlp(_ln(self.ln_2, x), x)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([_Block(width, heads) for _ in range(layers)])

    def forward(self, x, causal):
        for blk in self.resblocks:
            x = blk(x, causal)
        return x


class _PatchConv(nn.Module):
    def __init__(self, width, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch))
        nn.init.normal_(self.weight, std=(3 * patch * patch) ** -0.5)
        self._packed = None

    def packed(self, device):
        if self._packed is None or self._packed.device != device:
            self._packed = self.weight.detach().reshape(self.weight.shape[0], -1).to(device=device, dtype=torch.float16).contiguous()
        return self._packed


class VisionTransformer(nn.Module):
    """model.py:206-240."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.conv1 = _PatchConv(width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = _NormParams(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = _NormParams(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._proj_t = None

    def forward(self, image):
        n, c, h, w = image.shape
        p = self.patch_size
        if c != 3 or h != self.input_resolution or w != self.input_resolution:
            raise ValueError(f"expected [N, 3, {self.input_resolution}, {self.input_resolution}] images, got {tuple(image.shape)}")
        g = h // p
        dev = image.device
        # stride-P convolution without bias = one GEMM over the unfolded patches, rows ordered (n, gy, gx) like conv1(x).reshape
        patches = image.to(torch.float16).reshape(n, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(n * g * g, 3 * p * p).contiguous()
        x = K.gemm(patches, self.conv1.packed(dev), None).view(n, g * g, -1)
        cls = self.class_embedding.detach().to(device=dev, dtype=torch.float16)
        x = torch.cat([cls.expand(n, 1, -1), x], dim=1) + self.positional_embedding.detach().to(device=dev, dtype=torch.float16)
        x = _ln(self.ln_pre, x.contiguous())
        x = self.transformer(x, causal=False)
        x = _ln(self.ln_post, x[:, 0, :].contiguous())
        if self._proj_t is None or self._proj_t.device != dev:
            self._proj_t = self.proj.detach().t().to(device=dev, dtype=torch.float16).contiguous()
        return K.gemm(x, self._proj_t, None)


class CLIP(nn.Module):
    """model.py:243-372 (ViT image tower).  Features come back in fp16, like the reference's fp16 GPU model."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
                 transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ResNet image towers (RN50 ...) are not built: the FateZero metric uses ViT-B/32")
        self.context_length, self.vocab_size = context_length, vocab_size
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_width // 64, embed_dim)
        self.transformer = _Transformer(transformer_width, transformer_layers, transformer_heads)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(context_length, transformer_width))
        self.ln_final = _NormParams(transformer_width)
        self.text_projection = nn.Parameter(transformer_width ** -0.5 * torch.randn(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600369327783)  # ln(1 / 0.07)
        self._text_proj_t = None

    @property
    def dtype(self):
        return torch.float16

    @property
    def device(self):
        return self.token_embedding.weight.device

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}  # TorchScript extras
        for m in self.modules():
            if hasattr(m, "_packed"):
                m._packed = None
        self._text_proj_t = None
        self.visual._proj_t = None
        return super().load_state_dict(sd, strict=strict)

    @torch.no_grad()
    def encode_image(self, image):
        return self.visual(image.to(self.device))

    @torch.no_grad()
    def encode_text(self, text):
        text = text.to(self.device)
        x = (self.token_embedding(text) + self.positional_embedding).to(torch.float16).contiguous()
        x = self.transformer(x, causal=True)
        x = _ln(self.ln_final, x)
        eot = x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)].contiguous()  # EOT has the highest id (model.py:352-354)
        if self._text_proj_t is None or self._text_proj_t.device != x.device:
            self._text_proj_t = self.text_projection.detach().t().to(device=x.device, dtype=torch.float16).contiguous()
        return K.gemm(eot, self._text_proj_t, None)

    @torch.no_grad()
    def forward(self, image, text):
        img = self.encode_image(image).float()
        txt = self.encode_text(text).float()
        img = img / img.norm(dim=1, keepdim=True)
        txt = txt / txt.norm(dim=1, keepdim=True)
        logits_per_image = self.logit_scale.exp().float() * img @ txt.t()
        return logits_per_image, logits_per_image.t()


def build_model(state_dict: dict) -> CLIP:
    """Architecture from the shapes in an OpenAI state dict (model.py:399-436), weights loaded."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("ResNet image towers (RN50 ...) are not built: the FateZero metric uses ViT-B/32")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len({k.split(".")[3] for k in state_dict if k.startswith("visual.transformer.resblocks.")})
    patch = state_dict["visual.conv1.weight"].shape[-1]
    grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    width = state_dict["ln_final.weight"].shape[0]
    layers = len({k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks.")})
    model = CLIP(embed_dim, patch * grid, vision_layers, vision_width, patch, context_length, vocab_size, width, width // 64, layers)
    model.load_state_dict(state_dict)
    return model.eval()


# ------------------------------------------------------------------------------------------------------------------------
#                                          preprocessing, tokenizer, load
# ------------------------------------------------------------------------------------------------------------------------
_MEAN = (0.48145466, 0.4578275, 0.40821073)
_STD = (0.26862954, 0.26130258, 0.27577711)


class _Transform:
    """clip.py:79-86: Resize(n_px, bicubic) of the shorter side, CenterCrop(n_px), RGB, ToTensor, Normalize -- on PIL images, with
    PIL's own bicubic filter (what torchvision's Resize does for PIL input)."""

    def __init__(self, n_px):
        self.n_px = n_px

    def __call__(self, image):
        import numpy as np
        from PIL import Image
        w, h = image.size
        n = self.n_px
        if (w <= h and w != n) or (h < w and h != n):  # torchvision.transforms.functional.resize with an int size
            if w <= h:
                image = image.resize((n, int(n * h / w)), Image.BICUBIC)
            else:
                image = image.resize((int(n * w / h), n), Image.BICUBIC)
        w, h = image.size
        left, top = int(round((w - n) / 2.0)), int(round((h - n) / 2.0))
        image = image.crop((left, top, left + n, top + n)).convert("RGB")
        x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        mean, std = torch.tensor(_MEAN)[:, None, None], torch.tensor(_STD)[:, None, None]
        return (x - mean) / std


_tokenizer_cache = {}


def _openai_tokenizer(bpe_path: str) -> CLIPTokenizer:
    """The vocabulary of clip/simple_tokenizer.py:62-75 from its merges file (`bpe_simple_vocab_16e6.txt.gz`): 256 byte symbols,
    the same with `</w>`, one entry per merge (lines 1 .. 49152-256-2), then the two special tokens."""
    if bpe_path not in _tokenizer_cache:
        import json
        import tempfile
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as f:
            merges = f.read().split("\n")[1:49152 - 256 - 2 + 1]
        merges = [m for m in merges if m]
        base = list(_bytes_to_unicode().values())
        vocab = base + [b + "</w>" for b in base] + ["".join(m.split()) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        with tempfile.TemporaryDirectory() as td:
            json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(td, "vocab.json"), "w"))
            with open(os.path.join(td, "merges.txt"), "w", encoding="utf-8") as f:
                f.write("#version: 0.2\n" + "\n".join(merges) + "\n")
            _tokenizer_cache[bpe_path] = CLIPTokenizer(os.path.join(td, "vocab.json"), os.path.join(td, "merges.txt"))
    return _tokenizer_cache[bpe_path]


def _find_bpe(bpe_path=None):
    cands = [bpe_path, os.environ.get("FZ_CLIP_BPE"), os.path.join(os.path.dirname(__file__), "bpe_simple_vocab_16e6.txt.gz"),
             os.path.join("CLIP", "clip", "bpe_simple_vocab_16e6.txt.gz")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError("CLIP BPE merges file not found: pass bpe_path=, set FZ_CLIP_BPE, or run from a FateZero checkout "
                            "(CLIP/clip/bpe_simple_vocab_16e6.txt.gz)")


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False, bpe_path: str = None) -> torch.Tensor:
    """clip.py:195-235: [SOT] + bpe(text) + [EOT], zero-padded to `context_length`; too long raises unless `truncate`."""
    if isinstance(texts, str):
        texts = [texts]
    tok = _openai_tokenizer(_find_bpe(bpe_path))
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = tok.encode(html.unescape(html.unescape(t)))  # (the tokenizer lower-cases and cleans whitespace itself)
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = tok.eos_token_id
        out[i, :len(ids)] = torch.tensor(ids)
    return out


def available_models() -> List[str]:
    return ["ViT-B/32", "ViT-B/16", "ViT-L/14", "ViT-L/14@336px"]


def load(name: str, device: Union[str, torch.device] = "cuda", jit: bool = False, download_root: str = None):
    """`name`: path of an OpenAI checkpoint file (TorchScript archive or state dict), or a model name resolved to
    `<download_root or ~/.cache/clip>/<file>` as the reference's downloader leaves it (clip.py:94-130; nothing is downloaded
    here).  Returns (model on `device`, preprocess)."""
    if jit:
        raise NotImplementedError("jit=True: this build runs the native kernels, not the TorchScript graph")
    path = name
    if not os.path.isfile(path):
        fname = {"ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt", "ViT-L/14@336px": "ViT-L-14-336px.pt"}.get(name)
        if fname is None:
            raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
        path = os.path.join(download_root or os.path.expanduser("~/.cache/clip"), fname)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path}: no network here -- place the OpenAI checkpoint there or pass its path")
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()
    model = build_model(sd).to(device)
    return model, _Transform(model.visual.input_resolution)
