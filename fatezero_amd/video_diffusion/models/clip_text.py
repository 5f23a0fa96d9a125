"""CLIP text encoder + BPE tokenizer of Stable Diffusion 1.x.

The reference takes both from transformers ([3P] `transformers==4.25.1`: `models/clip/modeling_clip.py` CLIPTextModel,
`models/clip/tokenization_clip.py` CLIPTokenizer) -- `test_fatezero.py:82-93` loads them from the `tokenizer/` and
`text_encoder/` folders of the checkpoint and `_encode_prompt` (video_diffusion/pipelines/stable_diffusion.py:180-295) calls
`tokenizer(prompt, padding="max_length", max_length=tokenizer.model_max_length, truncation=True, return_tensors="pt")` and
`text_encoder(input_ids)[0]`; the prompt-to-prompt host code needs `tokenizer.encode / decode`
(video_diffusion/prompt_attention/ptp_utils.py:144-162, seq_aligner.py:61-196).  This module keeps those surfaces and the
checkpoint formats (HF `config.json` + `pytorch_model.bin` / `model.safetensors`; `vocab.json` + `merges.txt`).

Engine: the 12 pre-LN transformer layers run token-major in fp16 -- LayerNorm (csrc/norms.hip) and every Linear with its
bias / residual epilogue (csrc/igemm.hip) are the hand-written kernels.  What stays in PyTorch: the embedding lookup, the
12-head causal attention over 77 tokens (a 77 x 77 x 64 problem per head, run once per prompt, outside every loop) and the
quick-GELU between the two MLP GEMMs.

The tokenizer restates the published CLIP byte-pair encoding (openai/CLIP `simple_tokenizer.py`, the algorithm the HF slow
tokenizer implements): lower-case + whitespace clean-up, the CLIP split pattern, byte -> unicode mapping, greedy lowest-rank
merges with the `</w>` end-of-word marker.
"""
import html
import json
import os
from functools import lru_cache
from types import SimpleNamespace
from typing import List, Optional, Union

import torch
from torch import nn

from ... import kernels as K
from .resnet import _LinearParams, _NormParams


# ------------------------------------------------------------------------------------------------------------
#                                               text encoder
# ------------------------------------------------------------------------------------------------------------
class _TextOutput(tuple):
    """(last_hidden_state, pooler_output) with the attribute access of transformers' BaseModelOutputWithPooling."""

    def __new__(cls, last_hidden_state, pooler_output):
        o = super().__new__(cls, (last_hidden_state, pooler_output))
        o.last_hidden_state, o.pooler_output = last_hidden_state, pooler_output
        return o


def _ln(norm: _NormParams, x):
    g, b = norm.packed(x.device)
    return K.layernorm(x, g, b, eps=norm.eps)


class _Attention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.heads, self.dim = heads, hidden // heads
        self.q_proj = _LinearParams(hidden, hidden)
        self.k_proj = _LinearParams(hidden, hidden)
        self.v_proj = _LinearParams(hidden, hidden)
        self.out_proj = _LinearParams(hidden, hidden)
        self._qkv = None

    def forward(self, x, residual):
        """x: LayerNorm'ed [B, L, H] fp16 -> residual + out_proj(causal attention)."""
        b, l, hdim = x.shape
        if self._qkv is None or self._qkv[0].device != x.device:
            ws = [p.weight.detach() for p in (self.q_proj, self.k_proj, self.v_proj)]
            bs = [p.bias.detach() for p in (self.q_proj, self.k_proj, self.v_proj)]
            self._qkv = (torch.cat(ws, 0).to(device=x.device, dtype=torch.float16).contiguous(),
                         torch.cat(bs, 0).to(device=x.device, dtype=torch.float16).contiguous())
        qkv = K.gemm(x, self._qkv[0], self._qkv[1]).view(b, l, 3, self.heads, self.dim)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))  # [B, heads, L, d]
        s = (q * self.dim ** -0.5) @ k.transpose(-1, -2)
        s = s + torch.full((l, l), float("-inf"), device=x.device).triu_(1)     # causal mask (modeling_clip.py)
        o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(b, l, hdim).to(torch.float16)
        return self.out_proj.apply(o, res=residual)


class _MLP(nn.Module):
    def __init__(self, hidden, inter, act):
        super().__init__()
        self.fc1 = _LinearParams(hidden, inter)
        self.fc2 = _LinearParams(inter, hidden)
        self.act = act

    def forward(self, x, residual):
        h = self.fc1.apply(x).float()
        if self.act == "quick_gelu":
            h = h * torch.sigmoid(1.702 * h)
        elif self.act == "gelu":
            h = torch.nn.functional.gelu(h)
        else:
            raise NotImplementedError(self.act)
        return self.fc2.apply(h.to(torch.float16), res=residual)


class _EncoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = _Attention(cfg.hidden_size, cfg.num_attention_heads)
        self.layer_norm1 = _NormParams(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(cfg.hidden_size, cfg.intermediate_size, cfg.hidden_act)
        self.layer_norm2 = _NormParams(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, x):
        x = self.self_attn(_ln(self.layer_norm1, x), x)
        return self.mlp(_ln(self.layer_norm2, x), x)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class _TextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = _NormParams(cfg.hidden_size, eps=cfg.layer_norm_eps)


_CFG_DEFAULTS = dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, num_hidden_layers=12, num_attention_heads=8,
                     max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2, bos_token_id=0,
                     pad_token_id=1)


class CLIPTextModel(nn.Module):
    """Drop-in for `transformers.CLIPTextModel` as the reference uses it: `text_encoder(input_ids)[0]` -> [B, 77, hidden]."""

    def __init__(self, config=None, **kw):
        super().__init__()
        c = dict(_CFG_DEFAULTS)
        src = config if isinstance(config, dict) else (vars(config) if config is not None else {})
        c.update({k: v for k, v in src.items() if k in _CFG_DEFAULTS})
        c.update({k: v for k, v in kw.items() if k in _CFG_DEFAULTS})
        self.config = SimpleNamespace(**c)
        self.text_model = _TextTransformer(self.config)

    @property
    def device(self):
        return self.text_model.embeddings.token_embedding.weight.device

    @property
    def dtype(self):
        return self.text_model.embeddings.token_embedding.weight.dtype

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **unused):
        root = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(root, "config.json")) as f:
            cfg = json.load(f)
        cfg = cfg.get("text_config", cfg) if "hidden_size" not in cfg else cfg
        model = cls(cfg)
        st = os.path.join(root, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(root, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model.eval()

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the key layout of transformers 4.x (`text_model.` prefix, an `embeddings.position_ids` buffer) and 5.x."""
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("position_ids"):
                continue
            sd[k if k.startswith("text_model.") else "text_model." + k] = v
        for m in self.modules():
            if isinstance(m, (_LinearParams, _NormParams)):
                m._packed = None
            if isinstance(m, _Attention):
                m._qkv = None
        return super().load_state_dict(sd, strict=strict)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, **unused):
        if attention_mask is not None and not bool(attention_mask.all()):
            raise NotImplementedError("padding masks: SD-1.x feeds the text encoder without one (use_attention_mask is unset)")
        tm = self.text_model
        ids = input_ids.to(self.device)
        b, l = ids.shape
        pos = torch.arange(l, device=ids.device)
        x = (tm.embeddings.token_embedding(ids) + tm.embeddings.position_embedding(pos)[None]).to(torch.float16).contiguous()
        for layer in tm.encoder.layers:
            x = layer(x)
        x = _ln(tm.final_layer_norm, x)
        # pooled = the hidden state at the end-of-text token (transformers 4.25.1: input_ids.argmax(-1), EOS has the largest id)
        if self.config.eos_token_id == 2:
            eos = ids.argmax(-1)
        else:
            eos = (ids == self.config.eos_token_id).int().argmax(-1)
        out = x.to(self.dtype) if self.dtype != torch.float16 else x
        return _TextOutput(out, out[torch.arange(b, device=ids.device), eos])


# ------------------------------------------------------------------------------------------------------------
#                                                 tokenizer
# ------------------------------------------------------------------------------------------------------------
@lru_cache()
def _bytes_to_unicode():
    """The reversible byte <-> printable-unicode table of GPT-2 / CLIP BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class _Encoding(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class CLIPTokenizer:
    """Drop-in for the slow `transformers.CLIPTokenizer` (`AutoTokenizer.from_pretrained(..., use_fast=False)`)."""

    def __init__(self, vocab_file, merges_file, bos_token="<|startoftext|>", eos_token="<|endoftext|>",
                 pad_token="<|endoftext|>", model_max_length=77, **unused):
        import regex
        with open(vocab_file, encoding="utf-8") as f:
            self.encoder = json.load(f)
        self.decoder = {v: k for k, v in self.encoder.items()}
        with open(merges_file, encoding="utf-8") as f:
            lines = f.read().strip().split("\n")[1:]  # first line: "#version: ..."
        lines = lines[: 49152 - 256 - 2 + 1]
        self.bpe_ranks = {tuple(m.split()): i for i, m in enumerate(lines)}
        self.byte_encoder = _bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.bos_token, self.eos_token, self.pad_token = bos_token, eos_token, pad_token
        self.unk_token = eos_token
        self.bos_token_id, self.eos_token_id = self.encoder[bos_token], self.encoder[eos_token]
        self.pad_token_id = self.encoder[pad_token]
        self.model_max_length = model_max_length
        self.cache = {bos_token: bos_token, eos_token: eos_token}
        self.pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **unused):
        root = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        kw = {}
        cfg = os.path.join(root, "tokenizer_config.json")
        if os.path.exists(cfg):
            with open(cfg) as f:
                tc = json.load(f)
            for key in ("bos_token", "eos_token", "pad_token"):
                v = tc.get(key)
                if isinstance(v, dict):
                    v = v.get("content")
                if isinstance(v, str):
                    kw[key] = v
            if isinstance(tc.get("model_max_length"), int) and tc["model_max_length"] < 10 ** 6:
                kw["model_max_length"] = tc["model_max_length"]
        return cls(os.path.join(root, "vocab.json"), os.path.join(root, "merges.txt"), **kw)

    def __len__(self):
        return len(self.encoder)

    @property
    def vocab_size(self):
        return len(self.encoder)

    # -- BPE ------------------------------------------------------------------------------------------------
    def _bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = set(zip(word[:-1], word[1:]))
        if not pairs:
            return token + "</w>"
        while True:
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if best not in self.bpe_ranks:
                break
            first, second = best
            new, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = set(zip(word[:-1], word[1:]))
        out = " ".join(word)
        self.cache[token] = out
        return out

    def tokenize(self, text: str) -> List[str]:
        import regex
        text = html.unescape(html.unescape(text))
        text = regex.sub(r"\s+", " ", text).strip().lower()
        toks = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            toks.extend(self._bpe(tok).split(" "))
        return toks

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self.encoder.get(tokens, self.encoder[self.unk_token])
        return [self.encoder.get(t, self.encoder[self.unk_token]) for t in tokens]

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.decoder[ids]
        return [self.decoder[int(i)] for i in ids]

    def encode(self, text: str, add_special_tokens: bool = True, **unused) -> List[int]:
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        return [self.bos_token_id] + ids + [self.eos_token_id] if add_special_tokens else ids

    def decode(self, ids, skip_special_tokens: bool = False, **unused) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        if isinstance(ids, int):
            ids = [ids]
        special = {self.bos_token_id, self.eos_token_id}
        pieces = []  # the slow HF tokenizer joins BPE runs and special tokens with single spaces
        run = []
        for i in ids:
            if int(i) in special:
                if run:
                    pieces.append(self._detok(run))
                    run = []
                if not skip_special_tokens:
                    pieces.append(self.decoder[int(i)])
            else:
                run.append(self.decoder[int(i)])
        if run:
            pieces.append(self._detok(run))
        return " ".join(pieces)

    def _detok(self, tokens: List[str]) -> str:
        text = "".join(tokens)
        data = bytearray(self.byte_decoder[c] for c in text)
        return data.decode("utf-8", errors="replace").replace("</w>", " ").strip()

    def batch_decode(self, batch, **kw):
        return [self.decode(ids, **kw) for ids in batch]

    def __call__(self, text: Union[str, List[str]], padding=False, max_length: Optional[int] = None, truncation=False,
                 return_tensors: Optional[str] = None, **unused):
        texts = [text] if isinstance(text, str) else list(text)
        max_length = self.model_max_length if max_length is None else max_length
        rows, masks = [], []
        for t in texts:
            ids = self.encode(t)
            if truncation and len(ids) > max_length:
                ids = ids[: max_length - 1] + [self.eos_token_id]
            rows.append(ids)
        if padding in ("max_length", True, "longest"):
            width = max_length if padding == "max_length" else max(len(r) for r in rows)
            masks = [[1] * len(r) + [0] * (width - len(r)) for r in rows]
            rows = [r + [self.pad_token_id] * (width - len(r)) for r in rows]
        else:
            masks = [[1] * len(r) for r in rows]
        if return_tensors == "pt":
            return _Encoding(input_ids=torch.tensor(rows, dtype=torch.long), attention_mask=torch.tensor(masks, dtype=torch.long))
        if isinstance(text, str):
            return _Encoding(input_ids=rows[0], attention_mask=masks[0])
        return _Encoding(input_ids=rows, attention_mask=masks)
