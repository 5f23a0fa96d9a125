"""Synthetic stand-ins used by bench.py / smoke / tests: no checkpoints, CLIP vocabulary or datasets exist offline.

* `WordTokenizer` -- BOS + one id per whitespace-separated word + EOS, with the `encode` / `decode` /
  `__call__(padding='max_length')` surface the controllers and `_encode_prompt` use.  Token *indices* are what the
  prompt algebra consumes (word -> position), so a word-level vocabulary exercises exactly the same code.
* `HashTextEncoder` -- deterministic pseudo-random [n, 77, dim] "text embeddings" keyed by the token ids.
* `init_like_tuned_checkpoint` -- seeded random weights of the true architecture; the temporal branches
  (`conv_temporal.up`, `attn_temporal.to_out`) are made non-zero as in a Tune-A-Video checkpoint
  (config/teaser/jeep_posche.yaml:3 uses ./ckpt/jeep_tuned_200) so that their cost is actually paid.
"""
from types import SimpleNamespace

import torch


class WordTokenizer:
    model_max_length = 77
    bos_token_id, eos_token_id = 49406, 49407

    def __init__(self):
        self.vocab, self.inv = {}, {}

    def _id(self, word):
        if word not in self.vocab:
            i = 1000 + len(self.vocab)
            self.vocab[word], self.inv[i] = i, word
        return self.vocab[word]

    def encode(self, text):
        words = [w for w in text.split(" ") if w != ""]
        return [self.bos_token_id] + [self._id(w) for w in words] + [self.eos_token_id]

    def decode(self, ids):
        out = []
        for i in ids:
            i = int(i)
            out.append("<|startoftext|>" if i == self.bos_token_id else "<|endoftext|>" if i == self.eos_token_id
                       else self.inv[i])
        return " ".join(out)

    def __call__(self, prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        rows = []
        for p in prompts:
            ids = self.encode(p)[:max_length]
            ids = ids + [self.eos_token_id] * (max_length - len(ids))
            rows.append(ids)
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.int64))


class HashTextEncoder(torch.nn.Module):
    def __init__(self, dim=768, seed=0):
        super().__init__()
        self.dim, self.seed = dim, seed
        self._anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)

    def forward(self, input_ids, attention_mask=None):
        dev = self._anchor.device
        ids = input_ids.cpu()
        out = torch.empty(ids.shape[0], ids.shape[1], self.dim)
        for b in range(ids.shape[0]):
            for s in range(ids.shape[1]):
                g = torch.Generator().manual_seed(int(ids[b, s]) * 7919 + s * 104729 + self.seed)
                out[b, s] = torch.randn(self.dim, generator=g)
        return (out.to(dev),)


@torch.no_grad()
def init_like_tuned_checkpoint(unet, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name, p in unet.named_parameters():
        if "conv_temporal.up.weight" in name or "attn_temporal.to_out.0.weight" in name:
            fan_in = p[0].numel()
            w = torch.randn(p.shape, generator=g) * (0.3 * fan_in ** -0.5)
            p.copy_(w.to(p.device, p.dtype))
    return unet
