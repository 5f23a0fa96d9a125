"""Shared test helpers (test infrastructure)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class ReplayTokenizer:
    """Replays the CLIP-BPE encode/decode results recorded by oracle/gen_golden.py (the BPE vocabulary
    lives only under /root/reference and does not travel to the GPU box)."""

    def __init__(self):
        t = json.load(open(os.path.join(GOLD, "tokenizer_replay.json")))
        self.enc, self.dec = t["encode"], t["decode"]

    def encode(self, text):
        return list(self.enc[text])

    def decode(self, ids):
        return self.dec[",".join(str(int(i)) for i in ids)]


def load_json(name):
    return json.load(open(os.path.join(GOLD, name)))


def load_npz(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name)).items()}


def tensor_digest(t: torch.Tensor):
    """Must match oracle/gen_golden.py:tensor_digest."""
    t = t.detach().double().flatten().cpu()
    n = t.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37)
    idx = torch.linspace(0, n - 1, steps=min(n, 16)).long()
    return {"n": n, "sum": float(t.sum()), "wsum": float((t * w).sum()), "abs": float(t.abs().sum()),
            "sample": [float(x) for x in t[idx]]}


def assert_digest_close(d_test, d_gold, rtol=1e-4, atol=1e-5, what=""):
    assert d_test["n"] == d_gold["n"], what
    scale = max(d_gold["abs"], 1e-12)
    for k in ("sum", "wsum", "abs"):
        assert abs(d_test[k] - d_gold[k]) <= rtol * scale + atol, (what, k, d_test[k], d_gold[k])
    a, b = np.array(d_test["sample"]), np.array(d_gold["sample"])
    assert np.allclose(a, b, rtol=rtol * 10, atol=atol + rtol * float(np.abs(b).max())), (what, a, b)


def unpack_bits(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(bool)
