"""Deterministic, name-keyed synthetic weights (TEST INFRASTRUCTURE).

No checkpoints exist offline, so parity runs on procedurally generated weights that
depend only on (seed, parameter name, shape): the golden generator fills the *reference* model with
them and the tests fill the product / oracle model with the very same values, so only inputs and
outputs have to be stored as fixtures.  Unlike the reference initialisation, the temporal branches
(`conv_temporal.up`, `attn_temporal.to_out`) are non-zero so that they are actually exercised
(a tuned Tune-A-Video checkpoint has them non-zero).
"""
import zlib

import torch


def _scale(name: str, shape) -> float:
    if name.endswith(".bias"):
        return 0.05
    if len(shape) == 1:  # norm weights handled by caller
        return 0.1
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    base = fan_in ** -0.5
    if ".attn2.to_q" in name or ".attn2.to_k" in name:
        return 2.5 * base  # peaky, spatially varying cross-attention -> non-degenerate blend masks
    if "conv_temporal.up" in name or "attn_temporal.to_out" in name:
        return 0.3 * base
    # residual-branch outputs are damped so the net is as well conditioned as a trained one
    # (un-damped random residual stacks amplify rounding noise ~3x per block: useless for parity)
    if (".to_out.0." in name or ".proj_out." in name or ".conv2.weight" in name or ".ff.net.2." in name):
        return 0.4 * base
    return base


def procedural_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    is_norm_weight = name.endswith(".weight") and len(shape) == 1
    if is_norm_weight:
        return 1.0 + 0.1 * t
    return t * _scale(name, shape)


def procedural_state_dict(named_shapes, seed: int = 0):
    """named_shapes: iterable of (name, shape). Returns {name: fp32 tensor}."""
    return {n: procedural_tensor(n, s, seed) for n, s in named_shapes}
