"""TEST INFRASTRUCTURE -- CPU fp32 restatement of the Stable-Diffusion VAE the reference takes from diffusers.

The reference never defines the VAE: `test_fatezero.py:96-99` loads `diffusers.AutoencoderKL` and the pipelines call
`vae.encode(x).latent_dist.sample()` (video_diffusion/pipelines/p2p_ddim_spatial_temporal.py:94) and
`vae.decode(z).sample` (video_diffusion/pipelines/stable_diffusion.py:309).  The algorithm therefore lives in a
third-party dependency that is absent here: **diffusers==0.11.1** (requirements.txt:4) -- `models/vae.py`
(Encoder, Decoder, DiagonalGaussianDistribution, AutoencoderKL), `models/unet_2d_blocks.py` (DownEncoderBlock2D,
UpDecoderBlock2D, UNetMidBlock2D), `models/resnet.py` (ResnetBlock2D, Downsample2D, Upsample2D) and
`models/attention.py` (AttentionBlock).  This file restates that published architecture with plain `torch.nn.functional`
calls on a diffusers-format state dict (same key names), so that the HIP path of
`fatezero_amd/video_diffusion/models/vae.py` can be checked against an independent implementation.

PARITY UNPINNED in the strict sense -- diffusers cannot be imported offline and the reference holds no golden vectors for the VAE -- but
not unchecked: tests/test_vae_pin.py holds this restatement against (1) the SD-1.x checkpoint's key layout written out from the published
config (248 tensors, 83 653 863 parameters: every tensor must be consumed, names and shapes must be the native model's) and (2) a second
statement of the same network from a different published source, the original CompVis latent-diffusion autoencoder
(oracle/vae_ldm_ref.py: other module structure, key names, block order, attention arithmetic), through the published key mapping:
moments and images agree to 1e-6 at the real architecture.  Only tests/ may import this module.
"""
import math

import torch
import torch.nn.functional as F


def _gn(x, sd, p, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _resnet(x, sd, p, groups):
    """ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1): norm1-silu-conv1-norm2-silu-conv2 + shortcut."""
    h = _conv(F.silu(_gn(x, sd, p + ".norm1", groups)), sd, p + ".conv1")
    h = _conv(F.silu(_gn(h, sd, p + ".norm2", groups)), sd, p + ".conv2")
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".conv_shortcut", padding=0)
    return x + h


def _attention(x, sd, p, groups):
    """AttentionBlock(channels, num_head_channels=None): one head; q and k each scaled by channels^-1/4; fp32 softmax."""
    b, c, h, w = x.shape
    hs = _gn(x, sd, p + ".group_norm", groups).view(b, c, h * w).transpose(1, 2)
    q = F.linear(hs, sd[p + ".query.weight"], sd[p + ".query.bias"])
    k = F.linear(hs, sd[p + ".key.weight"], sd[p + ".key.bias"])
    v = F.linear(hs, sd[p + ".value.weight"], sd[p + ".value.bias"])
    scale = 1.0 / math.sqrt(math.sqrt(c))
    probs = torch.softmax((q * scale) @ (k * scale).transpose(1, 2), dim=-1)
    hs = F.linear(probs @ v, sd[p + ".proj_attn.weight"], sd[p + ".proj_attn.bias"])
    return x + hs.transpose(1, 2).reshape(b, c, h, w)


def _mid(x, sd, p, groups):
    x = _resnet(x, sd, p + ".resnets.0", groups)
    x = _attention(x, sd, p + ".attentions.0", groups)
    return _resnet(x, sd, p + ".resnets.1", groups)


def encode_moments(sd, cfg, x):
    """AutoencoderKL.encode up to the posterior parameters [N, 2 C_lat, h, w] (mean | logvar)."""
    g, nb, lpb = cfg["norm_num_groups"], len(cfg["block_out_channels"]), cfg["layers_per_block"]
    h = _conv(x, sd, "encoder.conv_in")
    for i in range(nb):
        for j in range(lpb):
            h = _resnet(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", g)
        if i != nb - 1:  # Downsample2D(padding=0): F.pad (0,1,0,1) then stride-2 conv without padding
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    h = _mid(h, sd, "encoder.mid_block", g)
    h = _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", g)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def decode(sd, cfg, z):
    """AutoencoderKL.decode(z).sample."""
    g, nb, lpb = cfg["norm_num_groups"], len(cfg["block_out_channels"]), cfg["layers_per_block"]
    h = _conv(z, sd, "post_quant_conv", padding=0)
    h = _conv(h, sd, "decoder.conv_in")
    h = _mid(h, sd, "decoder.mid_block", g)
    for i in range(nb):
        for j in range(lpb + 1):
            h = _resnet(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", g)
        if i != nb - 1:  # Upsample2D: nearest 2x, then conv
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    return _conv(F.silu(_gn(h, sd, "decoder.conv_norm_out", g)), sd, "decoder.conv_out")


def posterior(moments):
    """DiagonalGaussianDistribution: (mean, std) with logvar clamped to [-30, 20]."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean, torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
