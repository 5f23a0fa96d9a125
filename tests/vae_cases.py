"""Shared VAE parity cases: the native AutoencoderKL (fatezero_amd/video_diffusion/models/vae.py) against the CPU fp32
restatement of diffusers' VAE (oracle/vae_oracle.py) on the same seeded state dict."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd.video_diffusion.models.vae import AutoencoderKL  # noqa: E402
from oracle import vae_oracle  # noqa: E402

TINY = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 2, up_block_types=["UpDecoderBlock2D"] * 2,
            block_out_channels=[32, 64], layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=8, sample_size=16)
SD = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4, up_block_types=["UpDecoderBlock2D"] * 4,
          block_out_channels=[128, 256, 512, 512], layers_per_block=2, act_fn="silu", latent_channels=4, norm_num_groups=32,
          sample_size=512)


def seeded_vae(cfg, seed=0):
    torch.manual_seed(seed)
    vae = AutoencoderKL.from_config(cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in vae.state_dict().items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight") or k.endswith("norm_out.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5
        sd[k] = sd[k].half().float()  # both sides see fp16-representable weights
    vae.load_state_dict(sd)
    return vae.eval(), sd


def case_vae_roundtrip(device, cfg, n=2, hw=16, seed=0, tol_enc=2e-2, tol_dec=3e-2):
    vae, sd = seeded_vae(cfg, seed)
    vae = vae.to(device).half()
    g = torch.Generator().manual_seed(seed + 1)
    x = (torch.rand(n, 3, hw, hw, generator=g) * 2 - 1)
    x = x.half().float()
    mom_ref = vae_oracle.encode_moments(sd, cfg, x)
    post = vae.encode(x.to(device).half()).latent_dist
    mom = post.parameters.float().cpu()
    assert mom.shape == mom_ref.shape, (mom.shape, mom_ref.shape)
    e_enc = float((mom - mom_ref).abs().max() / mom_ref.abs().max())
    mean_ref, std_ref = vae_oracle.posterior(mom_ref)
    assert torch.allclose(post.mode().float().cpu(), mean_ref, atol=tol_enc * float(mom_ref.abs().max()))
    assert torch.allclose(post.std.float().cpu(), std_ref, rtol=5e-2, atol=1e-3)
    # sample() = mean + std * noise with the same noise -> same latents
    gen = torch.Generator(device=device).manual_seed(7)
    z = post.sample(gen)
    assert z.shape == mean_ref.shape and torch.isfinite(z.float()).all()
    # decode the ORACLE's latents on both sides, so that the decoder comparison does not inherit the encoder's error
    zin = (mean_ref * 0.5).half().float()
    img_ref = vae_oracle.decode(sd, cfg, zin)
    img = vae.decode(zin.to(device).half()).sample.float().cpu()
    assert img.shape == img_ref.shape == (n, 3, hw, hw), (img.shape, img_ref.shape)
    e_dec = float((img - img_ref).abs().max() / img_ref.abs().max())
    assert e_enc < tol_enc and e_dec < tol_dec, (e_enc, e_dec)
    return {"enc_rel_err": e_enc, "dec_rel_err": e_dec}
