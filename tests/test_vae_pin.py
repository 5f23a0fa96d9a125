"""Row (f)-1, the VAE: what CAN be pinned offline (neither diffusers nor an SD checkpoint exists here).

  1. the checkpoint KEY LAYOUT: every tensor name and shape of the SD-1.x `vae/diffusion_pytorch_model.bin`, written out from the published
     config + diffusers 0.11.1's module tree (oracle/vae_ldm_ref.py: sd_v1_vae_key_shapes, 248 tensors / 83 653 863 parameters), must be
     exactly the native AutoencoderKL's state dict, and a state dict with those names must load strictly;
  2. the restatement of diffusers' AutoencoderKL the native VAE is tested against (oracle/vae_oracle.py) must read every one of those
     tensors, and must agree with a SECOND statement of the same network taken from a different published source -- the original CompVis
     latent-diffusion autoencoder, different module structure / key names / block order / attention arithmetic -- through the published
     key mapping between the two formats.
The native VAE vs vae_oracle comparison itself lives in tests/test_vae_emu.py / test_vae_gpu.py."""
import math

import torch

import vae_cases as VC
from oracle import vae_ldm_ref as L
from oracle import vae_oracle as O


class _Recording(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.read = set()

    def __getitem__(self, k):
        self.read.add(k)
        return super().__getitem__(k)


def _random_sd(key_shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in key_shapes.items():
        if k.endswith(".bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * (1.0 / math.prod(shp[1:])) ** 0.5
    return sd


def test_sd_v1_vae_checkpoint_key_layout():
    ks = L.sd_v1_vae_key_shapes()
    assert len(ks) == 248 and sum(math.prod(s) for s in ks.values()) == 83653863
    from fatezero_amd.video_diffusion.models.vae import AutoencoderKL
    with torch.device("meta"):
        vae = AutoencoderKL.from_config(VC.SD)
    mine = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    assert set(mine) == set(ks), (sorted(set(mine) - set(ks))[:5], sorted(set(ks) - set(mine))[:5])
    assert mine == {k: tuple(s) for k, s in ks.items()}
    # (0.11.1 names the mid-block attention tensors group_norm / query / key / value / proj_attn -- Linear, not conv)
    assert ks["decoder.mid_block.attentions.0.query.weight"] == (512, 512) and ks["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1)


def _reduced_key_shapes(ch, lpb, latent=4):
    """the same tree at reduced widths (CI size), built by the same rules as sd_v1_vae_key_shapes: taken from the native model, whose
    equality with the written-out SD layout the test above establishes."""
    from fatezero_amd.video_diffusion.models.vae import AutoencoderKL
    cfg = dict(VC.SD, block_out_channels=list(ch), layers_per_block=lpb, norm_num_groups=8, latent_channels=latent,
               down_block_types=["DownEncoderBlock2D"] * len(ch), up_block_types=["UpDecoderBlock2D"] * len(ch))
    with torch.device("meta"):
        vae = AutoencoderKL.from_config(cfg)
    return cfg, {k: tuple(v.shape) for k, v in vae.state_dict().items()}


def _compare(cfg, key_shapes, hw, groups, seed):
    sd = _Recording(_random_sd(key_shapes, seed))
    nb, lpb = len(cfg["block_out_channels"]), cfg["layers_per_block"]
    ldm = L.diffusers_to_ldm(dict(sd), nb, lpb)
    assert len(ldm) == len(sd)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(2, 3, hw, hw, generator=g) * 2 - 1
    m_a = O.encode_moments(sd, cfg, x)
    m_b = L.encode_moments(ldm, nb, lpb, x, groups=groups)
    z = torch.randn(2, cfg["latent_channels"], hw // 2 ** (nb - 1), hw // 2 ** (nb - 1), generator=g)
    d_a = O.decode(sd, cfg, z)
    d_b = L.decode(ldm, nb, lpb, z, groups=groups)
    assert sd.read == set(key_shapes), sorted(set(key_shapes) - sd.read)[:5]  # the restatement consumes every tensor of the checkpoint
    e_m = float((m_a - m_b).abs().max() / m_b.abs().max())
    e_d = float((d_a - d_b).abs().max() / d_b.abs().max())
    assert m_a.shape == m_b.shape and d_a.shape == d_b.shape == (2, 3, hw, hw)
    return e_m, e_d


def test_diffusers_restatement_agrees_with_the_compvis_statement_reduced():
    cfg, ks = _reduced_key_shapes((32, 64, 64), 2)
    e_m, e_d = _compare(cfg, ks, 32, 8, seed=3)
    print("reduced VAE: diffusers restatement vs CompVis statement", e_m, e_d)
    assert e_m < 2e-5 and e_d < 2e-5, (e_m, e_d)


def test_diffusers_restatement_agrees_with_the_compvis_statement_sd_architecture():
    """The real SD-1.x VAE architecture (248 tensors, 83.7 M parameters) on the written-out checkpoint layout, 64 x 64 images."""
    e_m, e_d = _compare(VC.SD, L.sd_v1_vae_key_shapes(), 64, 32, seed=5)
    print("SD-1.x VAE: diffusers restatement vs CompVis statement", e_m, e_d)
    assert e_m < 2e-5 and e_d < 2e-5, (e_m, e_d)
