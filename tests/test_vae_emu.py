"""Native VAE against the CPU restatement of diffusers' AutoencoderKL, on the CPU emulation of the
kernels (test infrastructure; the MI355X version of the same cases is tests/test_vae_gpu.py)."""
import json
import os

import pytest
import torch

from fatezero_amd import _native, build
from fatezero_amd import kernels as K

import vae_cases as VC


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


def test_softmax_rows():
    g = torch.Generator().manual_seed(0)
    for rows, cols in [(5, 16), (9, 264), (3, 64 * 8 * 16 + 8)]:
        x = (torch.randn(rows, cols, generator=g) * 3).half()
        y = K.softmax_rows(x, scale=0.7)
        ref = torch.softmax(x.float() * 0.7, dim=-1)
        assert float((y.float() - ref).abs().max()) < 2e-3 * float(ref.max()) + 1e-6
        assert float((y.float().sum(-1) - 1).abs().max()) < 5e-3


def test_gemm_batched_per_batch_weights():
    # y[b] = x[b] w[b]^T in one launch (FzGemmDesc.w_batch_stride): the two products of the VAE's single-head attention
    g = torch.Generator().manual_seed(0)
    for (b, rows, k, o) in [(3, 70, 64, 40), (2, 33, 40, 136)]:
        x = torch.randn(b, rows, k, generator=g).half()
        w = (torch.randn(b, o, k, generator=g) * k ** -0.5).half()
        y = K.gemm_batched(x, w)
        ref = torch.einsum("brk,bok->bro", x.float(), w.float())
        assert float((y.float() - ref).abs().max()) < 4e-3 * max(1.0, float(ref.abs().max()))
    # views with row / batch strides (the q | k halves of one projection output)
    qk = torch.randn(2, 24, 64, generator=g).half()
    y = K.gemm_batched(qk[:, :, :32], qk[:, :, 32:])
    assert float((y.float() - torch.einsum("brk,bok->bro", qk[:, :, :32].float(), qk[:, :, 32:].float())).abs().max()) < 2e-2


def test_vae_state_dict_of_newer_diffusers_loads():
    """AutoencoderKL folders written by diffusers >= 0.15 name the mid-block attention to_q / to_k / to_v / to_out.0; older
    conversion scripts stored those projections as 1x1 convolutions: both load (diffusers remaps them too)."""
    vae, sd = VC.seeded_vae(VC.TINY, seed=5)
    newer = {}
    for k, v in sd.items():
        for old, new in (("query", "to_q"), ("key", "to_k"), ("value", "to_v"), ("proj_attn", "to_out.0")):
            if f".attentions.0.{old}." in k:
                k = k.replace(f".{old}.", f".{new}.")
                if k.endswith(".weight"):
                    v = v[:, :, None, None]
        newer[k] = v
    assert any(".to_q." in k for k in newer)
    vae2, _ = VC.seeded_vae(VC.TINY, seed=6)
    vae2.load_state_dict(newer)
    for (k1, v1), (k2, v2) in zip(vae.state_dict().items(), vae2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1


def test_vae_encode_decode_tiny():
    r = VC.case_vae_roundtrip("cpu", VC.TINY, n=2, hw=16)
    print(r)


def test_vae_odd_sizes_and_slicing():
    vae, sd = VC.seeded_vae(VC.TINY, seed=3)
    vae = vae.half()
    x = (torch.rand(3, 3, 24, 8) * 2 - 1).half()
    a = vae.encode(x).latent_dist.mode()
    vae.enable_slicing()
    b = vae.encode(x).latent_dist.mode()
    assert a.shape == (3, 4, 12, 4) and torch.equal(a, b)
    from oracle import vae_oracle
    ref = vae_oracle.posterior(vae_oracle.encode_moments(sd, VC.TINY, x.float()))[0]
    assert float((a.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


def test_from_pretrained_layout(tmp_path):
    """diffusers folder layout: <path>/vae/config.json + diffusion_pytorch_model.bin, key names of the 0.11.1 checkpoint."""
    vae, sd = VC.seeded_vae(VC.TINY, seed=5)
    root = tmp_path / "ckpt" / "vae"
    os.makedirs(root)
    json.dump(dict(VC.TINY, _class_name="AutoencoderKL", _diffusers_version="0.11.1"), open(root / "config.json", "w"))
    torch.save(sd, root / "diffusion_pytorch_model.bin")
    from fatezero_amd.video_diffusion.models.vae import AutoencoderKL
    loaded = AutoencoderKL.from_pretrained(str(tmp_path / "ckpt"), subfolder="vae")
    assert loaded.config.block_out_channels == (32, 64)
    for k in ("encoder.down_blocks.0.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.query.weight",
              "decoder.up_blocks.0.upsamplers.0.conv.bias", "quant_conv.weight", "post_quant_conv.bias",
              "decoder.up_blocks.1.resnets.0.conv_shortcut.weight"):
        assert k in loaded.state_dict(), k
    assert torch.equal(loaded.state_dict()["quant_conv.weight"], sd["quant_conv.weight"])
