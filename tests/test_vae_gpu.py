"""Native VAE on MI355X against the CPU restatement of diffusers' AutoencoderKL."""
import pytest
import torch

from fatezero_amd import kernels as K

import vae_cases as VC

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_softmax_rows_gpu():
    g = torch.Generator().manual_seed(0)
    for rows, cols in [(5, 16), (1024, 1024), (4096, 4096), (3, 64 * 8 * 16 + 8)]:
        x = (torch.randn(rows, cols, generator=g) * 3).half().to(DEV)
        y = K.softmax_rows(x, scale=0.7)
        ref = torch.softmax(x.float() * 0.7, dim=-1)
        assert float((y.float() - ref).abs().max()) < 2e-3 * float(ref.max()) + 1e-6


def test_vae_tiny_gpu():
    print(VC.case_vae_roundtrip(DEV, VC.TINY, n=2, hw=16))


def test_vae_sd_architecture_gpu():
    # the real SD-1.x VAE (128/256/512/512, 2 layers per block, 512-wide single-head attention) on 128x128 frames
    print(VC.case_vae_roundtrip(DEV, VC.SD, n=2, hw=128, seed=1, tol_enc=3e-2, tol_dec=4e-2))


def test_vae_through_the_pipeline_surface():
    # the two call sites of the reference: encode(...).latent_dist.sample() * 0.18215 and decode_latents (chunks of 16 frames)
    from fatezero_amd.video_diffusion.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline
    vae, _ = VC.seeded_vae(VC.TINY, seed=2)
    vae = vae.to(DEV).half()
    pipe = SpatioTemporalStableDiffusionPipeline.__new__(SpatioTemporalStableDiffusionPipeline)
    pipe.vae = vae
    frames = (torch.rand(5, 3, 32, 32, device=DEV) * 2 - 1).half()
    lat = vae.encode(frames).latent_dist.sample() * 0.18215
    assert lat.shape == (5, 4, 16, 16)
    video = lat.view(1, 5, 4, 16, 16).permute(0, 2, 1, 3, 4)
    img = pipe.decode_latents(video)
    assert img.shape == (1, 5, 32, 32, 3) and img.min() >= 0.0 and img.max() <= 1.0
