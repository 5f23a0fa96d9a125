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


def test_vae_sd_architecture_512_frames_vs_restated_diffusers_oracle_gpu():
    """The real frame size of BASELINE cfg2 (stable_diffusion.py:297-319 decodes 512 x 512 frames; p2p_ddim_spatial_temporal.py:94-96
    encodes them): one 512^2 frame through the SD-1.x VAE architecture against oracle/vae_oracle.py -- a RESTATEMENT of diffusers'
    AutoencoderKL (the package is absent offline: this parity is unpinned, as DESIGN.md says) -- mid-block attention over 4096
    tokens included; then the timing of the 8-frame clip (encode + decode) recorded for profiles/."""
    import time
    print("512^2 parity:", VC.case_vae_roundtrip(DEV, VC.SD, n=1, hw=512, seed=3, tol_enc=3e-2, tol_dec=4e-2))
    vae, _ = VC.seeded_vae(VC.SD, seed=3)
    vae = vae.to(DEV).half()
    frames = (torch.rand(8, 3, 512, 512, device=DEV) * 2 - 1).half()
    for _ in range(2):
        lat = vae.encode(frames).latent_dist.mode()
        img = vae.decode(lat).sample
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat = vae.encode(frames).latent_dist.mode()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    img = vae.decode(lat).sample
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert lat.shape == (8, 4, 64, 64) and img.shape == (8, 3, 512, 512) and torch.isfinite(img.float()).all()
    print(f"VAE 8 x 512^2 frames on MI355X: encode {1e3 * (t1 - t0):.1f} ms, decode {1e3 * (t2 - t1):.1f} ms")


def test_gemm_batched_per_batch_weights():
    g = torch.Generator().manual_seed(0)
    for (b, rows, k, o) in [(3, 300, 64, 200), (2, 4096, 512, 4096), (8, 100, 104, 72)]:
        x = torch.randn(b, rows, k, generator=g).half().to(DEV)
        w = (torch.randn(b, o, k, generator=g) * k ** -0.5).half().to(DEV)
        y = K.gemm_batched(x, w)
        ref = torch.einsum("brk,bok->bro", x.float(), w.float())
        assert float((y.float() - ref).abs().max()) < 4e-3 * max(1.0, float(ref.abs().max()))


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
