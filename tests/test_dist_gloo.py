"""The N > 1 path on CPU: world_size 2, gloo, emulation backend.  Each rank edits its own clip with the tiny model; the
all-gathered latents must equal what a single process computes for the same clips."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _job_factory():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fatezero_amd import _native, build
    _native.use_test_backend(build.build_emu())
    import pipeline_cases as PC
    from fatezero_amd.synthetic import WordTokenizer
    from fatezero_amd.video_diffusion.pipelines.p2p_ddim_spatial_temporal import P2pDDIMSpatioTemporalPipeline
    from fatezero_amd.video_diffusion.schedulers import DDIMScheduler
    unet = PC.build_unet("tiny16", {"lora": 16, "SparseCausalAttention_index": ["mid"]}, "cpu")
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=WordTokenizer(), unet=unet,
                                         scheduler=DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(2, 77, 64, generator=g)
    pipe._encode_prompt = lambda *a, **k: emb

    def job(i):
        gi = torch.Generator().manual_seed(100 + i)
        z0 = torch.randn(1, 4, 2, 8, 8, generator=gi)
        pipe.scheduler.set_timesteps(2)
        pipe.store_controller = type(pipe.store_controller)()
        lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1, text_embeddings=emb,
                                                 store_attention=True, LOW_RESOURCE=True, latents=z0)
        out = pipe(prompt="a red car", source_prompt="a blue car", edit_type="swap", num_inference_steps=2,
                   latents=lat[-1], output_type="latent", cross_replace_steps={"default_": 0.5}, self_replace_steps=0.5,
                   use_inversion_attention=True, is_replace_controller=True, save_self_attention=False, guidance_scale=3.0)
        return out["sdimage_output"].images
    return pipe, job


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), FZ_EMU_THREADS="2")
    torch.set_num_threads(2)
    from fatezero_amd import dist as D
    import torch.distributed as dist
    D.init("gloo")
    pipe, job = _job_factory()
    assert D.weights_agree(pipe.unet, "cpu")
    res = D.edit_clips(job, n_clips, "cpu")
    if rank == 0:
        q.put([r.clone() for r in res])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [2])  # one clip per rank (a ragged 3-over-2 split is covered by test_clip_partition)
def test_two_ranks_match_single_process(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    _, job = _job_factory()
    from fatezero_amd import _native
    try:
        for i in range(n_clips):
            ref = job(i)
            assert torch.equal(ref, got[i]), f"clip {i} differs between the 2-rank and the single-process run"
    finally:
        _native.reset_backend()


def test_clip_partition():
    from fatezero_amd.dist import clips_for_rank
    for n in (0, 1, 5, 8, 9):
        for w in (1, 2, 4, 8):
            allc = sum((clips_for_rank(n, w, r) for r in range(w)), [])
            assert allc == list(range(n))


# ---------------------------------------------------------------------------------------------------------------
# frame-sharding ONE clip: GroupNorm partials, K/V halos + anchors, temporal conv halo, temporal attention
# ---------------------------------------------------------------------------------------------------------------
def _frame_job_factory(frames, index_list):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fatezero_amd import _native, build
    _native.use_test_backend(build.build_emu())
    import pipeline_cases as PC
    from fatezero_amd.synthetic import WordTokenizer
    from fatezero_amd.video_diffusion.pipelines.p2p_ddim_spatial_temporal import P2pDDIMSpatioTemporalPipeline
    from fatezero_amd.video_diffusion.schedulers import DDIMScheduler
    # procedural weights have non-zero temporal LoRA `up` and temporal-attention output weights: every exchange is live
    unet = PC.build_unet("tiny16", {"lora": 16, "SparseCausalAttention_index": index_list}, "cpu")
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=None, tokenizer=WordTokenizer(), unet=unet,
                                         scheduler=DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(7)
    emb = torch.randn(2, 77, 64, generator=g)
    pipe._encode_prompt = lambda *a, **k: emb
    z0 = torch.randn(1, 4, frames, 8, 8, generator=g)

    def job():
        pipe.scheduler.set_timesteps(2)
        pipe.store_controller = type(pipe.store_controller)()
        lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1, text_embeddings=emb,
                                                 store_attention=True, LOW_RESOURCE=True, latents=z0)
        out = pipe(prompt="a red car", source_prompt="a blue car", edit_type="swap", num_inference_steps=2,
                   latents=lat[-1], output_type="latent", cross_replace_steps={"default_": 0.5}, self_replace_steps=0.5,
                   use_inversion_attention=True, is_replace_controller=True, save_self_attention=False, guidance_scale=3.0)
        return torch.stack([lat[-1], out["sdimage_output"].images])
    return pipe, job


def _frame_worker(rank, world, port, frames, index_list, run_model, q, transport="rccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), FZ_EMU_THREADS="2")
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    from fatezero_amd import dist as D
    import torch.distributed as dist
    D.init("gloo")
    shard = D.FrameShard(frames)
    if transport == "peer":
        # the one-sided transport (csrc/peer.hip) on the CPU emulation: the symmetric heaps are shared-memory files, the put / wait
        # kernels run on the emulator -- the same protocol (offset plan, epochs, per-sender flag words) the GPUs run over hipIpc
        from fatezero_amd import _native, build
        _native.use_test_backend(build.build_emu())
        shard.enable_peer_transport(nbytes=48 << 20, timeout_us=120_000_000)
    # -- the exchange primitives against slicing of the full tensor ---------------------------------------------
    full = torch.arange(2 * frames * 3, dtype=torch.float32).view(2, frames, 3) + 1.0
    loc = shard.local(full, 1).contiguous()
    assert torch.equal(shard.all_gather_frames(loc), full)
    zpad = torch.zeros(2, 2, 3)
    padded = torch.cat([zpad, full, zpad], 1)
    assert torch.equal(shard.with_halo(loc, 1, 2, zero_outside=True), padded[:, shard.f0 + 1: shard.f1 + 4])
    clamped = torch.cat([full[:, :1]] * 2 + [full] + [full[:, -1:]] * 2, 1)
    assert torch.equal(shard.with_halo(loc, 2, 1, zero_outside=False), clamped[:, shard.f0: shard.f1 + 3])
    anchors = shard.fetch_frames(loc, lambda r: [0, (frames - 1) // 2, frames - 1, frames + 3])
    assert torch.equal(anchors, full[:, [0, (frames - 1) // 2, frames - 1, frames - 1]])
    if run_model:
        pipe, job = _frame_job_factory(frames, index_list)
        assert D.weights_agree(pipe.unet, "cpu")
        pipe.frame_shard = shard
        shard.stats = {"posted": 0, "overlapped": 0, "blocking": 0, "device_side": 0}  # (the primitive checks above are not part of a forward)
        res = job()
        if shard.heap is not None:
            shard.heap.check()
        store = pipe.store_controller
        n_maps = [t.shape[0] for t in store.attention_store_all_step[0]["down_self"]] if store.attention_store_all_step else []
        if rank == 0:
            q.put((res.clone(), n_maps, shard.n_local, dict(shard.stats)))
    if shard.heap is not None:
        shard.heap.check()
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, *args, transport="rccl"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world + (7 if transport == "peer" else 0)
    procs = [ctx.Process(target=_frame_worker, args=(r, world, port) + args + (q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=900) if args[2] else None
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    return got


def _check_exchange_structure(stats):
    """What overlaps and what cannot: every sparse-causal K / V^T fetch and every temporal-attention K | V all-gather is posted
    BEFORE the Q projection and waited for after it (never blocking); the temporal convolution costs ONE exchange (its down(x)
    halo is recomputed, not exchanged a second time); the GroupNorm statistics gathers sit between the statistics kernel and the
    normalisation that needs them at once (blocking by data dependence: the layer chain is sequential)."""
    by = stats["by_tag"]
    assert by["kv"]["blocking"] == 0 and by["kv"]["overlapped"] > 0, stats
    assert by["temporal_attn"]["blocking"] == 0 and by["temporal_attn"]["overlapped"] > 0, stats
    # tiny16 UNet: 16 transformer blocks (one K and one V^T fetch + one temporal-attention gather each) per forward
    assert by["kv"]["overlapped"] == 2 * by["temporal_attn"]["overlapped"], stats
    n_forward = by["temporal_attn"]["overlapped"] // 16
    assert n_forward >= 4 and by["temporal_attn"]["overlapped"] == 16 * n_forward, stats
    # one exchange per temporal LoRA convolution: 53 PseudoConv3d with a LoRA pair per forward at most (conv_in, 22 resnets x 2,
    # 4 downsamplers, 3 upsamplers, conv_out), never two per convolution
    assert by["temporal_conv"]["blocking"] <= 53 * n_forward, stats
    # (+ the all-gather of the finished latents at the end of the inversion and of the edit)
    assert 0 <= stats["blocking"] - by["temporal_conv"]["blocking"] - by["groupnorm"]["blocking"] <= 4, stats


def test_frame_shard_exchanges_three_ranks_ragged():
    _spawn(3, 7, [-1, "first"], False)   # 3 + 2 + 2 frames: a middle rank with two neighbours, ragged all-gather


def test_frame_shard_exchanges_one_frame_per_rank():
    # 8 GPUs x 8 frames in bench.py's probe: a two-frame halo then comes from TWO ranks on each side
    _spawn(4, 4, [-1, "first"], False)


@pytest.mark.parametrize("frames,index_list", [(4, ["mid", 1]), (5, [-1, "first"])])  # 5 frames: ragged 3 + 2 split
def test_frame_sharded_clip_matches_single_process(frames, index_list):
    got, n_maps, n_local, stats = _spawn(2, frames, index_list, True)
    _check_exchange_structure(stats)
    _, job = _frame_job_factory(frames, index_list)
    from fatezero_amd import _native
    try:
        ref = job()
    finally:
        _native.reset_backend()
    # each rank keeps only its own frames' attention maps
    assert n_maps and all(n == n_local for n in n_maps), (n_maps, n_local)
    # same kernels, same inputs; only the merge order of the GroupNorm partials and the shapes of the temporal-conv GEMMs
    # differ (fp16 rounding of down(x)), which 2 + 2 DDIM steps with guidance amplify to a few 1e-3 of the latent range.
    # A wrong halo / anchor frame shows up as tens of percent.
    err = float((got.float() - ref.float()).abs().max())
    scale = float(ref.float().abs().max())
    assert torch.isfinite(got.float()).all()
    assert err <= 1.5e-2 * scale, (err, scale)


def test_frame_sharded_clip_four_ranks_eight_frames():
    """The teaser clip length on 4 ranks (2 frames each): interior ranks exchange halos with both neighbours, rank 0 serves the
    'first' anchor to everybody, every GroupNorm merges 4 ranks' partials."""
    got, n_maps, n_local, stats = _spawn(4, 8, [-1, "first"], True)
    _check_exchange_structure(stats)
    _, job = _frame_job_factory(8, [-1, "first"])
    from fatezero_amd import _native
    try:
        ref = job()
    finally:
        _native.reset_backend()
    assert n_local == 2 and n_maps and all(n == 2 for n in n_maps), (n_maps, n_local)
    err = float((got.float() - ref.float()).abs().max())
    scale = float(ref.float().abs().max())
    assert torch.isfinite(got.float()).all() and err <= 1.5e-2 * scale, (err, scale)


# ---------------------------------------------------------------------------------------------------------------
# the same exchanges over the one-sided peer transport (csrc/peer.hip; shared-memory heaps on the CPU emulation)
# ---------------------------------------------------------------------------------------------------------------
def test_peer_transport_exchanges_three_ranks_ragged():
    _spawn(3, 7, [-1, "first"], False, transport="peer")   # ragged all-gather, a middle rank with two neighbours, anchors


def test_peer_transport_exchanges_one_frame_per_rank():
    _spawn(4, 4, [-1, "first"], False, transport="peer")   # a two-frame halo comes from TWO ranks on each side


@pytest.mark.parametrize("frames,index_list", [(5, [-1, "first"])])  # ragged 3 + 2 split
def test_peer_transport_clip_matches_single_process(frames, index_list):
    """The whole frame-sharded inversion + edit with EVERY exchange on the peer transport: no collective call and no blocking wait is
    left inside the UNet (GroupNorm partials and temporal-convolution halos included) -- every exchange is a put kernel at the sender
    and a one-workgroup wait kernel in front of the consumer."""
    got, n_maps, n_local, stats = _spawn(2, frames, index_list, True, transport="peer")
    assert stats["blocking"] == 0 and stats["overlapped"] == 0 and stats["device_side"] == stats["posted"] > 0, stats
    by = stats["by_tag"]
    for tag in ("kv", "temporal_attn", "groupnorm", "temporal_conv"):
        assert by[tag].get("device_side", 0) > 0 and by[tag]["blocking"] == 0, (tag, stats)
    _, job = _frame_job_factory(frames, index_list)
    from fatezero_amd import _native
    try:
        ref = job()
    finally:
        _native.reset_backend()
    assert n_maps and all(n == n_local for n in n_maps), (n_maps, n_local)
    err = float((got.float() - ref.float()).abs().max())
    scale = float(ref.float().abs().max())
    assert torch.isfinite(got.float()).all() and err <= 1.5e-2 * scale, (err, scale)


def test_peer_transport_clip_eight_ranks_one_frame_each():
    """The driver's 8-GPU configuration in miniature: 8 ranks x 8 frames = ONE frame per rank over the peer transport -- a two-frame
    temporal-conv halo comes from two ranks on each side, the 'first' anchor is served by rank 0 to everybody, every GroupNorm merges
    eight ranks' partials, every put addresses up to 8 heaps."""
    got, n_maps, n_local, stats = _spawn(8, 8, [-1, "first"], True, transport="peer")
    assert stats["blocking"] == 0 and stats["device_side"] == stats["posted"] > 0, stats
    _, job = _frame_job_factory(8, [-1, "first"])
    from fatezero_amd import _native
    try:
        ref = job()
    finally:
        _native.reset_backend()
    assert n_local == 1 and n_maps and all(n == 1 for n in n_maps), (n_maps, n_local)
    err = float((got.float() - ref.float()).abs().max())
    scale = float(ref.float().abs().max())
    assert torch.isfinite(got.float()).all() and err <= 1.5e-2 * scale, (err, scale)
