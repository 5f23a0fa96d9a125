"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo in CPU tests.

Round-1 sharding: the job is data-parallel over CLIPS -- every rank runs the whole inversion -> edit job on its own
clip, with no collective inside the UNet.  RCCL only ever carries latents (SURVEY.md §8e "start / end"): rank 0's
weights checksum / prompts are broadcast-checked, and the edited latents are all-gathered.  Frame-sharding ONE clip
across ranks needs the exchanges listed in SURVEY.md §8e (GroupNorm partial sums, anchor/neighbour K/V, temporal
halo); the GroupNorm kernels are already split into stats / finalize / apply for that purpose, the exchanges
themselves are future work.
"""
import os
from typing import Callable, List, Optional

import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: Optional[str] = None):
    import torch.distributed as dist
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend)
    return world, rank, local


def clips_for_rank(n_clips: int, world: int, rank: int) -> List[int]:
    """Contiguous block partition of clip indices (ragged when n_clips % world != 0)."""
    base, rem = divmod(n_clips, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def weights_agree(model, device) -> bool:
    import torch.distributed as dist
    chk = torch.stack([p.detach().float().sum() for p in list(model.parameters())[:16]]).sum().reshape(1).to(device)
    ref = chk.clone()
    if dist.is_initialized():
        dist.broadcast(ref, 0)
    return bool(torch.allclose(ref, chk))


def edit_clips(job: Callable[[int], torch.Tensor], n_clips: int, device) -> Optional[List[torch.Tensor]]:
    """Run `job(clip_index) -> edited latents [1,4,F,h,w]` for this rank's clips and all-gather the results.
    Returns the list of all clips' latents (in clip order) on every rank."""
    import torch.distributed as dist
    world, rank, _ = env_world()
    mine = clips_for_rank(n_clips, world, rank)
    outs = [job(i) for i in mine]
    if world == 1 or not dist.is_initialized():
        return outs
    per = max(len(clips_for_rank(n_clips, world, r)) for r in range(world))
    shape = None
    if outs:
        shape = list(outs[0].shape)
    shp = [shape]
    gathered_shapes = [None] * world
    dist.all_gather_object(gathered_shapes, shp)
    shape = next(s[0] for s in gathered_shapes if s[0] is not None)
    dtype = outs[0].dtype if outs else torch.float32
    pad = torch.zeros([per] + shape, dtype=dtype, device=device)
    for j, o in enumerate(outs):
        pad[j] = o
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    result = []
    for r in range(world):
        for j in range(len(clips_for_rank(n_clips, world, r))):
            result.append(bufs[r][j])
    return result
