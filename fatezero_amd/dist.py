"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo in CPU tests.

Two ways to use N GPUs:

* clips (default of bench.py --gpus N): the job is data-parallel over CLIPS -- every rank runs the whole inversion ->
  edit job on its own clip, with no collective inside the UNet.  RCCL only ever carries latents: rank 0's weights
  checksum is broadcast-checked and the edited latents are all-gathered.
* frames (`FrameShard`): ONE clip's frames are partitioned over the ranks (each keeps its slice of the HBM map arena
  and of the blend masks).  Frames are coupled in four places, and those are the only collectives on the data path:
    - 5-D GroupNorm statistics span all frames  -> all-gather of the Welford partials, merged identically everywhere
      (fz_groupnorm_stats / fz_groupnorm_apply);
    - sparse-causal K/V of neighbour / anchor frames -> point-to-point fetch of those frames' K and V^T from their owners
      into an extended K/V frame axis the attention kernels index directly (FzAttnSelfDesc.kv_clip_len);
    - the k=3 temporal LoRA convolution           -> ONE two-frame halo exchange of x with the neighbours (down(x) of the halo
      frames is recomputed locally instead of exchanged a second time);
    - temporal attention over all F frames per pixel -> all-gather of that layer's K and V.
  CFG, the DDIM update, blend masks (normalised per frame) and the latent blend are per-frame and need nothing.
"""
import contextlib
import os
from typing import Callable, List, Optional, Sequence

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
    # a rank without clips (n_clips < world) learns shape AND dtype of the results from the others: the all_gather below needs
    # identical buffers on every rank
    meta = [(list(outs[0].shape), str(outs[0].dtype).replace("torch.", "")) if outs else None]
    gathered = [None] * world
    dist.all_gather_object(gathered, meta)
    shape, dtype_name = next(g[0] for g in gathered if g[0] is not None)
    dtype = getattr(torch, dtype_name)
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


# ---------------------------------------------------------------------------------------------------------------
# frame-sharding one clip
# ---------------------------------------------------------------------------------------------------------------
class FrameShard:
    """Contiguous block partition of the F frames of one clip over the ranks of a process group."""

    def __init__(self, clip_len: int, group=None, rank: Optional[int] = None, world: Optional[int] = None):
        import torch.distributed as dist
        self.group = group
        self.world = world if world is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        if clip_len < self.world:
            raise ValueError(f"cannot shard {clip_len} frames over {self.world} ranks: every rank needs a frame")
        self.clip_len = clip_len
        base, rem = divmod(clip_len, self.world)
        self.bounds = [r * base + min(r, rem) for r in range(self.world + 1)]
        self.f0, self.f1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.n_local = self.f1 - self.f0
        self.max_local = base + (1 if rem else 0)
        self.stats = {"posted": 0, "overlapped": 0, "blocking": 0}

    # -- bookkeeping ---------------------------------------------------------------------------------------------
    def frames_of(self, rank: int) -> range:
        return range(self.bounds[rank], self.bounds[rank + 1])

    def owner(self, frame: int) -> int:
        for r in range(self.world):
            if frame < self.bounds[r + 1]:
                return r
        raise IndexError(frame)

    def _peer(self, rank: int) -> int:
        import torch.distributed as dist
        return rank if self.group is None else dist.get_global_rank(self.group, rank)

    def local(self, x: torch.Tensor, dim: int) -> torch.Tensor:
        """This rank's frames of a tensor that holds all F frames along `dim`."""
        return x.narrow(dim, self.f0, self.n_local)

    # -- collectives ---------------------------------------------------------------------------------------------
    # Every exchange is POSTED (returns a `Pending`) and WAITED for separately: RCCL runs the transfer on its own stream, ordered
    # behind the work already queued on the compute stream at post time, and `wait()` only makes the compute stream depend on
    # its completion -- whatever the caller launches between post and wait overlaps with the transfer (the K / V^T fetches ride
    # under the Q projection, attention.py).  `stats` counts, per shard object, how many waits had compute launched in between
    # (`overlapped`) and how many came straight after the post (`blocking`): tests/test_dist_gloo.py asserts the structure.
    def all_gather_frames_async(self, x_local: torch.Tensor, tag: str = "other") -> "Pending":
        """x_local [B, F_local, ...] -> Pending of [B, F, ...] on every rank (frames in clip order)."""
        import torch.distributed as dist
        if self.world == 1:
            return Pending(self, [], lambda: x_local, tag=tag)
        b = x_local.shape[0]
        rest = tuple(x_local.shape[2:])
        if self.n_local == self.max_local:
            mine = x_local.contiguous()
        else:  # ragged partition: pad to the largest block
            mine = x_local.new_zeros((b, self.max_local) + rest)
            mine[:, : self.n_local] = x_local
        bufs = [torch.empty_like(mine) for _ in range(self.world)]
        work = dist.all_gather(bufs, mine, group=self.group, async_op=True)
        return Pending(self, [work], lambda: torch.cat([bufs[r][:, : len(self.frames_of(r))] for r in range(self.world)], dim=1),
                       keep=(mine,), tag=tag)

    def all_gather_frames(self, x_local: torch.Tensor, tag: str = "other") -> torch.Tensor:
        return self.all_gather_frames_async(x_local, tag).wait()

    def fetch_frames_async(self, x_local: torch.Tensor, wanted: Callable[[int], Sequence[int]], *,
                           zero_outside: bool = False, tag: str = "other") -> "Pending":
        """Point-to-point gather of whole frames.  `wanted(rank)` lists the GLOBAL frame indices rank `rank` needs (the
        same pure function on every rank, so each one also knows what to send).  Indices outside [0, F) are clamped
        into the clip (sparse-causal attention, attention.py:383-386) or, with zero_outside, return zeros (the zero
        padding of the temporal convolution).  x_local: [B, F_local, ...] -> Pending of [B, len(wanted(my rank)), ...]."""
        import torch.distributed as dist

        def resolve(g):
            if 0 <= g < self.clip_len:
                return g
            return None if zero_outside else min(max(g, 0), self.clip_len - 1)

        mine = [resolve(g) for g in wanted(self.rank)]
        b, rest = x_local.shape[0], tuple(x_local.shape[2:])
        out = x_local.new_zeros((b, len(mine)) + rest)
        ops, recvs, keep = [], [], []
        for r in range(self.world):
            if r == self.rank:
                continue
            theirs = [resolve(g) for g in wanted(r)]
            send = [g for g in theirs if g is not None and self.f0 <= g < self.f1]
            if send:  # one message per (sender, receiver): the frames in the receiver's slot order
                buf = torch.stack([x_local[:, g - self.f0] for g in send], dim=1).contiguous()
                keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, self._peer(r), group=self.group))
            slots = [i for i, g in enumerate(mine) if g is not None and self.owner(g) == r]
            if slots:
                buf = x_local.new_empty((b, len(slots)) + rest)
                recvs.append((slots, buf))
                ops.append(dist.P2POp(dist.irecv, buf, self._peer(r), group=self.group))
        for i, g in enumerate(mine):
            if g is not None and self.f0 <= g < self.f1:
                out[:, i] = x_local[:, g - self.f0]
        reqs = dist.batch_isend_irecv(ops) if ops else []

        def finish():
            for slots, buf in recvs:
                out[:, slots] = buf
            return out
        return Pending(self, reqs, finish, keep=tuple(keep), tag=tag)

    def fetch_frames(self, x_local: torch.Tensor, wanted: Callable[[int], Sequence[int]], *,
                     zero_outside: bool = False, tag: str = "other") -> torch.Tensor:
        return self.fetch_frames_async(x_local, wanted, zero_outside=zero_outside, tag=tag).wait()

    def with_halo(self, x_local: torch.Tensor, left: int, right: int, *, zero_outside: bool, tag: str = "halo") -> torch.Tensor:
        """[B, F_local, ...] -> [B, left + F_local + right, ...]: the neighbours' boundary frames on both sides."""
        def wanted(r):
            fr = self.frames_of(r)
            return list(range(fr.start - left, fr.start)) + list(range(fr.stop, fr.stop + right))
        halo = self.fetch_frames(x_local, wanted, zero_outside=zero_outside, tag=tag)
        return torch.cat([halo[:, :left], x_local, halo[:, left:]], dim=1)


class Pending:
    """An exchange in flight: `wait()` completes it and returns the assembled tensor."""

    def __init__(self, shard: FrameShard, reqs, finish, keep=(), tag="other"):
        from . import kernels as K
        self.shard, self.reqs, self.finish, self.keep, self.tag = shard, list(reqs), finish, keep, tag
        self.launches_at_post = K.launch_count()
        self.result = None
        if self.reqs:
            shard.stats["posted"] += 1

    def wait(self) -> torch.Tensor:
        if self.finish is not None:
            from . import kernels as K
            if self.reqs:
                kind = "overlapped" if K.launch_count() > self.launches_at_post else "blocking"
                self.shard.stats[kind] += 1
                by = self.shard.stats.setdefault("by_tag", {}).setdefault(self.tag, {"overlapped": 0, "blocking": 0})
                by[kind] += 1
            for req in self.reqs:
                req.wait()
            self.result = self.finish()
            self.finish, self.reqs, self.keep = None, [], ()
        return self.result


_active_shard: Optional[FrameShard] = None


def active_shard() -> Optional[FrameShard]:
    """The FrameShard the model code must honour right now (None: all frames of the clip are local)."""
    return _active_shard


@contextlib.contextmanager
def frame_sharded(shard: Optional[FrameShard]):
    global _active_shard
    prev = _active_shard
    _active_shard = shard if (shard is not None and shard.world > 1) else None
    try:
        yield shard
    finally:
        _active_shard = prev
