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

Two transports carry those exchanges.  "rccl" (torch.distributed collectives / batched isend-irecv; gloo in the CPU tests) is the
portable one.  "peer" (`FrameShard.enable_peer_transport()`, csrc/peer.hip) is the MI355X-native one: every rank owns a symmetric heap
that its peers map (hipIpc over xGMI; shared-memory files on the CPU emulation), a sender copies its message straight into the
receivers' heaps with a tiny kernel that then publishes an epoch flag, a receiver queues a one-workgroup wait kernel in front of the
consumer -- no collective call, no host round trip: the ~100 small messages per UNet forward cost two tiny launches each instead of an
RCCL collective each, and the MB-sized ones ride on a side stream under the projections that follow.
"""
import contextlib
import math
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
        self.stats = {"posted": 0, "overlapped": 0, "blocking": 0, "device_side": 0}
        self.heap: Optional["PeerHeap"] = None

    def enable_peer_transport(self, nbytes: int = 1 << 30, device=None, timeout_us: int = 20_000_000):
        """Carry every exchange of this shard through peer-mapped symmetric heaps (csrc/peer.hip) instead of collectives: a collective
        call over ALL ranks of the group -- they exchange the IPC handles of their heaps here, once."""
        if self.world > 1 and self.heap is None:
            self.heap = PeerHeap(self.world, self.rank, nbytes, device, group=self.group, timeout_us=timeout_us)
        return self

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
        if self.heap is not None:
            return self._peer_all_gather(x_local, tag)
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

        if self.heap is not None and self.world > 1:
            return self._peer_fetch(x_local, wanted, resolve, tag)
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

    # -- the same two primitives over the peer-mapped heaps (csrc/peer.hip) ---------------------------------------
    def _peer_all_gather(self, x_local: torch.Tensor, tag: str) -> "PeerPending":
        heap = self.heap
        b, rest = x_local.shape[0], tuple(x_local.shape[2:])
        if self.n_local == self.max_local:
            mine = x_local.contiguous()
        else:  # ragged partition: every rank's block has the size of the largest
            mine = x_local.new_zeros((b, self.max_local) + rest)
            mine[:, : self.n_local] = x_local
        block = _round16(mine.numel() * mine.element_size())
        off = heap.alloc(self.world * block)
        epoch = heap.next_epoch()
        heap.put(mine, off + self.rank * block, list(range(self.world)), epoch)  # into every heap, the own one included

        nb = mine.numel() * mine.element_size()
        even = self.clip_len % self.world == 0

        def finish():
            heap.wait(list(range(self.world)), epoch)
            if even and nb == block:
                # equal blocks, no padding: ONE copy out of the heap, [world, B, F/P, ...] -> [B, F, ...]
                allb = heap.view(off, self.world * block).view(mine.dtype).view((self.world, b, self.max_local) + rest)
                return allb.transpose(0, 1).reshape((b, self.clip_len) + rest) if b > 1 else allb.reshape((1, self.clip_len) + rest).clone()
            blocks = heap.view(off, self.world * block).view(self.world, block)[:, :nb]
            allb = blocks.contiguous().view(mine.dtype).view((self.world, b, self.max_local) + rest)
            return torch.cat([allb[r, :, : len(self.frames_of(r))] for r in range(self.world)], dim=1)
        return PeerPending(self, finish, tag)

    def _peer_fetch(self, x_local: torch.Tensor, wanted, resolve, tag: str) -> "PeerPending":
        """Every (sender, receiver) pair with frames to move is ONE message: the frames in the receiver's slot order, written into the
        region of the receiver's heap that the (pure, rank-independent) plan assigns to that sender."""
        heap = self.heap
        b, rest = x_local.shape[0], tuple(x_local.shape[2:])
        frame = b * math.prod(rest) * x_local.element_size()
        frame_pad = _round16(frame)
        plan = {}  # receiver -> (resolved slots, {sender: [slot indices]})
        for r in range(self.world):
            slots = [resolve(g) for g in wanted(r)]
            by_sender = {}
            for i, g in enumerate(slots):
                if g is not None and self.owner(g) != r:
                    by_sender.setdefault(self.owner(g), []).append(i)
            plan[r] = (slots, by_sender)
        incoming = max((sum(len(v) for v in plan[r][1].values()) for r in range(self.world)), default=0)
        off = heap.alloc(max(incoming, 1) * frame_pad)  # the same allocation on every rank
        epoch = heap.next_epoch()

        def region(r, s):  # offset of sender s's frames inside receiver r's allocation (senders in rank order)
            return off + sum(len(plan[r][1][t]) for t in sorted(plan[r][1]) if t < s) * frame_pad
        for r in range(self.world):
            if r == self.rank or self.rank not in plan[r][1]:
                continue
            idx = plan[r][1][self.rank]
            msg = torch.stack([x_local[:, plan[r][0][i] - self.f0] for i in idx], dim=0)  # [n, B, ...]
            if frame_pad != frame:
                padded = msg.new_zeros((len(idx), frame_pad // msg.element_size()))
                padded[:, : frame // msg.element_size()] = msg.reshape(len(idx), -1)
                msg = padded
            heap.put(msg.contiguous(), region(r, self.rank), [r], epoch)
        slots, by_sender = plan[self.rank]

        def finish():
            out = x_local.new_zeros((b, len(slots)) + rest)
            for i, g in enumerate(slots):
                if g is not None and self.f0 <= g < self.f1:
                    out[:, i] = x_local[:, g - self.f0]
            if by_sender:
                heap.wait(sorted(by_sender), epoch)
                for s_rank, idx in by_sender.items():
                    raw = heap.view(region(self.rank, s_rank), len(idx) * frame_pad).view(len(idx), frame_pad)[:, :frame]
                    got = raw.reshape(-1).view(x_local.dtype).view((len(idx), b) + rest)
                    out[:, idx] = got.transpose(0, 1)
            return out
        return PeerPending(self, finish, tag, counted=bool(by_sender) or any(self.rank in plan[r][1] for r in range(self.world)))

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


def _round16(n: int) -> int:
    return (n + 15) // 16 * 16


class PeerPending:
    """An exchange over the peer heaps: the puts are already queued (stream-ordered kernels); `wait()` queues the one-workgroup wait
    kernel on the compute stream and assembles the result from the own heap.  Nothing blocks the host, no collective is called:
    counted as `device_side` in the shard's statistics."""

    def __init__(self, shard: FrameShard, finish, tag="other", counted=True):
        self.shard, self.finish, self.tag, self.result = shard, finish, tag, None
        self.counted = counted
        if counted:
            shard.stats["posted"] += 1

    def wait(self) -> torch.Tensor:
        if self.finish is not None:
            if self.counted:
                self.shard.stats["device_side"] += 1
                by = self.shard.stats.setdefault("by_tag", {}).setdefault(self.tag, {"overlapped": 0, "blocking": 0})
                by["device_side"] = by.get("device_side", 0) + 1
            self.result = self.finish()
            self.finish = None
        return self.result


class PeerHeap:
    """One rank's symmetric heap (a byte buffer + control words: a flag per sender rank, done counters, an error word) and the mapped
    heaps of its peers.  On the GPU the buffers are ordinary device allocations shared through HIP IPC (PyTorch's CUDA-IPC tensor
    reducer: hipIpcGetMemHandle / hipIpcOpenMemHandle, dmabuf mode) -- over xGMI a peer's stores land in this rank's HBM; under the CPU
    emulation backend they are shared-memory files, so the gloo test ranks exercise the very same protocol.

    Offsets are handed out by a deterministic bump allocator that wraps around: every rank calls `alloc` in the same order with the
    same sizes (the exchange plans are pure functions of the clip geometry), so an offset means the same message everywhere.  Reuse
    is safe two exchanges later at the earliest -- a rank passes the wait of exchange e only after every sender has put e, and a
    sender's put e is queued behind its own consumer of exchange e - 1 -- and the ring is sized for hundreds."""
    N_FLAGS, N_DONE = 64, 64
    SIDE_STREAM_BYTES = 256 << 10  # messages at least this large are put from a side stream (they overlap the kernels queued next)

    def __init__(self, world: int, rank: int, nbytes: int, device=None, group=None, timeout_us: int = 20_000_000):
        import torch.distributed as dist
        from . import _native as N
        if world > self.N_FLAGS:
            raise ValueError("PeerHeap: at most 64 ranks")
        self.world, self.rank, self.nbytes, self.timeout_us = world, rank, int(nbytes), int(timeout_us)
        self.cursor, self.epoch, self._put_count = 0, 0, 0
        self._ptrs, self._flag_arrays = None, {}
        self._side_event = None  # completion of the newest side-stream put (puts must publish their epochs in order, see put())
        self.on_gpu = not N.is_test_backend()
        self._side = None
        if self.on_gpu:
            from torch.multiprocessing.reductions import reduce_tensor
            device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
            self.ctl = torch.zeros(self.N_FLAGS + self.N_DONE + 8, dtype=torch.int32, device=device)
            torch.cuda.synchronize(device)
            handles = [None] * world
            dist.all_gather_object(handles, (reduce_tensor(self.buf), reduce_tensor(self.ctl)), group=group)
            self.peer_buf, self.peer_ctl = [], []
            for r in range(world):
                if r == rank:
                    self.peer_buf.append(self.buf)
                    self.peer_ctl.append(self.ctl)
                else:
                    (fb, ab), (fc, ac) = handles[r]
                    self.peer_buf.append(fb(*ab))
                    self.peer_ctl.append(fc(*ac))
        else:
            token = [f"{os.getpid()}_{os.environ.get('MASTER_PORT', '0')}"]
            dist.broadcast_object_list(token, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
            base = f"/dev/shm/fz_peer_{token[0]}_"

            def open_(r, create):
                n_ctl = self.N_FLAGS + self.N_DONE + 8
                fb, fc = base + f"{r}_buf", base + f"{r}_ctl"
                if create:
                    with open(fb, "wb") as f:
                        f.truncate(self.nbytes)
                    with open(fc, "wb") as f:
                        f.truncate(4 * n_ctl)
                return (torch.from_file(fb, shared=True, size=self.nbytes, dtype=torch.uint8),
                        torch.from_file(fc, shared=True, size=n_ctl, dtype=torch.int32))
            mine = open_(rank, True)
            dist.barrier(group=group)
            pairs = [mine if r == rank else open_(r, False) for r in range(world)]
            dist.barrier(group=group)
            for suffix in ("_buf", "_ctl"):  # the mappings stay valid; the names go
                try:
                    os.unlink(base + f"{rank}{suffix}")
                except OSError:
                    pass
            self.buf, self.ctl = mine
            self.peer_buf = [p[0] for p in pairs]
            self.peer_ctl = [p[1] for p in pairs]
        dist.barrier(group=group)

    # -- offsets and epochs: identical sequences on every rank ----------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        nbytes = (int(nbytes) + 255) // 256 * 256
        if nbytes * 3 > self.nbytes:
            raise ValueError(f"PeerHeap of {self.nbytes} bytes is too small for a {nbytes}-byte exchange (needs 3x)")
        if self.cursor + nbytes > self.nbytes:
            self.cursor = 0
        off = self.cursor
        self.cursor += nbytes
        return off

    def next_epoch(self) -> int:
        self.epoch = (self.epoch + 1) & 0x7FFFFFFF
        return self.epoch

    def view(self, off: int, nbytes: int) -> torch.Tensor:
        return self.buf[off: off + nbytes]

    # -- the two kernels ------------------------------------------------------------------------------------------
    def put(self, msg: torch.Tensor, off: int, ranks: Sequence[int], epoch: int):
        """Copy `msg` (contiguous) to byte offset `off` of the heaps of `ranks` and publish `epoch` in their flag word for this rank."""
        import ctypes as C
        from . import _native as N
        from . import kernels as K
        assert msg.is_contiguous()
        nbytes = msg.numel() * msg.element_size()
        if nbytes % 16 or msg.data_ptr() % 16:
            flat = msg.reshape(-1).view(torch.uint8)
            padded = torch.zeros(_round16(nbytes), dtype=torch.uint8, device=msg.device)
            padded[:nbytes] = flat
            msg, nbytes = padded, padded.numel()
        assert off % 16 == 0 and off + nbytes <= self.nbytes
        # A receiver waits with `flag >= epoch` on ONE flag word per sender, so a sender's puts must complete in the order of their
        # epochs: a small put (epoch e + 1, main stream) overtaking a large one (epoch e, side stream) would let wait(e) pass before
        # the panel has landed, and the late put would then move the flag BACK to e.  Side-stream puts are ordered among themselves
        # (one stream) and behind everything queued on the main stream at their post (wait_stream); a main-stream put waits for the
        # newest side-stream put's event.  Re-use of a heap region is ordered by the same chain.
        side = None
        if self.on_gpu and nbytes >= self.SIDE_STREAM_BYTES:
            if self._side is None:
                self._side = torch.cuda.Stream(device=msg.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream(msg.device))
        elif self.on_gpu and self._side_event is not None:
            torch.cuda.current_stream(msg.device).wait_event(self._side_event)
            self._side_event = None
        if self._ptrs is None:  # base addresses of every heap / control block, read once
            self._ptrs = ([t.data_ptr() for t in self.peer_buf], [t.data_ptr() for t in self.peer_ctl], self.ctl.data_ptr())
        bufp, ctlp, myctl = self._ptrs
        ranks = list(ranks)
        for i0 in range(0, len(ranks), 8):
            grp = ranks[i0: i0 + 8]
            dst = (C.c_void_p * len(grp))(*[bufp[r] + off for r in grp])
            last = i0 + 8 >= len(ranks)
            key = (tuple(grp), last)
            flg = self._flag_arrays.get(key)
            if flg is None:
                # every put but the last of a multi-group message publishes to a scratch flag (puts of one stream run in order)
                flg = (C.c_void_p * len(grp))(*([ctlp[r] + 4 * self.rank for r in grp] if last else
                                                [myctl + 4 * (self.N_FLAGS + self.N_DONE + 1)] * len(grp)))
                self._flag_arrays[key] = flg
            done = myctl + 4 * (self.N_FLAGS + self._put_count % self.N_DONE)
            self._put_count += 1
            if side is not None:
                with torch.cuda.stream(side):
                    rc = N.lib().fz_peer_put(msg.data_ptr(), nbytes, dst, flg, len(grp), epoch, done, K._stream(msg))
                    self._side_event = side.record_event()
                msg.record_stream(side)
            else:
                rc = N.lib().fz_peer_put(msg.data_ptr(), nbytes, dst, flg, len(grp), epoch, done, K._stream(msg))
            N.check(rc, "fz_peer_put")

    def wait(self, senders: Sequence[int], epoch: int):
        from . import _native as N
        from . import kernels as K
        mask = 0
        for r in senders:
            mask |= 1 << r
        base = self.ctl.data_ptr()
        rc = N.lib().fz_peer_wait(base, mask, epoch, base + 4 * (self.N_FLAGS + self.N_DONE), self.timeout_us, K._stream(self.ctl))
        if rc:
            N.check(rc, "fz_peer_wait")

    def check(self):
        """Synchronising: did any wait time out since the last check?"""
        if int(self.ctl[self.N_FLAGS + self.N_DONE]) != 0:
            self.ctl[self.N_FLAGS + self.N_DONE] = 0
            raise RuntimeError("PeerHeap: a peer did not deliver within the time limit (fz_peer_wait timed out)")


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
