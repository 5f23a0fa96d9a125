"""Native issue path for the UNet forward: the launch list recorded once, replayed from native code at the following DDIM steps.

The reference issues `self.unet(latent_model_input, t, encoder_hidden_states=...)` (p2p_ddim_spatial_temporal.py:286,
stable_diffusion.py:360-372) by walking the module tree in Python at every one of the 50 + 50 steps of a job.  The walk decides ~520-700
kernel launches (descriptors, grids, pointers) that are a pure function of (clip geometry, text context, controller kind): only data behind
pointers differs between steps.  `IssuePlans.run` therefore

  1. lets the FIRST forward of a kind run through Python untouched (weight packs, the text context's K / V^T, scratch buffers get cached);
     a kind = (clip geometry, shape of the text context, controller class and `issue_signature()`), NOT the job: the plans of one job serve
     the next one -- a new context of the same shape is projected into the K / V^T buffers the records already point at (`bind_context`),
     a new controller object answers the same events;
  2. RECORDS the second one (`fz_plan_begin` ... `fz_plan_end`, csrc/plan.hip): the forward runs as usual and every launch the library
     makes is appended to a native plan.  What the attention CONTROLLER does stays live: each `controller.attention_plan(...)` call of the
     walk becomes an *event* -- recording is paused around it (its own launches, e.g. the blend-mask kernels, are issued by the controller at
     every replay) and the tensors of the `AttnPlan` it returns are remembered with the index of the first launch that may use them;
  3. REPLAYS from the third forward on: the records between two events are re-issued by ONE native call each (`fz_plan_replay`), the
     controller is asked for the step's plan at each event exactly as the walk would (its counters, arena slots and masks advance as in the
     reference protocol), and where the step's tensors live elsewhere than the recorded ones -- the step's slab of the map arena, the
     per-step cross-attention coefficients, a fresh mask -- the pointers of that layer's records are rewritten (`fz_plan_relocate`).
     The forward's inputs (latent tokens, timestep embedding) are relocated the same way.  Where the controller's planning never reads the
     forward's own activations (the built-in controllers: slots of the arena, constants, masks from the STORED maps -- `issue_events_first`)
     all its calls are taken first and the whole launch list goes out as ONE hipGraph launch (`fz_plan_graph_launch`: kernel nodes in
     record order, the relocated ones refreshed with hipGraphExecKernelNodeSetParams).

The buffers the records point into are (a) the recorded forward's own allocations -- on the GPU out of a private `torch.cuda.MemPool`
shared by all plans of a process (blocks freed during the forward are reused inside it exactly as the recording saw, and never handed to
anyone else); without one (the CPU emulation backend of the tests) the plan keeps every allocation of its forward alive -- and (b) the
process-wide scratch of the launch stream (split-K slabs, GroupNorm partials: `kernels.scratch_buffers`), which the plan holds references to;
a plan is keyed on the launch stream and on `kernels.scratch_generation()`: when the scratch was regrown or evicted since, the forward is
recorded again.  INVARIANT of the shared pool: persistent buffers (context K / V^T, weight packs, scratch) must exist before the recording --
the warm-up walk is there to create them -- because a block the pool hands out later can alias what an older plan's replay overwrites.
A torch *compute* op inside the recorded part of the forward would not be re-issued by a replay: a dispatch mode watches the recording and
such a forward is never replayed (`IssuePlans.stats["unrecordable"]` says which op).

Scope: one process, one GPU, the built-in controllers (`issue_signature()`); a frame-sharded forward (fatezero_amd.dist) and foreign
controllers run the Python walk as before.  Off by default: `UNetPseudo3DConditionModel.enable_issue_plans()` / FZ_ISSUE_PLANS=1.
"""
import ctypes as C
import os

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _native as N
from . import kernels as K

_TENSOR_FIELDS = ("p", "row_mask", "mapper_t", "coef", "cur_out", "capture_first")
_PARK_BASE = 0x7000_0000_0000_0000  # (far outside any address space a device or host allocation can have; fields 2^44 bytes apart)

# aten ops a recorded forward may run between its launches: allocation and metadata only
_ALLOC_OPS = {"aten.empty.memory_format", "aten.empty_like.default", "aten.empty_strided.default", "aten.new_empty.default",
              "aten.new_empty_strided.default"}
_VIEW_OPS = {"aten.view.default", "aten._unsafe_view.default", "aten.reshape.default", "aten._reshape_alias.default", "aten.as_strided.default",
             "aten.slice.Tensor", "aten.select.int", "aten.permute.default", "aten.transpose.int", "aten.t.default", "aten.expand.default",
             "aten.unsqueeze.default", "aten.squeeze.dim", "aten.squeeze.default", "aten.detach.default", "aten.alias.default",
             "aten.split.Tensor", "aten.split_with_sizes.default", "aten.unbind.int", "aten.narrow.default", "aten.unflatten.int",
             "aten.flatten.using_ints", "aten.chunk.default", "aten.view_as.default", "aten.lift_fresh.default", "aten.unfold.default",
             "aten.movedim.int"}


class _Watch(TorchDispatchMode):
    """Sees every aten op of the recorded forward: keeps allocations alive where no private pool does, names compute ops."""

    def __init__(self, rec):
        super().__init__()
        self.rec = rec

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        rec = self.rec
        if rec.live == 0:
            name = str(func)
            if name in _ALLOC_OPS:
                if rec.keep is not None:
                    rec.keep.append(out)
            elif name not in _VIEW_OPS:
                rec.foreign.append(name)  # a compute op (or a host read) of torch's own: a replay would not repeat it
        return out


class _Event:
    """One `controller.attention_plan(...)` call of the walk: its arguments, what it returned at recording time, and the records
    [first, first + count) issued between it and the next event (the only ones that can use the tensors it returned)."""
    __slots__ = ("args", "first", "count", "n_plain", "mode", "fields")

    def __init__(self, args, first, plan):
        self.args, self.first, self.count = args, first, 0
        self.n_plain, self.mode = plan.n_plain, plan.mode
        # field -> [pointer the records currently hold, bytes]
        self.fields = {f: [getattr(plan, f).data_ptr(), getattr(plan, f).numel() * getattr(plan, f).element_size()]
                       for f in _TENSOR_FIELDS if getattr(plan, f) is not None}


class _Recording:
    """State of the one forward being recorded (attention._plan_for reports the controller calls to it)."""

    def __init__(self, handle, keep):
        self.handle, self.keep = handle, keep
        self.events, self.foreign = [], []
        self.live = 0          # > 0: inside a controller call -- launches and torch ops are the controller's, not the plan's
        self.pool_ctx = None   # re-entered around the recorded stretches (GPU only)
        self.enter_pool = None

    def controller_call(self, planner, args):
        L = N.lib()
        if self.handle is None:  # the warm-up walk under the private pool: nothing is recorded, the controller still allocates outside it
            if self.pool_ctx is not None:
                self.pool_ctx.__exit__(None, None, None)
                self.pool_ctx = None
            try:
                return planner(*args)
            finally:
                if self.enter_pool is not None:
                    self.pool_ctx = self.enter_pool()
                    self.pool_ctx.__enter__()
        N.check(L.fz_plan_pause(self.handle, 1), "fz_plan_pause")
        if self.pool_ctx is not None:
            self.pool_ctx.__exit__(None, None, None)
            self.pool_ctx = None
        self.live += 1
        try:
            plan = planner(*args)
        finally:
            self.live -= 1
            if self.enter_pool is not None:
                self.pool_ctx = self.enter_pool()
                self.pool_ctx.__enter__()
            N.check(L.fz_plan_pause(self.handle, 0), "fz_plan_pause")
        self.events.append(_Event(args, int(L.fz_plan_launches(self.handle)), plan))
        return plan


class PlanMismatch(RuntimeError):
    """A replay met a forward its plan does not describe (another kind of controller step, a tensor of another size, an argument word
    that is not the pointer it was taken for)."""


_recording = None  # the forward being recorded (attention._plan_for looks here)
ENABLED = True     # same-box A/B runs (scripts/ab_bench.py fatezero_amd.issue ENABLED): False -> every forward is walked


def recording():
    return _recording


def _model_switches():
    """The module-level switches of the model code that change which kernels a forward launches (same-box A/B runs flip them at run time):
    a plan recorded under one setting is not the launch list of another."""
    from .video_diffusion.models import attention as A, lora as Lo, resnet as R
    return (A.LN_FUSION, A.QKV_FUSION, A.LN_FROM_PRODUCER, R.GN_FROM_EPILOGUE, Lo.LORA_PAIR_FUSION, Lo.LORA_PAIR_GN, Lo.LORA_PAIR_ALWAYS)


class ForwardPlan:
    """A recorded forward: the native plan, its events, the tensors it must keep alive and its output."""

    def __init__(self, handle, events, keep, out, inputs, n_launches, ctx_kv, scratch=()):
        self.handle, self.events, self.keep, self.out = handle, events, keep, out
        self.scratch = list(scratch)  # the stream's scratch tensors the records point into: alive as long as the plan is
        self.graph_ok = True          # False after a failed hipGraph build / launch: this plan goes out record by record from then on
        self.inputs = inputs  # name -> [pointer the records hold, bytes]
        self.n = n_launches
        # the text context the recorded forward attended to, and ITS K / V^T projections [(module, K, V^T)]: the records of the cross-attention
        # layers point at these buffers (the walk projects a context once per pass and caches the result on the module, attention.py)
        self.ctx_kv = ctx_kv
        self.replays = 0
        self._lib = N.lib()  # the library that made the handle destroys it

    def __del__(self):
        try:
            if self.handle is not None:
                self._lib.fz_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _relocate(self, first, count, slot, tensor, what):
        ptr = tensor.data_ptr()
        if ptr == slot[0]:
            return
        nbytes = tensor.numel() * tensor.element_size()
        if nbytes != slot[1]:
            raise PlanMismatch(f"issue plan: {what} changed size ({slot[1]} -> {nbytes} bytes): the plan does not describe this forward")
        self._move(first, count, slot, nbytes, ptr, what)

    def _move(self, first, count, slot, nbytes, new_ptr, what):
        """Relocation matches argument words by VALUE; the number of words a slot's range matched when the forward was recorded is the number
        it must match at every replay (every other argument byte is the same as then): anything else is a word that is not this pointer."""
        n = N.lib().fz_plan_relocate(self.handle, first, count, C.c_void_p(slot[0]), nbytes, C.c_void_p(new_ptr))
        if n < 0:
            raise RuntimeError("fz_plan_relocate failed")
        if len(slot) > 2 and n != slot[2]:
            raise PlanMismatch(f"issue plan: {what} matched {n} argument words where the recording had {slot[2]}: a non-pointer argument holds "
                               "a value inside the tensor's address range -- this forward cannot be replayed safely")
        slot[0] = new_ptr

    def _relocate_many(self, first, count, pairs, what):
        """Several pointers of one record range at once.  Done one after the other, a tensor that landed where ANOTHER field's previous tensor
        lived would be moved twice (a -> b, then every b -> c): the changed ones go through distinct addresses no allocation can have first."""
        changed = [(slot, t) for slot, t in pairs if t.data_ptr() != slot[0]]
        if len(changed) <= 1:
            for slot, t in changed:
                self._relocate(first, count, slot, t, what)
            return
        for i, (slot, t) in enumerate(changed):
            nbytes = t.numel() * t.element_size()
            if nbytes != slot[1]:
                raise PlanMismatch(f"issue plan: {what} changed size ({slot[1]} -> {nbytes} bytes): the plan does not describe this forward")
            self._move(first, count, slot, nbytes, _PARK_BASE + (i << 44), what)
        for slot, t in changed:
            self._relocate(first, count, slot, t, what)

    def bind_context(self, ctx):
        """A new text context of the recorded shape (the next job, the other pass): its K / V^T into the buffers the records point at."""
        for m, kk, vt, kvp in self.ctx_kv:
            m.project_context_into(ctx, kk, vt, kvp)

    def context_is_bound(self, ctx):
        """Do the buffers the records point at hold the projections of `ctx`?  The modules' caches say: whoever writes those buffers
        (bind_context of any plan that shares them) or replaces them (a walked forward with another context) updates the cache."""
        for m, kk, *_ in (self.ctx_kv[0], self.ctx_kv[-1]) if self.ctx_kv else ():
            c = m._ctx_kv
            if c is None or c[0] is not ctx or c[1] != ctx._version or c[2] is not kk:
                return False
        return True

    def replay(self, x, temb_act, controller, events_first=False, graph=False):
        """events_first: every controller call is taken before the first launch (a controller whose planning never reads the forward's own
        activations says so: `issue_events_first`) and the records go out in one piece -- as ONE hipGraph launch with `graph`."""
        L = N.lib()
        h, stream = self.handle, K._stream(x.data)
        head = self.events[0].first if self.events else self.n  # conv_in and the time-embedding projections: in front of the first attention
        self._relocate_many(0, head, [(self.inputs["x"], x.data), (self.inputs["temb"], temb_act)], "a forward input")
        planner = None if controller is None else controller.attention_plan
        at = 0
        alive = []  # what the controller handed out stays allocated until the launches that read it are queued
        for ev in self.events:
            if ev.first > at and not events_first:
                rc = L.fz_plan_replay(h, at, ev.first - at, stream)
                if rc:
                    N.check(rc, "fz_plan_replay")
                at = ev.first
            plan = planner(*ev.args)
            alive.append(plan)
            if plan.n_plain != ev.n_plain or plan.mode != ev.mode:
                raise PlanMismatch("issue plan: the controller answered a different attention plan than the recorded forward "
                                   f"(mode {ev.mode} -> {plan.mode}, plain frames {ev.n_plain} -> {plan.n_plain}): issue_signature() "
                                   "does not separate the two kinds of step")
            pairs = []
            for f in _TENSOR_FIELDS:
                t = getattr(plan, f)
                slot = ev.fields.get(f)
                if (t is None) != (slot is None):
                    raise PlanMismatch(f"issue plan: AttnPlan.{f} is {'absent' if t is None else 'present'} where the recorded forward "
                                       "had the opposite: issue_signature() does not separate the two kinds of step")
                if t is not None:
                    pairs.append((slot, t))
            if pairs:
                self._relocate_many(ev.first, ev.count, pairs, "an AttnPlan tensor")
        if graph and at == 0 and self.graph_ok:
            rc = L.fz_plan_graph_launch(h, stream)
            if rc:  # graph creation / instantiation / node update failed: nothing of this forward was issued -- the records still can be
                self.graph_ok = False
                rc = L.fz_plan_replay(h, 0, self.n, stream)
                if rc:
                    N.check(rc, "fz_plan_replay")
        elif self.n > at:
            rc = L.fz_plan_replay(h, at, self.n - at, stream)
            if rc:
                N.check(rc, "fz_plan_replay")
        del alive
        K._launches[0] += self.n
        self.replays += 1
        return self.out.like(self.out.data.clone())  # (the caller may keep a forward's result across the next forward, as after the walk)


class IssuePlans:
    """Per-UNet cache of recorded forwards.  `run` returns the forward's output tokens, or None: take the Python walk."""
    MAX_PLANS = 8
    _pool = None  # torch.cuda.MemPool shared by every plan of the process

    def __init__(self, unet, graph=None):
        self.unet = unet
        # one hipGraph launch per forward where the controller's calls can all be taken first (FZ_ISSUE_GRAPH=0: per-record launches)
        self.graph = (os.environ.get("FZ_ISSUE_GRAPH", "1") != "0") if graph is None else bool(graph)
        self.seen = {}
        self.plans = {}
        self._probe = None
        self.stats = {"walked": 0, "recorded": 0, "replayed": 0, "contexts_bound": 0, "unsupported": 0, "unrecordable": []}

    # -- what kind of forward is this ----------------------------------------------------------------------
    def _controller(self):
        if self._probe is None:
            from .video_diffusion.models.attention import CrossAttention
            self._probe = next(m for m in self.unet.modules() if isinstance(m, CrossAttention))
        return self._probe.controller

    def clear(self):
        self.seen.clear()
        self.plans.clear()

    def _key(self, x, temb_act, ctx, controller):
        from . import dist as D
        if D.active_shard() is not None or x.data.dtype != torch.float16 or not x.data.is_contiguous():
            return None
        if controller is None:
            sig = ("none",)
        else:
            sig_fn = getattr(controller, "issue_signature", None)
            sig = None if sig_fn is None or getattr(controller, "attention_plan", None) is None else sig_fn()
            if sig is None:
                return None
        return (tuple(x.data.shape), x.b, x.f, x.h, x.w, str(x.data.device), K._scratch_key(x.data)[1], tuple(temb_act.shape), tuple(ctx.shape),
                ctx.dtype, type(controller), sig, _model_switches())

    # -- record --------------------------------------------------------------------------------------------
    @classmethod
    def _pool_entry(cls, device):
        if torch.device(device).type != "cuda":
            return None
        if cls._pool is None:
            cls._pool = torch.cuda.MemPool()
        pool = cls._pool
        return lambda: torch.cuda.use_mem_pool(pool, device=device)

    def _record(self, key, x, temb_act, ctx, controller):
        global _recording
        L = N.lib()
        handle = C.c_void_p()
        N.check(L.fz_plan_begin(C.byref(handle)), "fz_plan_begin")
        enter_pool = self._pool_entry(x.data.device)
        rec = _Recording(handle, None if enter_pool is not None else [])
        rec.enter_pool = enter_pool
        _recording = rec
        try:
            with _Watch(rec):
                if enter_pool is not None:
                    rec.pool_ctx = enter_pool()
                    rec.pool_ctx.__enter__()
                try:
                    out = self.unet._forward_body(x, temb_act, ctx)
                finally:
                    if rec.pool_ctx is not None:
                        rec.pool_ctx.__exit__(None, None, None)
                        rec.pool_ctx = None
        except BaseException:
            _recording = None
            L.fz_plan_end(handle)
            L.fz_plan_destroy(handle)
            raise
        _recording = None
        N.check(L.fz_plan_end(handle), "fz_plan_end")
        n = int(L.fz_plan_launches(handle))
        for i, ev in enumerate(rec.events):
            ev.count = (rec.events[i + 1].first if i + 1 < len(rec.events) else n) - ev.first
        if rec.foreign:
            self.stats["unrecordable"].append(sorted(set(rec.foreign)))
            self.plans[key] = None  # this kind of forward is walked from now on
            L.fz_plan_destroy(handle)
            return out
        result = out.like(out.data.clone())  # the plan keeps `out`: every replay writes it
        inputs = {"x": [x.data.data_ptr(), x.data.numel() * x.data.element_size()],
                  "temb": [temb_act.data_ptr(), temb_act.numel() * temb_act.element_size()]}
        from .video_diffusion.models.attention import CrossAttention
        ctx_kv = []
        for m in self.unet.modules():
            if isinstance(m, CrossAttention) and m._ctx_kv is not None:
                if m._ctx_kv[0] is not ctx:  # (cannot happen: the recorded forward just attended to ctx)
                    raise RuntimeError("issue plan: a cross-attention layer holds the projections of another context")
                ctx_kv.append((m, m._ctx_kv[2], m._ctx_kv[3], m._ctx_kv[4]))
        if len(self.plans) >= self.MAX_PLANS:
            self.plans.pop(next(iter(self.plans)))
        keep = rec.keep
        if keep is not None:
            keep.append(x.data)
            keep.append(temb_act)
        plan = self.plans[key] = ForwardPlan(handle, rec.events, keep, out, inputs, n, ctx_kv, scratch=K.scratch_buffers(x.data))
        plan.scratch_gen = K.scratch_generation()
        # how many argument words each relocatable range matches now (a move onto itself counts without changing anything): the check of _move
        head = rec.events[0].first if rec.events else n
        for slot in inputs.values():
            slot.append(int(L.fz_plan_relocate(handle, 0, head, C.c_void_p(slot[0]), slot[1], C.c_void_p(slot[0]))))
        for ev in rec.events:
            for slot in ev.fields.values():
                slot.append(int(L.fz_plan_relocate(handle, ev.first, ev.count, C.c_void_p(slot[0]), slot[1], C.c_void_p(slot[0]))))
        self.stats["recorded"] += 1
        return result

    # -- entry ---------------------------------------------------------------------------------------------
    def run(self, x, temb_act, ctx):
        if not ENABLED:
            return None
        controller = self._controller()
        key = self._key(x, temb_act, ctx, controller)
        if key is None:
            self.stats["unsupported"] += 1
            return None
        if key in self.plans:
            plan = self.plans[key]
            if plan is None:
                self.stats["walked"] += 1
                return None
            if plan.scratch_gen != K.scratch_generation():
                # the stream's scratch was regrown or released since the recording (a later job with more frames, another stream's scratch
                # evicting this one's): the records point at the old buffers -- which the plan kept alive, but nobody else writes them any
                # more in the order the walk would.  Drop the plan; this forward is walked (warm-up) and the next one recorded afresh.
                del self.plans[key]
                self.seen[key] = 1
                self.stats["rerecorded"] = self.stats.get("rerecorded", 0) + 1
                self.stats["walked"] += 1
                return self._walk_in_pool(x, temb_act, ctx)
            first = controller is None or bool(getattr(controller, "issue_events_first", False))
            self.stats["replayed"] += 1
            if not plan.context_is_bound(ctx):
                plan.bind_context(ctx)
                self.stats["contexts_bound"] += 1
            try:
                return plan.replay(x, temb_act, controller, events_first=first, graph=first and self.graph)
            except PlanMismatch:
                # the controller answered another kind of step than the recorded one (its issue_signature() is incomplete) or a relocation
                # did not match: the controller's state has advanced, so THIS forward cannot be walked instead -- the error goes up, but the
                # kind is walked from now on: a caller that restarts the job does not meet the plan again
                self.plans[key] = None
                raise
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if len(self.seen) > 64:
            self.seen.pop(next(iter(self.seen)))
        if n == 0:
            self.stats["walked"] += 1
            return self._walk_in_pool(x, temb_act, ctx)  # the first forward of a kind warms the caches the walk fills lazily
        return self._record(key, x, temb_act, ctx, controller)

    def _walk_in_pool(self, x, temb_act, ctx):
        """The warm-up walk allocates from the private pool as the recording will: the recorded forward then finds the pool in its steady
        state (freed blocks of the right sizes to reuse) instead of growing it allocation by allocation -- a compact set of addresses."""
        global _recording
        enter_pool = self._pool_entry(x.data.device)
        if enter_pool is None:
            return None
        rec = _Recording(None, None)
        rec.enter_pool = enter_pool
        _recording = rec
        rec.pool_ctx = enter_pool()
        rec.pool_ctx.__enter__()
        try:
            out = self.unet._forward_body(x, temb_act, ctx)
        finally:
            _recording = None
            if rec.pool_ctx is not None:
                rec.pool_ctx.__exit__(None, None, None)
        return out.like(out.data.clone())  # (out of the general allocator: pool blocks stay with the plans)
