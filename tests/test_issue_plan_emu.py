"""Native issue plans (csrc/plan.hip, fatezero_amd/issue.py) on the CPU emulation backend: the record / relocate / replay machinery of the
library, and the pipeline with the UNet forwards of the steady-state steps replayed from recorded plans against the same pipeline walked in
Python -- same kernels, same arguments, so bit for bit.  (The GPU leg with the private memory pool: tests/test_pipeline_gpu.py.)"""
import ctypes as C

import pytest
import torch

from fatezero_amd import _native, build

import pipeline_cases as PC


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())
    yield
    _native.reset_backend()


def test_record_relocate_replay():
    from fatezero_amd import kernels as K
    L = _native.lib()
    g = torch.Generator().manual_seed(3)
    x1, x2 = (torch.randn(96, 320, generator=g).half() for _ in range(2))
    gam, bet = torch.randn(320, generator=g).half(), torch.randn(320, generator=g).half()
    want1, want2 = K.layernorm(x1, gam, bet, eps=1e-5), K.layernorm(x2, gam, bet, eps=1e-5)
    eye = torch.eye(320).half()   # (what a record points at must outlive the plan: the plan owns host memory only)
    h = C.c_void_p()
    assert L.fz_plan_begin(C.byref(h)) == 0
    h2 = C.c_void_p()
    assert L.fz_plan_begin(C.byref(h2)) < 0                      # one recording at a time
    y = K.layernorm(x1, gam, bet, eps=1e-5)                      # recorded (and run)
    assert L.fz_plan_pause(h, 1) == 0
    K.layernorm(x2, gam, bet, eps=1e-5)                          # live only: not part of the plan
    assert L.fz_plan_pause(h, 0) == 0
    z = K.gemm(y, eye)                         # a second record that reads the first one's output
    assert L.fz_plan_end(h) == 0
    n = L.fz_plan_launches(h)
    assert n == 2, n
    assert torch.equal(y, want1)
    zwant2 = K.gemm(want2, eye)
    # replay as recorded: same buffers, same result
    y.zero_(); z.zero_()
    assert L.fz_plan_replay(h, 0, n, None) == 0
    assert torch.equal(y, want1)
    # the input lives elsewhere now: one pointer of record 0 moves, nothing of record 1
    nbytes = x1.numel() * 2
    assert L.fz_plan_relocate(h, 1, 1, C.c_void_p(x1.data_ptr()), nbytes, C.c_void_p(x2.data_ptr())) == 0
    assert L.fz_plan_relocate(h, 0, n, C.c_void_p(x1.data_ptr()), nbytes, C.c_void_p(x2.data_ptr())) == 1
    assert L.fz_plan_replay(h, 0, n, None) == 0
    assert torch.equal(y, want2) and torch.equal(z, zwant2)
    # a pointer INTO the old range keeps its offset: rows 32.. of x1 as the input of a one-record plan
    h3 = C.c_void_p()
    assert L.fz_plan_begin(C.byref(h3)) == 0
    y3 = K.layernorm(x1[32:], gam, bet, eps=1e-5)
    assert L.fz_plan_end(h3) == 0
    assert L.fz_plan_relocate(h3, 0, 1, C.c_void_p(x1.data_ptr()), nbytes, C.c_void_p(x2.data_ptr())) == 1
    assert L.fz_plan_replay(h3, 0, 1, None) == 0
    assert torch.equal(y3, want2[32:])
    assert L.fz_plan_replay(h, 0, n + 1, None) < 0 and L.fz_plan_relocate(h, 1, n, None, 8, None) < 0   # ranges are checked
    L.fz_plan_destroy(h); L.fz_plan_destroy(h3)


def test_the_recorder_is_per_thread():
    """include/fatezero_hip.h: no process-wide state -- a recording sees the launches of the thread that began it; another thread can launch (unrecorded)
    and record a plan of its own at the same time."""
    import threading
    from fatezero_amd import kernels as K
    L = _native.lib()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(64, 320, generator=g).half()
    gam, bet = torch.randn(320, generator=g).half(), torch.randn(320, generator=g).half()
    h = C.c_void_p()
    assert L.fz_plan_begin(C.byref(h)) == 0
    K.layernorm(x, gam, bet, eps=1e-5)
    seen = {}

    def other():
        K.layernorm(x, gam, bet, eps=1e-5)                       # issued, not recorded into the first thread's plan
        seen["outer"] = L.fz_plan_launches(h)
        h2 = C.c_void_p()
        seen["begin"] = L.fz_plan_begin(C.byref(h2))             # this thread has no recording yet: allowed
        K.layernorm(x, gam, bet, eps=1e-5)
        K.layernorm(x, gam, bet, eps=1e-5)
        seen["end"] = L.fz_plan_end(h2)
        seen["own"] = L.fz_plan_launches(h2)
        L.fz_plan_destroy(h2)

    t = threading.Thread(target=other)
    t.start(); t.join()
    K.layernorm(x, gam, bet, eps=1e-5)
    assert L.fz_plan_end(h) == 0
    assert seen == {"outer": 1, "begin": 0, "end": 0, "own": 2}, seen
    assert L.fz_plan_launches(h) == 2
    L.fz_plan_destroy(h)


@pytest.mark.parametrize("name", ["pipe_f3_mid_next"])   # (pipe_small_refine_reweight: with the spill tier under it, below; more scenarios on MI355X)
def test_pipeline_replayed_from_plans_is_bit_identical(name, monkeypatch):
    base, pipe0 = PC.run_pipeline_case(name, "cpu", return_pipe=True)
    monkeypatch.setenv("FZ_ISSUE_PLANS", "1")
    res, pipe = PC.run_pipeline_case(name, "cpu", return_pipe=True)
    stats = pipe.unet._issuer.stats
    print(name, stats)
    assert stats["replayed"] >= 3 and stats["recorded"] >= 1 and not stats["unrecordable"] and not stats["unsupported"], stats
    assert torch.equal(pipe.last_edited_latents, pipe0.last_edited_latents)
    assert res == base
    for st, st0 in zip(pipe.store_controller.attention_store_all_step, pipe0.store_controller.attention_store_all_step):
        for k in st0:
            for a, b in zip(st[k], st0[k]):
                assert torch.equal(a, b), k


def test_plans_over_the_spill_tier(monkeypatch):
    """Replayed capture launches follow the step's slab into the staging ring of the arena's spill tier (disk_store=True): the pointers of
    every capture layer are relocated at every step."""
    from fatezero_amd.video_diffusion.prompt_attention import attention_store as AS
    base, pipe0 = PC.run_pipeline_case("pipe_small_refine_reweight", "cpu", return_pipe=True)
    monkeypatch.setenv("FZ_ISSUE_PLANS", "1")
    monkeypatch.setenv("FZ_ARENA_HBM_GB", "0")
    monkeypatch.setattr(AS, "SPILL_RING", 2)
    res, pipe = PC.run_pipeline_case("pipe_small_refine_reweight", "cpu", return_pipe=True, disk_store=True)
    assert pipe.unet._issuer.stats["replayed"] >= 3 and pipe.store_controller.arena.fetch_stats["h2d"] >= 1
    assert torch.equal(pipe.last_edited_latents, pipe0.last_edited_latents) and res == base


def test_a_forward_with_a_torch_compute_op_is_never_replayed(monkeypatch):
    """A torch op of its own inside the recorded stretch would not be repeated by a replay: the recording notices and that kind of forward
    stays on the Python walk."""
    from fatezero_amd.video_diffusion.models import unet_3d_condition as U
    unet = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    unet.enable_issue_plans()
    orig = U.UNetPseudo3DConditionModel._forward_body

    def body(self, x, temb_act, ctx):
        y = orig(self, x, temb_act, ctx)
        return y.like(y.data * 1.0)   # aten.mul: not one of the library's launches
    monkeypatch.setattr(U.UNetPseudo3DConditionModel, "_forward_body", body)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 4, 2, 16, 16, generator=g).half()
    ctx = torch.randn(1, 77, 64, generator=g).half()
    outs = [unet(z, 10 * i + 1, ctx).sample for i in range(4)]
    st = unet._issuer.stats
    assert st["unrecordable"] == [["aten.mul.Tensor"]] and st["replayed"] == 0 and st["walked"] == 3, st
    assert all(torch.isfinite(o).all() for o in outs)


def test_replay_refuses_a_step_of_another_kind():
    """Two kinds of step that a controller's issue_signature() fails to tell apart: the replay compares what the controller answers with
    what the recording saw and stops instead of issuing the wrong launch list."""
    from fatezero_amd.video_diffusion.prompt_attention.attention_register import register_attention_control, DummyController
    from fatezero_amd.video_diffusion.models.attention import AttnPlan
    from fatezero_amd import kernels as K
    from types import SimpleNamespace

    class Flip(DummyController):
        capture = False

        def attention_plan(self, is_cross, place, n_frames, clip_len, heads, lq, lk, device):
            if self.capture and is_cross:
                return AttnPlan(0, K.FZ_ATTN_CAPTURE, p=torch.empty(n_frames, heads, lq, K.CROSS_P_STRIDE, dtype=torch.float16))
            return AttnPlan(n_frames)
    unet = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    unet.enable_issue_plans()
    ctrl = Flip()
    register_attention_control(SimpleNamespace(unet=unet), ctrl)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 4, 2, 16, 16, generator=g).half()
    ctx = torch.randn(1, 77, 64, generator=g).half()
    ys = [unet(z, 5, ctx).sample for _ in range(3)]
    assert unet._issuer.stats["replayed"] == 1 and torch.equal(ys[0], ys[2]) and torch.equal(ys[1], ys[2])
    ctrl.capture = True
    with pytest.raises(RuntimeError, match="issue_signature"):
        unet(z, 5, ctx)
    # the error went up (the controller's state had advanced: this forward could not be walked instead), but that kind of forward is walked
    # from now on: a caller that starts over does not meet the plan again
    walked = unet._issuer.stats["walked"]
    y = unet(z, 5, ctx).sample
    assert unet._issuer.stats["walked"] == walked + 1 and torch.isfinite(y.float()).all()


def test_plans_serve_the_next_job():
    """The plans of one job replay the next one: a new text context of the same shape is projected into the K / V^T buffers the records point
    at, a new controller object answers the recorded events -- against a walked run of the same two jobs, bit for bit."""
    from types import SimpleNamespace
    from fatezero_amd.video_diffusion.prompt_attention.attention_register import register_attention_control
    from fatezero_amd.video_diffusion.prompt_attention.attention_store import AttentionStore
    g = torch.Generator().manual_seed(4)
    z = [torch.randn(1, 4, 2, 16, 16, generator=g).half() for _ in range(2)]
    ctxs = [torch.randn(1, 77, 64, generator=g).half() for _ in range(2)]

    def jobs(plans):
        unet = PC.build_unet("tiny16", {"lora": 16}, "cpu")
        if plans:
            unet.enable_issue_plans()
        outs, maps = [], []
        for j in range(2):
            store = AttentionStore()
            store.LOW_RESOURCE = True
            register_attention_control(SimpleNamespace(unet=unet), store)
            x = z[j]
            for i in range(4):
                x = unet(x, 900 - 200 * i, ctxs[j]).sample
                store.step_callback(x)
                outs.append(x.clone())
            maps.append([[m.clone() for k in sorted(st) for m in st[k]] for st in store.attention_store_all_step])
        return unet, outs, maps
    _, outs0, maps0 = jobs(False)
    unet, outs1, maps1 = jobs(True)
    st = unet._issuer.stats
    assert st["walked"] == 1 and st["recorded"] == 1 and st["replayed"] == 6 and st["contexts_bound"] == 1, st
    assert all(torch.equal(a, b) for a, b in zip(outs0, outs1))
    for ja, jb in zip(maps0, maps1):
        for sa, sb in zip(ja, jb):
            assert len(sa) == len(sb) and all(torch.equal(a, b) for a, b in zip(sa, sb))


def test_timestep_as_kernel_argument_matches_the_tensor_path(monkeypatch):
    """A host timestep enters the time embedding as a kernel ARGUMENT (freq * float) and the result is kept per timestep -- no host-to-device
    copy, i.e. no stream synchronisation, in front of a forward (-1.6 % on the judged job, profiles/r05_issue_plans_job_ab.txt).  Same
    arithmetic as the tensor path (a device tensor t times freq): bit for bit."""
    from fatezero_amd.video_diffusion.models import unet_3d_condition as U
    unet = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    a = unet.time_embed(481, 2, torch.device("cpu"))
    assert unet.time_embed(481, 2, torch.device("cpu")) is a                      # kept
    assert unet.time_embed(torch.tensor(481), 2, torch.device("cpu")) is a       # a 0-dim host tensor is a host scalar too
    assert unet.time_embed(461, 2, torch.device("cpu")) is not a
    monkeypatch.setattr(U, "TIME_EMBED_CACHE", False)
    b = unet.time_embed(481, 2, torch.device("cpu"))
    assert b is not a and torch.equal(a, b)
    c = unet.time_embed(torch.tensor([481, 461]), 2, torch.device("cpu"))         # one timestep per batch entry: the tensor path
    assert torch.equal(c[0], a[0]) and not torch.equal(c[1], a[1])
    unet.load_state_dict(unet.state_dict())                                       # new weights: nothing kept
    assert not unet._temb_cache


def test_relocations_that_chain_are_not_applied_twice():
    """Two pointers of one record range change at once and the first one's NEW address is the second one's OLD address (the allocator handed a
    freed buffer of one field to another): moved one after the other the first would travel twice."""
    from fatezero_amd import kernels as K
    from fatezero_amd.issue import ForwardPlan
    L = _native.lib()
    g = torch.Generator().manual_seed(8)
    pool = torch.randn(3, 64, 320, generator=g).half()      # three equally sized buffers at known addresses
    xa, xb, xc = pool[0], pool[1], pool[2]
    gam, bet = torch.ones(320).half(), torch.zeros(320).half()
    h = C.c_void_p()
    assert L.fz_plan_begin(C.byref(h)) == 0
    ya = K.layernorm(xa, gam, bet, eps=1e-5)                 # record 0 reads xa
    yb = K.layernorm(xb, gam, bet, eps=1e-5)                 # record 1 reads xb
    assert L.fz_plan_end(h) == 0
    plan = ForwardPlan(h, [], None, None, {}, 2, [])
    nbytes = xa.numel() * 2
    slots = ([xa.data_ptr(), nbytes], [xb.data_ptr(), nbytes])
    plan._relocate_many(0, 2, [(slots[0], xb), (slots[1], xc)], "test")   # xa -> xb while xb -> xc
    assert L.fz_plan_replay(h, 0, 2, None) == 0
    assert torch.equal(ya, K.layernorm(xb, gam, bet, eps=1e-5)) and torch.equal(yb, K.layernorm(xc, gam, bet, eps=1e-5))
    assert slots[0][0] == xb.data_ptr() and slots[1][0] == xc.data_ptr()


def test_a_relocation_that_matches_a_non_pointer_word_is_refused():
    """fz_plan_relocate matches argument words by value.  The host layer counts, when a forward is recorded, how many words each relocatable
    range matches, and a replay whose relocation matches another number stops: here an int64 row count (96) sits inside the 'address range'
    [64, 128) of a fake tensor."""
    from fatezero_amd import kernels as K
    from fatezero_amd.issue import ForwardPlan
    L = _native.lib()
    x = torch.randn(96, 320).half()
    gam, bet = torch.ones(320).half(), torch.zeros(320).half()
    h = C.c_void_p()
    assert L.fz_plan_begin(C.byref(h)) == 0
    K.layernorm(x, gam, bet, eps=1e-5)
    assert L.fz_plan_end(h) == 0
    plan = ForwardPlan(h, [], None, None, {}, 1, [])
    good = [x.data_ptr(), x.numel() * 2, 1]                  # the input pointer: one word, as recorded
    plan._move(0, 1, good, good[1], x.data_ptr(), "x")       # (onto itself: counts, changes nothing)
    fake = [64, 64, 0]                                        # a slot whose recorded range [64, 128) matched nothing ...
    with pytest.raises(RuntimeError, match="matched 1 argument words where the recording had 0"):
        plan._move(0, 1, fake, 64, 1 << 40, "a fake tensor")  # ... and now matches the int64 `rows` = 96 of the launch


def test_plans_follow_weights_and_model_switches(monkeypatch):
    """New weights drop every plan (the records point at the packed weights of the old ones); a launch-list switch of the model code flipped
    at run time selects other plans instead of replaying the wrong list."""
    from fatezero_amd.video_diffusion.models import attention as A
    g = torch.Generator().manual_seed(2)
    z = torch.randn(1, 4, 2, 16, 16, generator=g).half()
    ctx = torch.randn(1, 77, 64, generator=g).half()
    unet = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    ref = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    unet.enable_issue_plans()
    for _ in range(3):
        y = unet(z, 7, ctx).sample
    assert unet._issuer.stats["replayed"] == 1 and torch.equal(y, ref(z, 7, ctx).sample)
    sd = {k: (v * 1.25 if k.endswith("conv_in.weight") else v) for k, v in unet.state_dict().items()}
    unet.load_state_dict(sd)
    ref.load_state_dict(sd)
    assert not unet._issuer.plans and not unet._issuer.seen
    y2 = [unet(z, 7, ctx).sample for _ in range(3)][-1]                # walked, recorded, replayed -- on the new weights
    assert unet._issuer.stats["replayed"] == 2 and torch.equal(y2, ref(z, 7, ctx).sample) and not torch.equal(y2, y)
    monkeypatch.setattr(A, "QKV_FUSION", not A.QKV_FUSION)             # another launch list: its own walk / record / replay
    walked = unet._issuer.stats["walked"]
    y3 = [unet(z, 7, ctx).sample for _ in range(3)][-1]
    assert unet._issuer.stats["walked"] == walked + 1 and unet._issuer.stats["replayed"] == 3 and len(unet._issuer.plans) == 2
    assert torch.equal(y3, ref(z, 7, ctx).sample)


def test_a_plan_does_not_outlive_the_scratch_it_points_at():
    """Round-5 advisor: a plan's records hold raw pointers into the process-wide scratch of the launch stream (split-K slabs, GroupNorm
    partials).  The plan keeps those tensors alive, and a scratch that was regrown or evicted since the recording (kernels.scratch_generation)
    makes the forward be walked and recorded again instead of replayed against the old buffers."""
    from fatezero_amd import kernels as K
    g = torch.Generator().manual_seed(6)
    z = torch.randn(1, 4, 2, 16, 16, generator=g).half()
    ctx = torch.randn(1, 77, 64, generator=g).half()
    unet = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    ref = PC.build_unet("tiny16", {"lora": 16}, "cpu")
    unet.enable_issue_plans()
    want = ref(z, 7, ctx).sample
    for _ in range(3):
        y = unet(z, 7, ctx).sample
    st = unet._issuer.stats
    assert st["replayed"] == 1 and torch.equal(y, want)
    plan = next(iter(unet._issuer.plans.values()))
    held = {t.data_ptr() for t in plan.scratch}
    assert held and held == {t.data_ptr() for t in K.scratch_buffers(z)}     # the stream's scratch, referenced by the plan
    # the scratch goes away under the plan (what a larger job's regrowth or the eviction of a fifth stream's scratch set does)
    gen = K.scratch_generation()
    K.release_scratch()
    assert K.scratch_generation() != gen and not K.scratch_buffers(z)
    assert {t.data_ptr() for t in plan.scratch} == held                      # ... but not under the plan's feet
    ys = [unet(z, 7, ctx).sample for _ in range(3)]                           # walked again, recorded again, replayed
    assert st.get("rerecorded") == 1 and st["recorded"] == 2 and st["replayed"] == 2
    assert all(torch.equal(v, want) for v in ys)
