"""The spill tier of the map arena (attention_store.MapArena; the reference's disk_store=True, attention_store.py:103-108) on its own: a store
fed with synthetic maps, no UNet -- residency by budget, the staging ring wrapping during capture, and maps_of_step under access orders the
edit never produces (random, repeated, forward): whatever the order, every step reads back exactly what was captured."""
import random

import pytest
import torch

from fatezero_amd import _native, build
from fatezero_amd.video_diffusion.prompt_attention import attention_store as AS


@pytest.fixture(scope="module", autouse=True)
def emu_backend():
    _native.use_test_backend(build.build_emu())   # (between_steps accumulates the running cross-map sum with fz_accumulate)
    yield
    _native.reset_backend()


LAYERS = [("down_cross", 2, 4, 64, 77, True), ("down_self", 2, 4, 64, 128, False), ("mid_cross", 2, 4, 16, 77, True), ("up_self", 2, 4, 64, 128, False)]


def capture(store, steps, seed=0):
    g = torch.Generator().manual_seed(seed)
    want = []
    for s in range(steps):
        step = []
        for key, f, h, lq, lk, is_cross in LAYERS:
            cm = store.new_slot(key, f, h, lq, lk, is_cross, torch.device("cpu"))
            cm.storage.copy_(torch.randn(cm.storage.shape, generator=g).half())
            step.append((key, cm.storage.clone()))
        want.append(step)
        store.between_steps()
    return want


def check_step(store, s, want):
    maps = store.maps_of_step(s)
    seen = {k: 0 for k in AS.KEYS}
    for key, ref in want[s]:
        cm = maps[key][seen[key]]
        seen[key] += 1
        assert torch.equal(cm.storage, ref), (s, key)
        assert cm.view.shape[-1] == (77 if key.endswith("cross") else ref.shape[-1])


@pytest.mark.parametrize("ring", [2, 3])
def test_every_access_order_reads_back_what_was_captured(ring, monkeypatch):
    monkeypatch.setattr(AS, "SPILL_RING", ring)
    store = AS.AttentionStore(disk_store=True, hbm_budget_bytes=0)     # nothing but the first step stays resident
    store.expected_steps = 9
    want = capture(store, 9)
    arena = store.arena
    assert sorted(arena.spilled) == list(range(1, 9)) and len(arena.ring) == ring
    assert arena.spilled_bytes == 8 * arena.step_bytes
    for s in reversed(range(9)):                                       # the edit's order (use_inversion_attention): newest first
        check_step(store, s, want)
    h2d_edit = arena.fetch_stats["h2d"]
    assert h2d_edit <= 8 - ring + 2, arena.fetch_stats                 # the steps still in the ring cost no copy, every other one exactly one
    for s in range(9):                                                 # forward
        check_step(store, s, want)
    rnd = random.Random(5)
    for _ in range(60):                                                # random, with repeats
        check_step(store, rnd.randrange(9), want)
    # the reference-shaped views of attention_store_all_step: host copies for the spilled steps
    for s in range(9):
        flat = [v for k in AS.KEYS for v in store.attention_store_all_step[s][k]]
        refs = {k: [r for kk, r in want[s] if kk == k] for k in AS.KEYS}
        assert sum(len(v) for v in refs.values()) == len(flat)
        for k in AS.KEYS:
            for v, r in zip(store.attention_store_all_step[s][k], refs[k]):
                assert torch.equal(v, r[..., : v.shape[-1]])


def test_budget_decides_how_many_steps_stay_resident():
    probe = AS.AttentionStore()
    capture(probe, 1)
    step_bytes = probe.arena.step_bytes
    store = AS.AttentionStore(disk_store=True, hbm_budget_bytes=step_bytes * (1 + 3 + AS.SPILL_RING))   # first step + 3 more + the ring
    store.expected_steps = 8
    want = capture(store, 8)
    assert sorted(store.arena.spilled) == [4, 5, 6, 7], sorted(store.arena.spilled)
    for s in (7, 3, 0, 5, 4, 6, 1, 2):
        check_step(store, s, want)
    plain = AS.AttentionStore(disk_store=False, hbm_budget_bytes=0)   # without disk_store the budget is not consulted: everything resident
    plain.expected_steps = 4
    w2 = capture(plain, 4)
    assert not plain.arena.spilled
    for s in range(4):
        check_step(plain, s, w2)


def test_a_recycled_larger_block_does_not_raise_residency():
    """Round-5 review: a 12-step arena released into the pool, then a budgeted store -- the pooled 11-step block must neither be handed out
    beyond the budget nor be held unaccounted."""
    big = AS.AttentionStore()
    big.expected_steps = 12
    capture(big, 12)
    step_bytes = big.arena.step_bytes
    big.release_arena()
    assert [t.numel() for t in AS.MapArena._pool] == [11 * step_bytes]
    budget = step_bytes * (1 + 3 + AS.SPILL_RING)
    store = AS.AttentionStore(disk_store=True, hbm_budget_bytes=budget)
    store.expected_steps = 8
    want = capture(store, 8)
    assert sorted(store.arena.spilled) == [4, 5, 6, 7], sorted(store.arena.spilled)
    assert store.arena._block.numel() == 3 * step_bytes and store.arena.total_bytes == budget == store.arena_bytes
    for s in (7, 0, 4, 3, 6, 5, 2, 1):
        check_step(store, s, want)
    # without a budget a recycled larger block IS reused -- sliced to the steps asked for, counted whole
    store.release_arena()
    big2 = AS.AttentionStore()
    big2.expected_steps = 12
    capture(big2, 12)
    big2.release_arena()
    small = AS.AttentionStore()
    small.expected_steps = 5
    w = capture(small, 5)
    assert small.arena._block.numel() == 11 * step_bytes and small.arena.reserved.numel() == 4 * step_bytes
    assert small.arena.total_bytes == 12 * step_bytes
    for s in range(5):
        check_step(small, s, w)


def test_a_step_that_outgrows_its_staging_slab_is_refused():
    store = AS.AttentionStore(disk_store=True, hbm_budget_bytes=0)
    capture(store, 2)
    for key, f, h, lq, lk, is_cross in LAYERS:
        store.new_slot(key, f, h, lq, lk, is_cross, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="overflows its staging slab"):
        store.new_slot("up_self", 2, 4, 64, 128, False, torch.device("cpu"))


def test_release_returns_the_host_slabs_to_the_pool():
    store = AS.AttentionStore(disk_store=True, hbm_budget_bytes=0)
    capture(store, 5)
    n = len(store.arena.spilled)
    assert n == 4
    store.release_arena()
    assert len(AS.MapArena._host_pool) == n and not store.arena.spilled and not store._all_step_maps
    again = AS.AttentionStore(disk_store=True, hbm_budget_bytes=0)
    capture(again, 5)
    assert len(AS.MapArena._host_pool) == 0                            # recycled
