#!/usr/bin/env python3
"""Host issue time of ONE UNet forward: the Python walk against the native issue plan (fatezero_amd/issue.py, csrc/plan.hip).

    python scripts/host_issue_time.py [--frames 8] [--controller none|store] [--null]   ->  one JSON line

Full SD-1.x width pseudo-3D UNet (lora 160), 64x64 latents, procedural weights.  What is timed is the HOST: wall time of `forward_tokens` from
call to return with nothing synchronised inside (the GPU queue is drained before each timed call), median of the timed forwards.
  * on a GPU box: real launches (hipLaunchKernelGGL each) -- the number that bounds a frame-sharded clip, where a rank's kernels are short;
  * `--null` (the authoring container: CPU emulation library with launches turned into no-ops, fz_emu_set_null_launch): what the host layer
    itself costs, launch calls excluded.
`--controller store`: with the capture controller of the inversion pass registered (32 controller events per forward, each a live Python call
plus the relocation of that layer's capture pointer into the step's arena slab)."""
import argparse
import ctypes
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SD15 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, cross_attention_dim=768,
            attention_head_dim=8, norm_num_groups=32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--controller", default="none", choices=["none", "store"])
    ap.add_argument("--null", action="store_true")
    ap.add_argument("--forwards", type=int, default=24)
    a = ap.parse_args()
    import fatezero_amd._native as N
    if a.null:
        from fatezero_amd import build
        N.use_test_backend(build.build_emu())
        N.lib().fz_emu_set_null_launch(1)
        dev = "cpu"
    else:
        dev = "cuda"
    from fatezero_amd.video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel
    from fatezero_amd.video_diffusion.models.resnet import Tokens
    from fatezero_amd.video_diffusion.prompt_attention.attention_register import register_attention_control
    from fatezero_amd.video_diffusion.prompt_attention.attention_store import AttentionStore
    mc = {"lora": 160, "SparseCausalAttention_index": [-1, "first"], "least_sc_channel": 1280}
    g = torch.Generator().manual_seed(0)
    out = {"frames": a.frames, "controller": a.controller, "launches": "null (host layer only)" if a.null else "real", "device": dev}
    for mode in ("walk", "plan"):
        unet = UNetPseudo3DConditionModel(sample_size=64, **SD15, **mc).half().to(dev).eval()
        store = None
        if a.controller == "store":
            store = AttentionStore()
            store.LOW_RESOURCE = True
            register_attention_control(SimpleNamespace(unet=unet), store)
        if mode == "plan":
            unet.enable_issue_plans()
        x = Tokens(torch.randn(a.frames, 4096, 4, generator=g).half().to(dev), 1, a.frames, 64, 64)
        ctx = (torch.randn(1, 77, 768, generator=g) * 0.5).half().to(dev)
        ts = []
        with torch.no_grad():
            for i in range(a.forwards):
                if dev == "cuda":
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                unet.forward_tokens(x, 981 - 20 * i, ctx)
                ts.append(time.perf_counter() - t0)
                if store is not None:
                    store.step_callback(x.data)   # closes the step: the next forward captures into the next slab of the arena
        if dev == "cuda":  # the same forwards back to back, nothing synchronised in between: what the GPU needs for one
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.no_grad():
                torch.cuda.synchronize()
                e0.record()
                for i in range(10):
                    unet.forward_tokens(x, 481 - 20 * i, ctx)
                    if store is not None:
                        store.step_callback(x.data)
                e1.record()
                torch.cuda.synchronize()
            out[mode + "_gpu_ms_per_forward"] = round(e0.elapsed_time(e1) / 10, 3)
        steady = sorted(ts[4:])
        out[mode + "_ms"] = round(1e3 * steady[len(steady) // 2], 3)
        out[mode + "_ms_min"] = round(1e3 * steady[0], 3)
        if mode == "plan":
            out["plan_stats"] = {k: v for k, v in unet._issuer.stats.items()}
            plan = next(p for p in unet._issuer.plans.values() if p is not None)
            out["launches_per_forward"], out["controller_events"] = plan.n, len(plan.events)
        if store is not None:
            store.release_arena()
        del unet
    out["speedup"] = round(out["walk_ms"] / out["plan_ms"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
