#!/usr/bin/env python3
"""bench.py -- edited frames/s of the FateZero hot path on MI355X.

A "step" is ONE FULL JOB of the hot path on one synthetic 8-frame 512x512 clip (BASELINE.json configs[1],
config/teaser/jeep_posche.yaml shape): a 50-step DDIM inversion with attention-map capture into the HBM arena,
followed by one 50-step classifier-free-guidance edit with attention fusion (Replace controller, cross 0.5 / self 0.5,
blend-mask-gated self-attention, th 0.3) -- latents in, latents out (VAE / CLIP / file I/O excluded on both sides,
SURVEY.md §8d).  value = frames edited per second over the whole job, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ddim-steps T] [--frames F] [--no-cpu-baseline]

N > 1: one rank per GPU over RCCL.  Launched under torch.distributed.run the ranks are taken from the environment; a BARE
`python bench.py --gpus N` re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1` (spawn_command below).  Default sharding: every rank edits its own clip (data-parallel over clips,
weak scaling; RCCL carries only latents: a weights checksum is broadcast-checked and the edited latents are all-gathered
over xGMI; no collective inside the UNet).  `--shard frames`: ONE clip's frames split over the ranks (strong scaling;
GroupNorm / K-V / temporal exchanges, fatezero_amd/dist.py).  With N > 1 the clips line carries `n_ranks_seen` from an RCCL
all-reduce and a watchdog-guarded `frame_sharded` measurement (one extra job after the clips measurement; `--no-frame-shard-probe`
skips it).

Extra JSON fields: `roofline` for the judged kernel (the 64x64-level fused spatio-temporal flash attention, 4096 x 8192 x
d=40: algorithmic FLOPs / HIP-event time measured live on the launch stream), `rooflines` = the same for the other
hand-written kernels that matter (3x3 convolutions and projection GEMMs: MFMA; attention-map capture / inject: HBM), and
`cpu_baseline` (the CPU oracle of the same loop on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SRC_PROMPT = "a silver jeep driving down a curvy road in the countryside,"
TGT_PROMPT = "a Porsche car driving down a curvy road in the countryside,"
EDIT_KW = dict(cross_replace_steps={"default_": 0.5}, self_replace_steps=0.5, use_inversion_attention=True,
               is_replace_controller=True, blend_words=[["silver", "jeep"], ["Porsche", "car"]],
               blend_self_attention=True, blend_th=[0.3, 0.3], save_self_attention=False, guidance_scale=7.5)
# the config's FIRST prompt pair (config/teaser/jeep_posche.yaml:22-47, p2p_config 0): source prompt -> itself, Refine + Reweight.
# The reference's validation loop edits every entry of `editing_prompts` from one inversion (p2p_validation_loop.py:95-140), so
# the config-faithful job is 1 inversion + 2 edits (--n-edit 2); the primary metric keeps n_edit = 1.
EDIT0_PROMPT = SRC_PROMPT
EDIT0_KW = dict(cross_replace_steps={"default_": 0.8}, self_replace_steps=0.9, use_inversion_attention=True,
                is_replace_controller=False, eq_params={"words": ["watercolor", "painting"], "values": [10, 10]},
                blend_words=[["jeep"], ["car"]], blend_self_attention=True, blend_th=[0.3, 0.3], save_self_attention=False,
                guidance_scale=7.5)
SD15 = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
            cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32)


class KernelTimer:
    """HIP-event timing of selected kernel launches on the launch stream (torch's current stream IS the stream every
    fz_* kernel is launched on)."""

    def __init__(self):
        self.events = []
        self.enabled = False

    def wrap(self, module, fn_name, select):
        orig = getattr(module, fn_name)
        timer = self

        def wrapped(*a, **k):
            tag = select(*a, **k) if timer.enabled else None
            if tag is None:
                return orig(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **k)
            e.record()
            timer.events.append((tag, s, e))
            return r
        setattr(module, fn_name, wrapped)

    def summary(self):
        out = {}
        for tag, s, e in self.events:
            ms = s.elapsed_time(e)
            d = out.setdefault(tag, [0, 0.0])
            d[0] += 1
            d[1] += ms
        return {k: {"launches": v[0], "avg_ms": v[1] / v[0], "total_ms": v[1]} for k, v in out.items()}


def build_pipeline(device, seed=0, model_config=None):
    from fatezero_amd.synthetic import HashTextEncoder, WordTokenizer, init_like_tuned_checkpoint
    from fatezero_amd.video_diffusion.models import UNetPseudo3DConditionModel
    from fatezero_amd.video_diffusion.pipelines.p2p_ddim_spatial_temporal import P2pDDIMSpatioTemporalPipeline
    from fatezero_amd.video_diffusion.schedulers import DDIMScheduler
    torch.manual_seed(seed)
    with torch.device(device):
        unet = UNetPseudo3DConditionModel(**SD15, **(model_config or {"lora": 160}))
    init_like_tuned_checkpoint(unet, seed)
    unet = unet.half().eval()
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=HashTextEncoder(768).to(device), tokenizer=WordTokenizer(),
                                         unet=unet, scheduler=DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    return pipe


PMC_JOB_FILES = ("r06_pmc_job.json", "r05_pmc_job.json", "r04_pmc_job.json")  # newest in-situ pass first

# The same tree measured 2.14 - 2.48 s per job across boxes of this pool (an UNCHANGED flash kernel moved 9 % between the round-3 and the
# round-4 driver box), so the line carries a normaliser: `box` = a fixed flash launch and a fixed 1 GB copy timed BEFORE the warm-up, the
# same launch again straight behind the timed jobs (the chip hot), and the chip's clock / socket power sampled DURING the timed region.
# FLASH_CALIB_REF_US is a convention, not a measurement of any particular box: the hot calibration launch (8 frames x 4096 x 8192 x d 40,
# random operands) read 396 us on the round-5 closing box; `value_normalised` = value x hot calibration / 400.
FLASH_CALIB_REF_US = 400.0


def measure_box(K, device, n=12, hot=False):
    """Box calibration, nothing of the job in it: (1) ONE fixed launch of the judged kernel -- attn_flash d = 40, 8 frames x Lq 4096 x Lk
    8192, uniform random operands -- median of n launches after 60 untimed ones, HIP events on the launch stream; (2) a device-to-device
    copy of 1 GiB (1 GiB read + 1 GiB written), median of 5.  hot: only (1), called straight behind the timed region -- the chip at the
    power / clock state the job leaves it in (1.09 kW, ~2.06 GHz) instead of the cool 2.4 GHz a fresh process finds: the launch reads
    ~12 % longer there, and THAT figure is what the headline is normalised with."""
    out = {}
    try:
        g = torch.Generator().manual_seed(7)
        q = (torch.randn(8, 4096, 320, generator=g) * 0.5).half().to(device)
        k = (torch.randn(8, 4096, 320, generator=g) * 0.5).half().to(device)
        vt = torch.randn(8, 320, 4096, generator=g).half().to(device)
        o = torch.empty_like(q)
        kw = dict(clip_len=8, heads=8, index_list=[-1, "first"], mode=K.FZ_ATTN_FLASH, scale=40 ** -0.5, q_log2_scaled=True)
        for i in range(60):  # ~30 ms of the same launch first: the clocks leave idle before anything is timed
            K.attn_self(q, k, vt, o, **kw)
        ev = []
        for i in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            K.attn_self(q, k, vt, o, **kw)
            e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        us = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
        out["flash_calib_us"] = us[len(us) // 2]
        out["flash_calib_us_min_max"] = [us[0], us[-1]]
        out["flash_calib"] = "attn_flash_kernel<40>, 8 frames x 4096 x 8192, random operands, median of %d launches after 60 untimed ones" % n
        out["flash_calib_ref_us"] = FLASH_CALIB_REF_US
        if hot:
            return {"flash_calib_hot_us": out["flash_calib_us"], "flash_calib_hot_us_min_max": out["flash_calib_us_min_max"]}
        a = torch.empty(1 << 29, dtype=torch.float16, device=device)
        b = torch.empty_like(a)
        ev = []
        for i in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            b.copy_(a)
            e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev[2:])
        out["copy_1GiB_TBps"] = 2.0 * (1 << 30) / (ms[len(ms) // 2] * 1e-3) / 1e12
        del a, b, q, k, vt, o
    except Exception as e:  # noqa: BLE001 -- the calibration is a report, never a reason to lose the measurement
        out["error"] = repr(e)
    return out


class SmiSampler:
    """Shader clock (MHz) and socket power (W) of the device under test, sampled every `period` s on a helper thread while the timed region
    runs.  The box's sysfs lists EVERY amdgpu device of the node (the process sees one): the hwmon directory is picked by the PCI address
    torch reports for the device, else the card drawing the most power during the region is the one reported (`picked_by`).  hwmon files
    (one small read each); `rocm-smi --showclocks --showpower` when there is no hwmon."""

    def __init__(self, index=0, period=0.5):
        import threading
        self.index, self.period = index, period
        self.samples, self._stop = {}, threading.Event()
        self._thread = threading.Thread(target=self._run, name="fz-smi", daemon=True)
        self.source, self.picked_by = None, None
        self._hw = self._find_hwmons()
        self._mine = self._pci_hwmon(index)

    @staticmethod
    def _find_hwmons():
        import glob
        out = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            out += sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))[:1]
        return out

    @staticmethod
    def _pci_hwmon(index):
        import glob
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hw = sorted(glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*"))
            return os.path.realpath(hw[0]) if hw else None
        except Exception:  # noqa: BLE001
            return None

    @staticmethod
    def _read_hwmon(hw):
        mhz = w = None
        try:
            mhz = float(open(os.path.join(hw, "freq1_input")).read()) / 1e6
        except (OSError, ValueError):
            pass
        for name in ("power1_average", "power1_input"):
            try:
                w = float(open(os.path.join(hw, name)).read()) / 1e6
                break
            except (OSError, ValueError):
                continue
        return mhz, w

    def _read_smi(self):
        import re
        import subprocess
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception:  # noqa: BLE001
            return {}
        out = {}
        for m in re.finditer(r"GPU\[(\d+)\]\s*:\s*sclk clock level[^(]*\((\d+)Mhz\)", txt):
            out.setdefault("smi%s" % m.group(1), [None, None])[0] = float(m.group(2))
        for m in re.finditer(r"GPU\[(\d+)\]\s*:[^\n]*Power \(W\):\s*([0-9.]+)", txt):
            out.setdefault("smi%s" % m.group(1), [None, None])[1] = float(m.group(2))
        return {k: tuple(v) for k, v in out.items()}

    def _run(self):
        while not self._stop.is_set():
            got = {}
            for hw in self._hw:
                mhz, w = self._read_hwmon(hw)
                if mhz is not None or w is not None:
                    got[os.path.realpath(hw)] = (mhz, w)
            self.source = "hwmon"
            if not got:
                got = self._read_smi()
                self.source = "rocm-smi"
            for k, v in got.items():
                self.samples.setdefault(k, []).append(v)
            self._stop.wait(self.period)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=15)

        def stat(v):
            v = sorted(x for x in v if x)
            return None if not v else {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        key = None
        if self._mine is not None and self._mine in self.samples:
            key, self.picked_by = self._mine, "pci address of the torch device"
        elif self.samples:  # the card under load
            key = max(self.samples, key=lambda k: (stat([w for _, w in self.samples[k]]) or {"median": 0.0})["median"])
            self.picked_by = "highest median power among %d devices" % len(self.samples)
        if key is None:
            return {"samples": 0, "source": self.source, "sclk_MHz": None, "socket_power_W": None}
        sm = self.samples[key]
        return {"samples": len(sm), "source": self.source, "device": key, "picked_by": self.picked_by, "devices_seen": len(self.samples),
                "sclk_MHz": stat([m for m, _ in sm]), "socket_power_W": stat([w for _, w in sm])}


def pmc_job_traffic():
    """HBM traffic per kernel class from the IN-SITU counter passes over one bench job (scripts/pmc_job.sh: rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE in two separate runs of this very script, every dispatch of the job attributed to a class by kernel name --
    and, for the projection GEMMs, by the launch log this script writes with FZ_BENCH_LAUNCHLOG set).  Counters cannot be read from
    inside this process, so the committed summary under profiles/ is what the line quotes: bytes per launch, FETCH_SIZE doubled
    (gfx950 tallies a 128-byte request as 64, MI355X_MICROARCH.md, HBM section), both counters in KiB."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in PMC_JOB_FILES:
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                return json.load(f), f"profiles/{name}"
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic_per_launch(frames_per_launch):
    """HBM bytes of the judged kernel per launch: the in-situ job pass when it exists (pmc_job_traffic), else the A/B-harness pass of
    scripts/pmc_flash.sh (same kernel, 8 frames per launch, scaled linearly to this run's frames per launch)."""
    job, src = pmc_job_traffic()
    if job is not None and "flash" in job.get("classes", {}):
        c = job["classes"]["flash"]
        return c["traffic_bytes_per_launch"], f"{src} (in situ: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over one bench job, {c['launches']} launches)"
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("r03_pmc_flash_d40_final2.json", "r03_pmc_flash_d40_final.json", "r02_pmc_flash_d40_final.json", "r01_pmc_flash_d40_final.json"):  # newest measurement of the shipped kernel first
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                pmc = json.load(f)
            per8 = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            return per8 * frames_per_launch / 8.0, f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def run_job(pipe, z0, ddim_steps, device, n_edit=1, blend_th=None):
    """One full job: capture inversion + n_edit CFG edits (1: the Porsche edit; 2: both prompts of the config). Returns the
    (last) edited latents.  blend_th: overrides the config's 0.3 for the Porsche edit (the split-mask job, see pick_split_threshold)."""
    pipe.scheduler.set_timesteps(ddim_steps)
    pipe.release_attention_maps()                          # previous job's 75 GB arena block goes back to the pool
    pipe.store_controller = type(pipe.store_controller)()  # fresh store per job
    emb_src = pipe._encode_prompt(SRC_PROMPT, device, 1, True, None)
    lat = pipe.prepare_latents_ddim_inverted(image=None, batch_size=1, num_images_per_prompt=1, text_embeddings=emb_src,
                                             store_attention=True, LOW_RESOURCE=True, latents=z0)
    if n_edit >= 2:
        # p2p_config 0 of the YAML.  `eq_params` names words ("watercolor", "painting") that are NOT in this prompt -- the YAML was
        # copied from jeep_watercolor -- and `blend_words` names "car" in a prompt without it: get_equalizer / get_word_inds then
        # select nothing, as in the reference (ptp_utils.py:144-160 returns an empty index array)
        pipe(prompt=EDIT0_PROMPT, source_prompt=SRC_PROMPT, edit_type="swap", num_inference_steps=ddim_steps,
             latents=lat[-1], output_type="latent", **EDIT0_KW)
    kw = EDIT_KW if blend_th is None else dict(EDIT_KW, blend_th=[blend_th, blend_th])
    out = pipe(prompt=TGT_PROMPT, source_prompt=SRC_PROMPT, edit_type="swap", num_inference_steps=ddim_steps,
               latents=lat[-1], output_type="latent", **kw)
    return out["sdimage_output"].images


def stored_rows_fraction(pipe):
    """Share of the self-attention rows of the last edit that took the STORED map (blend mask 0), over every blend-mask call."""
    ab = getattr(getattr(pipe, "last_edit_controller", None), "attention_blend", None)
    ml = getattr(ab, "mask_list", None)
    if not ml:
        return None
    return 1.0 - float(sum(float(m.float().sum()) for m in ml) / sum(m.numel() for m in ml))


def pick_split_threshold(pipe, z0, device, steps=10, iters=10):
    """With the bench's procedural weights the normalised blend-word score is near-uniform (>= 0.9 of its per-frame maximum almost
    everywhere) and the config's blend_th = 0.3 leaves 100 % of the rows on the LIVE attention: the 275 masked-inject launches of a job
    then read no stored map (round-4 review).  The YAML knob that moves the split is blend_th itself (the tests use 0.55 for the same
    reason): the stored-row share is monotone in it, so bisect it on short `steps`-step jobs (the same timestep range, coarser) towards
    one half; the threshold found is used for the `split_mask_job` beside the primary and for the kernel-breakdown job."""
    lo, hi, seen = 0.3, 0.9995, {}
    best = None
    for _ in range(iters):
        th = 0.5 * (lo + hi)
        run_job(pipe, z0, steps, device, blend_th=th)
        f = stored_rows_fraction(pipe)
        if f is None:
            return None, seen
        seen[round(th, 6)] = f
        if best is None or abs(f - 0.5) < abs(seen[best] - 0.5):
            best = round(th, 6)
        if f < 0.5:
            lo = th
        else:
            hi = th
    return best, seen


# The reference's OWN modules (imported unmodified through oracle/refshim) against the port on the same UNet forward, measured once in the
# authoring container (8 cores; profiles/r04_cpu_ref_vs_port.txt): neither /root/reference nor diffusers exists on the GPU box, so the line
# carries the ratio instead of a second timing -- cpu_baseline.value is the PORT's; the reference's own code is slower by these factors.
REFERENCE_OVER_PORT = {"inversion_forward": 1.195, "cfg_edit_forward": 1.785, "source": "profiles/r04_cpu_ref_vs_port.txt (3 frames x 512^2, fp32, "
                       "8 cores of the authoring container; outputs agree to 4.8e-6)"}


def cpu_baseline(pipe, ddim_steps, frames, sample_frames=3, k=2):
    """The CPU oracle (oracle/fatezero_oracle.py, fp32 restatement of the reference loop) on the host cores, after BASELINE.md
    section 3: after ONE warm-up step, k capture-inversion steps and k CFG edit steps (edit steps 0..k-1: inside both replace
    windows, the expensive case) of a `sample_frames`-frame 512x512 clip with the bench's weights and controller.  THREE frames:
    with [-1, 'first'] the two K/V slots of frame 2 are frames 1 and 0 -- distinct, as in the 8-frame job (a 2-frame sample has
    both slots on frame 0) -- and GroupNorm spans three frames.  Steps are homogeneous and the cost is linear in the frame count
    (sparse-causal attention: every frame attends two frames), so the job time is extrapolated as
    T * (t_inv + t_edit) / k * frames / sample_frames.  k = 2.  The reference's own modules cannot
    be timed here: neither /root/reference nor diffusers exists on the GPU box, hence kind = "port"; `reference_over_port` carries what the
    reference's own code measured against the port on the same forward, and `value_reference_estimate` the port's value divided by it."""
    import platform
    from oracle.host_cpu import cpu_budget, size_torch_pool
    size_torch_pool()  # the box's cgroup CPU quota, not the 256 logical CPUs torch sees (oracle/host_cpu.py)
    from oracle import fatezero_oracle as O
    sd = {kk: v.float().cpu() for kk, v in pipe.unet.state_dict().items()}
    cfg = O.UNetConfig(block_out_channels=SD15["block_out_channels"], attention_head_dim=8, cross_attention_dim=768,
                       norm_num_groups=32, model_config={"lora": 160})
    unet = O.OracleUNet(sd, cfg)
    tok = pipe.tokenizer
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 4, sample_frames, 64, 64, generator=g)
    emb = torch.randn(2, 77, 768, generator=g)
    sched = O.DDIMSchedule(ddim_steps)
    ts = [int(t) for t in sched.timesteps]
    store = O.StoreController()
    store.LOW_RESOURCE = True
    warm = O.StoreController()
    warm.LOW_RESOURCE = True
    t0 = time.time()
    unet(z, ts[-1], emb[1:], warm)  # warm-up step (thread pool, allocator), not timed
    t_warm = time.time() - t0
    t0 = time.time()
    for i in range(k):
        t = ts[len(ts) - 1 - i]
        eps = unet(z, t, emb[1:], store)
        z = sched.inverse_step(eps, t, z)
        store.step_callback(z)
    t_inv = time.time() - t0
    store.LOW_RESOURCE = False
    # the edit controller reads inversion step T-1-cur_step: give it a store that LOOKS like a T-step inversion whose last
    # k entries are the ones just recorded (edit steps 0..k-1 read exactly those)
    store.attention_store_all_step = [store.attention_store_all_step[0]] * (ddim_steps - k) + store.attention_store_all_step
    ctrl = O.make_edit_controller(tok, [SRC_PROMPT, TGT_PROMPT], store, ddim_steps, True, {"default_": 0.5}, 0.5,
                                  blend_words=EDIT_KW["blend_words"], blend_th=(0.3, 0.3), blend_self_attention=True,
                                  save_self_attention=False)
    t0 = time.time()
    for i in range(k):
        t = ts[i]
        eps2 = unet(torch.cat([z, z]), t, emb, ctrl)
        eu, ec = eps2.chunk(2)
        z = sched.step(eu + 7.5 * (ec - eu), t, z)
        z = ctrl.step_callback(z)
    t_edit = time.time() - t0
    job_s = ddim_steps * (t_inv + t_edit) / k * frames / sample_frames
    cpu = platform.processor() or platform.machine()
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    ref_job_s = ddim_steps * (t_inv * REFERENCE_OVER_PORT["inversion_forward"] + t_edit * REFERENCE_OVER_PORT["cfg_edit_forward"]) / k \
        * frames / sample_frames
    return {"value": frames / job_s, "unit": "edited frames/s", "cores": torch.get_num_threads(), "kind": "port", "k": k,
            "reference_over_port": REFERENCE_OVER_PORT, "value_reference_estimate": frames / ref_job_s,
            "cpu_model": cpu, "logical_cpus_visible": os.cpu_count(), "cpu_budget": cpu_budget(),
            "sample": f"after 1 warm-up step ({t_warm:.1f} s): {k} capture-inversion steps ({t_inv:.1f} s) + {k} CFG edit steps "
                      f"({t_edit:.1f} s) of a {sample_frames}-frame 512x512 clip, full-size SD-1.x pseudo-3D UNet fp32 "
                      f"(oracle/fatezero_oracle.py), same weights / controller; extrapolated x{ddim_steps}/{k} steps and "
                      f"x{frames}/{sample_frames} frames"}


def cpu_cfg1_full(pipe, frames=8, latent=32, steps=10):
    """cfg1 (config/low_resource_teaser, 8 frames x 256^2, 10 DDIM steps) measured IN FULL on the host cores
    -- no extrapolation: 10 capture-inversion steps + 10 CFG edit steps of the CPU oracle at full SD-1.x width, cfg1's model config
    ({lora 160, SparseCausalAttention_index ['mid'], least_sc_channel 640}) and controller (Refine + Reweight x10, no blend)."""
    from oracle.host_cpu import size_torch_pool
    size_torch_pool()
    from oracle import fatezero_oracle as O
    mc = {"lora": 160, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 640}
    sd = {kk: v.float().cpu() for kk, v in pipe.unet.state_dict().items()}
    unet = O.OracleUNet(sd, O.UNetConfig(block_out_channels=SD15["block_out_channels"], attention_head_dim=8, cross_attention_dim=768,
                                         norm_num_groups=32, model_config=mc))
    src = "a silver jeep driving down a curvy road in the countryside"
    tgt = "watercolor painting of a silver jeep driving down a curvy road in the countryside"
    g = torch.Generator().manual_seed(2)
    z = torch.randn(1, 4, frames, latent, latent, generator=g)
    emb = torch.randn(2, 77, 768, generator=g)
    store = O.StoreController()
    t0 = time.time()
    lat = O.ddim_inversion(unet, O.DDIMSchedule(steps), z, emb[1:], store)
    t_inv = time.time() - t0
    ctrl = O.make_edit_controller(pipe.tokenizer, [src, tgt], store, steps, False, {"default_": 0.8}, 0.8,
                                  eq_params={"words": ["watercolor"], "values": [10]}, save_self_attention=False)
    t0 = time.time()
    out = O.ddim_edit(unet, O.DDIMSchedule(steps), lat[-1], emb, ctrl, guidance_scale=7.5)
    t_edit = time.time() - t0
    return {"value": frames / (t_inv + t_edit), "unit": "edited frames/s", "t_inversion_s": t_inv, "t_edit_s": t_edit,
            "cores": torch.get_num_threads(), "kind": "port", "outputs_finite": bool(torch.isfinite(out).all()),
            "sample": f"cfg1 in full: {frames} frames x {8 * latent}^2, {steps}+{steps} DDIM steps, no extrapolation"}


def spawn_command(argv, gpus, port=None):
    """The command a bare `python bench.py --gpus N` (no WORLD_SIZE in the environment) re-executes itself as."""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def install_timers(K, timer):
    """HIP-event brackets (on the launch stream) around the hand-written kernels the rooflines are quoted for."""
    def sel_attn(q, k, vt, out, **kw):
        mode = kw.get("mode", K.FZ_ATTN_FLASH)
        nf = kw.get("n_frames") or q.shape[0]
        if mode == K.FZ_ATTN_FLASH:
            if q.shape[1] == 4096 and q.shape[2] == 320:
                # kv slots the kernel actually contracts: a frame whose slots all resolve to ONE source frame reads it once
                # (csrc/attn_flash.hip); the algorithmic figure prices every slot, as the reference computes it
                n_kv, clip, f0 = max(1, len(kw["index_list"])), kw["clip_len"], kw.get("frame0", 0)
                read = nf * n_kv
                if kw.get("kv_slots_override") is None and not kw.get("kv_clip_len") and len(kw["index_list"]) > 1:
                    kabs, kval = K.kv_slots(kw["index_list"], clip)
                    read = 0
                    for n in range(f0, f0 + nf):
                        src = {v if a else min(max(n % clip + v, 0), clip - 1) for a, v in zip(kabs, kval)}
                        read += 1 if len(src) == 1 else n_kv
                return ("flash", nf, n_kv, read)
            return None
        if not timer.extra:
            return None
        p = kw["p"]
        per_frame = p.shape[1] * p.shape[2] * p.shape[3] * 2  # bytes of the fp16 map of one frame
        # algorithmic bytes of the launch = the map itself -- q / k / V^T / o are < 2 % of it.  A masked inject only
        # READS the rows whose blend mask is 0 (rows that keep the live attention never touch the stored map): the third field prices
        # those (one host sync per launch -- this runs in the extra, untimed job only)
        read = nf * per_frame
        rm = kw.get("row_mask")
        if mode == K.FZ_ATTN_INJECT and rm is not None:
            m0 = kw.get("mask_frame_off", 0)
            read *= float(1.0 - rm.reshape(-1, rm.shape[-1])[m0: m0 + nf].float().mean())
        if mode == K.FZ_ATTN_CAPTURE:
            return ("capture", nf * per_frame, nf * per_frame)
        return ("inject", nf * per_frame, nf * per_frame, round(read))  # (4th field: the bytes of the rows really read)
    timer.wrap(K, "attn_self", sel_attn)

    def sel_conv(x, wt, bias, **kw):
        if not timer.extra or x.shape[2] % 8:
            return None
        h, w = kw["hw"]
        st = kw.get("stride", 1)
        ho, wo = ((2 * h if kw.get("upsample") else h) - 1) // st + 1, ((2 * w if kw.get("upsample") else w) - 1) // st + 1
        n, cin, cout = x.shape[0], x.shape[2], wt.shape[0]
        alg = 2.0 * (x.numel() + n * ho * wo * cout * (2 if kw.get("res") is not None else 1) + 9 * cin * cout)
        return ("conv3x3", 2.0 * 9 * cin * cout * n * ho * wo, alg)
    timer.wrap(K, "conv3x3", sel_conv)

    def sel_conv_up2(x, wt_up, bias, **kw):  # nearest-2x + 3x3 as four 2x2 convolutions: priced at the FLOPs of the operation (nine taps per output)
        if not timer.extra:
            return None
        h, w = kw["hw"]
        n, cin, cout = x.shape[0], x.shape[2], wt_up.shape[1]
        alg = 2.0 * (x.numel() + n * 4 * h * w * cout + 9 * cin * cout)
        return ("conv3x3", 2.0 * 9 * cin * cout * n * 4 * h * w, alg)
    timer.wrap(K, "conv3x3_up2", sel_conv_up2)

    def sel_gemm(x, w, bias=None, **kw):
        if not timer.extra:
            return None
        tag = gemm_class(x, w, kw)
        return None if tag is None else tag
    timer.wrap(K, "gemm", sel_gemm)

    def sel_qkvt(x, w, split, **kw):  # the fused q | k | V^T projection: a plain projection with N = 3 C (no residual)
        if not timer.extra:
            return None
        return gemm_class(x, w, {})
    timer.wrap(K, "gemm_qkvt", sel_qkvt)

    def sel_gemm_gn(x, w, bias=None, **kw):  # a projection whose epilogue also writes the next GroupNorm's statistics (proj_out + residual)
        if not timer.extra:
            return None
        return gemm_class(x, w, kw)
    timer.wrap(K, "gemm_gn", sel_gemm_gn)

    def sel_gemm_lnout(x, w, bias, ln, **kw):  # a projection whose epilogue also writes the next LayerNorm's output (one more rows x N write)
        if not timer.extra:
            return None
        return gemm_class(x, w, dict(kw, ln_out=True))
    if hasattr(K, "gemm_lnout"):
        timer.wrap(K, "gemm_lnout", sel_gemm_lnout)

    def sel_ff_chain(xn, packed, b2, inner, **kw):  # the whole feed-forward of a 64x64-level block in one launch (csrc/ff_chain.hip)
        if not timer.extra:
            return None
        c = xn.shape[-1]
        rows = xn.numel() // c
        n_io = 2 + (kw.get("res") is not None) + (kw.get("ln") is not None)     # xn in, y out (+ res in) (+ LayerNorm(y) out)
        return ("ff_chain", 2.0 * rows * (c * 2 * inner + inner * c), 2.0 * (rows * c * n_io + 3 * c * inner))
    if hasattr(K, "ff_chain"):
        timer.wrap(K, "ff_chain", sel_ff_chain)

    def sel_xattn_chain(x, packed, kv_packed, bias_out, **kw):  # attn2 of a 64x64-level block in one launch (csrc/xattn_chain.hip)
        if not timer.extra:
            return None
        n, l, c = x.shape
        rows, lk, front = n * l, kw.get("lk", 77), kw.get("front_eps") is not None
        # 2 rows (2 C C + 2 lk C) FLOP (to_q, to_out, Q K^T, P V; + 2 rows C C with attn1.to_out in front) = 26 FLOP per algorithmic byte:
        # far below the chip's ridge of 312 -- the launch is priced against the HBM roof on the bytes it must move
        n_io = 3 + (kw.get("ln") is not None) + (1 if front else 0)                           # x, res in, y out (+ LayerNorm(y)) (+ y1)
        nbytes = 2.0 * rows * c * n_io + float(packed.numel() + kv_packed.numel())
        return ("xattn_chain", nbytes, nbytes)
    if hasattr(K, "xattn_chain"):
        timer.wrap(K, "xattn_chain", sel_xattn_chain)


def gemm_class(x, w, kw):
    """Roofline class of one projection GEMM (>= 1024 rows).  The launches fall into two regimes: plain projections with
    K <= 640 move 2 (rows K + rows N + K N) bytes (+ one more rows x N per residual) for 2 rows K N FLOP -- <= 213 FLOP per byte at
    K = N = 640, below the chip's ridge of 2500 T / 8 T = 312: their roof is HBM; the GEGLU projections (N = 8 C, gate in the epilogue)
    and the long-K ones (K >= 1280) sit above it: MFMA.  Returns (class, work in the class's unit, algorithmic bytes)."""
    rows = x.numel() // x.shape[-1]
    if rows < 1024:
        return None
    if w is None:  # LayerNorm folded into the projection (fz_gemm_ln): the weights travel in `ln`
        w = kw["ln"].w
    k, o = x.shape[-1], w.shape[0]
    geglu = bool(kw.get("geglu"))
    n_out = o // 2 if geglu else o
    n_res = (kw.get("res") is not None) + (kw.get("res2") is not None)
    alg = 2.0 * (rows * k + rows * n_out * (1 + n_res + (1 if kw.get("ln_out") else 0)) + k * o)
    flops = 2.0 * rows * k * o
    if geglu or k > 640:
        return ("gemm_mfma", flops, alg)
    return ("gemm_hbm", alg, alg)


def install_launch_log(K, path):
    """FZ_BENCH_LAUNCHLOG=<path> (scripts/pmc_job.sh): one entry per MODE-0 igemm dispatch of this process, in launch order -- the
    roofline class of the GEMM (gemm_class; 'gemm_small' below 1024 rows, 'gemm_vt' for the transposed V projection) -- so that the
    counter rows rocprofv3 writes per dispatch can be attributed to the classes the line reports.  Written at exit."""
    import atexit
    log = []
    for fn_name in ("gemm", "gemm_vt", "gemm_batched", "gemm_qkvt", "gemm_gn", "gemm_lnout"):
        orig = getattr(K, fn_name)

        def wrapped(x, w, *a, _orig=orig, _name=fn_name, **k):
            if _name in ("gemm", "gemm_qkvt", "gemm_gn", "gemm_lnout"):
                tag = gemm_class(x, w, k if _name != "gemm_qkvt" else {})
                log.append("gemm_small" if tag is None else tag[0])
            else:
                log.append(_name)
            return _orig(x, w, *a, **k)
        setattr(K, fn_name, wrapped)
    atexit.register(lambda: json.dump(log, open(path, "w")))


def rooflines(summ):
    """(judged flash roofline, list of the other kernels' rooflines) from the timer summary."""
    roof, others = None, []
    fl = {k: v for k, v in summ.items() if k[0] == "flash"}
    if fl:
        flops_total, flops_read, ms_total, launches, frames_total = 0.0, 0.0, 0.0, 0, 0
        for (tag, nf, n_kv, read), v in fl.items():
            frames_total += nf * v["launches"]
            flops_total += 4.0 * 4096 * (n_kv * 4096) * 320 * nf * v["launches"]  # 4*Lq*Lk*C per frame
            flops_read += 4.0 * 4096 * 4096 * 320 * read * v["launches"]
            ms_total += v["total_ms"]
            launches += v["launches"]
        achieved = flops_total / (ms_total * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic_per_launch(frames_total / launches)
        roof = {"kernel": "attn_flash_kernel<40> (64x64 level, Lq 4096, Lk 8192, d 40)", "bound": "mfma",
                "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0, "traffic": traffic,
                "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "launches": launches,
                "avg_launch_ms": ms_total / launches, "algorithmic_flops_per_launch": flops_total / launches,
                # frames 0 and 1 of a clip see frame 0 in both [-1, 'first'] slots and read it once (the same softmax): the rate over
                # the key tiles the kernel really contracts, beside the algorithmic one SURVEY section 8(d) defines
                "launches_note": "every attn_flash d = 40 launch of the process between timer reset and summary: 500 per timed job (250 x 8 frames "
                                 "+ 250 x 16 frames) plus the box calibration's 8-frame launches where they fall inside -- avg_launch_ms and "
                                 "algorithmic_flops_per_launch are means over that mix",
                "contracted_fraction": flops_read / flops_total,
                "achieved_over_contracted_tiles": achieved * flops_read / flops_total}
    if roof is not None:  # q, k (two source frames' worth is re-read from L2, not algorithmic), V^T in, o out: 4 x Lq x C halves per frame
        roof["algorithmic_bytes_per_launch"] = 2.0 * 4 * 4096 * 320 * frames_total / launches
        if roof["traffic"]:
            roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
    job_pmc, job_src = pmc_job_traffic()
    for name, kernel, bound, peak, unit, scale in (
            ("conv3x3", "all 3x3 convolutions at 2*9*Cin*Cout FLOP per output pixel: igemm_kernel<.., MODE 1 / 3>, conv_halo_kernel<9> (whole, or K slices + the split-K "
                        "tail), conv_halo_kernel<4> (nearest-2x + 3x3 as four 2x2 convolutions: 4 / 9 of the multiply-adds, priced at the operation's nine)", "mfma", 2500.0,
             "TFLOP/s", 1e12),
            ("gemm_mfma", "igemm_kernel<.., MODE 0> MFMA class: GEGLU projections and K >= 1280 projections with >= 1024 rows (2*K*N FLOP per row)",
             "mfma", 2500.0, "TFLOP/s", 1e12),
            ("gemm_hbm", "igemm_kernel<.., MODE 0> HBM class: plain projections with K <= 640 and >= 1024 rows (algorithmic bytes "
                         "2 (rows K + rows N (1 + residuals) + K N))", "hbm", 8000.0, "GB/s", 1e9),
            ("ff_chain", "ff_chain_kernel (64x64 level: GEGLU up-projection -> gate -> down-projection + residual + LayerNorm in one launch, "
                         "2 rows (C 2 inner + inner C) FLOP)", "mfma", 2500.0, "TFLOP/s", 1e12),
            ("xattn_chain", "xattn_chain_kernel (64x64 level: to_q -> 77-key cross-attention -> to_out + residual + LayerNorm in one launch; roof: HBM "
                            "on its algorithmic bytes 2 rows C (x, res, y, LN(y)) + the packed operands)", "hbm", 8000.0, "GB/s", 1e9),
            ("capture", "attn_self_kernel<CAPTURE> (bytes of the fp16 probability maps written to the HBM arena)", "hbm", 8000.0, "GB/s", 1e9),
            ("inject", "attn_self_kernel<INJECT> (bytes of the stored maps read back)", "hbm", 8000.0, "GB/s", 1e9)):
        sel = {k: v for k, v in summ.items() if k[0] == name}
        if not sel:
            continue
        work = sum(k[1] * v["launches"] for k, v in sel.items())
        alg = sum(k[2] * v["launches"] for k, v in sel.items())
        ms = sum(v["total_ms"] for v in sel.values())
        n = sum(v["launches"] for v in sel.values())
        ach = work / (ms * 1e-3) / scale
        ent = {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
               "traffic": None, "launches": n, "total_ms": ms, "algorithmic_bytes_per_launch": alg / n,
               "sampled": "one extra job after the timed region, HIP events per launch"}
        if name == "inject":
            # with procedural weights the blend-word score is near-uniform and the config's th = 0.3 keeps ~99 % of the rows on the live
            # attention: those rows run QK^T + softmax and read NO stored map, so the launch moves far fewer bytes than the map it is
            # priced on (SURVEY 8(d)'s figure) -- say so, and give the rate over the rows really read
            read = sum((k[3] if len(k) > 3 else k[1]) * v["launches"] for k, v in sel.items())
            ent["stored_rows_fraction"] = read / work if work else None
            ent["achieved_over_rows_read"] = ach * read / work if work else None
            if read and work:  # the roofline of the launch on what it READ: stored rows only (live rows run QK^T + softmax and read no map)
                ent["priced_on"] = "bytes of the stored rows actually read (blend mask 0), not the whole map"
                ent["achieved_whole_map"], ent["frac_whole_map"] = ach, ach / peak
                ent["achieved"], ent["frac"] = ach * read / work, ach * read / work / peak
                ent["algorithmic_bytes_per_launch"] = read / n
        if bound == "mfma":  # the same launches against the OTHER roof, so that the class can be read off the line
            ent["algorithmic_GBps"] = alg / (ms * 1e-3) / 1e9
        c = (job_pmc or {}).get("classes", {}).get(name)
        if c is not None:
            ent.update(traffic=c["traffic_bytes_per_launch"], traffic_unit="bytes/launch", traffic_launches=c["launches"],
                       traffic_source=f"{job_src} (in situ, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)",
                       traffic_over_algorithmic=(c["traffic_bytes_per_launch"] / ent["algorithmic_bytes_per_launch"])
                       if ent["algorithmic_bytes_per_launch"] > 0 else None)
        others.append(ent)
    return roof, others


def _all_ranks_ok(ok, device):
    """Collective agreement on a local success flag: True only when EVERY rank of the default group reports success (a rank that falls back to
    the collectives while its peers stay on the peer transport would deadlock the first exchange of the UNet)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def make_frame_shard(fz_dist, frames, transport, heap_gb, device):
    """FrameShard for this job + the transport that carries its exchanges.  'peer' / 'auto': map the peers' symmetric heaps and run ONE
    small all-gather through them as a self-test (bounded wait); 'auto' falls back to the collectives when that fails.  The outcome of
    every stage (peer access, heap mapping + self-test) is AGREED over the process group (all_reduce MIN of a success flag): all ranks
    take the same transport, whichever rank saw the failure."""
    shard = fz_dist.FrameShard(frames)
    used = "rccl"
    if transport in ("peer", "auto") and shard.world > 1:
        why = None
        try:
            # kernels will store straight into the other GPUs' memory: only where the runtime says every pair of this node's devices has
            # peer access (a missing link would be a GPU fault, not an exception)
            n_dev = torch.cuda.device_count() if hasattr(torch.cuda, "device_count") else 1
            me = device.index if getattr(device, "index", None) is not None else 0
            if hasattr(torch.cuda, "can_device_access_peer") and n_dev > 1:
                for other in range(min(n_dev, shard.world)):
                    if other != me and not torch.cuda.can_device_access_peer(me, other):
                        raise RuntimeError(f"no peer access between devices {me} and {other}")
        except Exception as e:  # noqa: BLE001
            why = repr(e)
        if not _all_ranks_ok(why is None, device):  # before the collective heap exchange: nobody enters it alone
            why = why or "a peer rank has no peer access"
        else:
            try:
                shard.enable_peer_transport(nbytes=int(heap_gb * (1 << 30)), device=device, timeout_us=10_000_000)
                probe = torch.full((1, shard.n_local, 8), float(shard.rank + 1), device=device)
                got = shard.all_gather_frames(probe, tag="selftest")
                torch.cuda.synchronize()
                shard.heap.check()
                want = torch.cat([torch.full((1, len(shard.frames_of(r)), 8), float(r + 1)) for r in range(shard.world)], 1)
                if not torch.equal(got.cpu(), want):
                    raise RuntimeError("peer transport self-test returned wrong data")
            except Exception as e:  # noqa: BLE001 -- whatever went wrong, the collectives still work
                why = repr(e)
            if not _all_ranks_ok(why is None, device):
                why = why or "the peer transport self-test failed on another rank"
        if why is None:
            used = "peer"
        else:
            if transport == "peer":
                raise RuntimeError(f"--transport peer: {why}")
            shard.heap = None
            used = f"rccl (peer transport unavailable: {why})"
    shard.stats = {"posted": 0, "overlapped": 0, "blocking": 0, "device_side": 0}
    return shard, used


def start_line_insurance(line):
    """A helper process that prints `line` (the finished measurement) if this process dies before printing one itself: it blocks reading a
    pipe, exits silently on "done", and prints when the pipe closes without it.  Returns the Popen handle (None if it cannot be started)."""
    import subprocess
    try:
        p = subprocess.Popen([sys.executable, "-c",
                              "import sys\nline = sys.argv[1]\nmsg = sys.stdin.read()\nif 'done' not in msg:\n    print(line, flush=True)\n",
                              json.dumps(line)], stdin=subprocess.PIPE, text=True, close_fds=True)
        return p
    except OSError:
        return None


def cancel_line_insurance(p):
    if p is None or p.stdin is None or p.stdin.closed:
        return
    try:
        p.stdin.write("done")
        p.stdin.close()
        p.wait(timeout=10)
    except Exception:  # noqa: BLE001
        pass


def promote_frame_sharded(line, fs, world):
    """`--shard auto`, frames >= 2 x GPUs: the frame-sharded clip becomes the primary number (one clip, strong scaling) and the
    one-clip-per-GPU measurement taken first stays in the line as `clips_dp` -- provided the sharded job completed, is finite, and the
    N GPUs working on ONE clip are not slower than ONE GPU working on it (clips value / N): an exchange path that bad is reported
    (`frame_sharded`, `frame_sharded_not_promoted`), not made the headline."""
    if "error" in fs or not fs.get("outputs_finite"):
        line["frame_sharded_not_promoted"] = "did not complete"
        return False
    if fs["value"] < line["value"] / world:
        line["frame_sharded_not_promoted"] = (f"{world} GPUs on one clip ({fs['value']:.3f} frames/s) slower than one GPU on it "
                                              f"({line['value'] / world:.3f} frames/s)")
        return False
    line["clips_dp"] = {k: line[k] for k in ("value", "ms_per_step", "scaling")}
    line["clips_dp"]["parallelism"] = line["config"]["parallelism"]
    line.update(value=fs["value"], ms_per_step=fs["ms_per_job"], scaling="strong")
    line["config"]["parallelism"] = f"{world}-way frame-sharded clip"
    # the definition of `value` changed with the promotion: say so in the metric itself, and keep BOTH numbers as top-level fields
    # under fixed names (value_clips_dp / value_frame_sharded, set by the caller) so that lines stay comparable across rounds
    line["metric"] += f" [value = ONE clip, frames sharded over {world} GPUs (strong scaling); one clip per GPU: value_clips_dp]"
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1, help="timed jobs")
    ap.add_argument("--warmup", type=int, default=1, help="untimed warm-up jobs")
    ap.add_argument("--no-kernel-breakdown", action="store_true", help="skip the extra, untimed job that brackets every conv / GEMM / capture / inject launch")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--warmup-ddim-steps", type=int, default=0, help="DDIM steps of the warm-up jobs (0 = same as timed)")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--latent-size", type=int, default=64,
                    help="latent height = width (64 = 512^2 frames, the judged configuration; 72 = the 576^2 frames of BASELINE cfg5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-k", type=int, default=2, help="DDIM steps of each kind in the CPU-oracle sample (after one warm-up step)")
    ap.add_argument("--cpu-cfg1", action="store_true",
                    help="also run BASELINE cfg1 (8 f x 256^2 x 10 steps) IN FULL on the CPU oracle (minutes of host time) -> cpu_baseline.cfg1_full")
    ap.add_argument("--cfg1", action="store_true",
                    help="BASELINE configs[0] instead of the judged cfg2: config/low_resource_teaser (8 f x 256^2, 10 DDIM steps, model config "
                         "{lora 160, SparseCausalAttention_index ['mid'], least_sc_channel 640}, Refine + Reweight x10, no blend) -- the case "
                         "--cpu-cfg1 measures in full on the CPU")
    ap.add_argument("--n-edit", type=int, default=1, choices=[1, 2],
                    help="edits per inversion in the timed jobs: 1 = the primary metric (SURVEY 8d); 2 = both prompts of the config")
    ap.add_argument("--no-n-edit2-probe", action="store_true",
                    help="skip the extra, untimed-region job that measures the config-faithful n_edit = 2 job beside the primary")
    ap.add_argument("--shard", choices=["auto", "clips", "frames"], default="auto",
                    help="N > 1: 'clips' = one clip per GPU (weak scaling, latents only on the wire); 'frames' = ONE clip's frames split "
                         "over the GPUs (strong scaling; GroupNorm / K-V / temporal exchanges over RCCL: SURVEY 8e's natural axis); 'auto' "
                         "(default) measures clips first, then the frame-sharded clip (K timed jobs under a watchdog), and reports the "
                         "frame-sharded number as `value` when frames >= 2 x GPUs, it completed and was not slower than ONE GPU on the clip -- the "
                         "other one rides beside it")
    ap.add_argument("--transport", choices=["auto", "peer", "rccl"], default="rccl",
                    help="what carries the exchanges of a frame-sharded clip: 'peer' = one-sided puts into peer-mapped symmetric heaps "
                         "(csrc/peer.hip: no collective call on the data path), 'rccl' = torch.distributed collectives; 'auto' = peer when "
                         "its self-test round trip succeeds on EVERY rank, else rccl (the line says which).  Default rccl: the peer transport "
                         "has met HIP IPC between two processes on one GPU but never xGMI -- opt in with --transport auto / peer")
    ap.add_argument("--peer-heap-gb", type=float, default=2.0, help="symmetric heap per GPU for --transport peer")
    ap.add_argument("--blend-th", type=float, default=None,
                    help="blend_th of the Porsche edit in the TIMED jobs instead of the config's 0.3 (the in-situ PMC pass of scripts/pmc_job.sh "
                         "runs the split-mask job with it, so that the inject class it attributes really reads stored rows); the line says so")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration (fixed flash launch, 1 GiB copy) and the clock / power sampler")
    ap.add_argument("--split-steps", type=int, default=10, help="DDIM steps of the short jobs the split threshold is bisected on")
    ap.add_argument("--split-iters", type=int, default=10, help="bisection steps of the split threshold")
    ap.add_argument("--no-split-mask", action="store_true",
                    help="skip the threshold sweep + the extra job whose blend mask splits the rows (the kernel breakdown then runs with th = 0.3)")
    ap.add_argument("--issue-plans", action="store_true",
                    help="issue the steady-state UNet forwards from recorded native plans (fatezero_amd/issue.py) instead of the Python walk; "
                         "the judged flash launches then carry HIP-event brackets only in the walked / recorded steps of each pass")
    ap.add_argument("--no-frame-shard-probe", action="store_true",
                    help="N > 1, --shard clips: skip the extra frame-sharded job reported under `frame_sharded` (it runs AFTER the clips "
                         "measurement is complete, under a 120 s watchdog that prints the clips line and exits if the exchange path stalls)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # bare `python bench.py --gpus N`: become N ranks
        import subprocess
        sys.exit(subprocess.call(spawn_command(sys.argv[1:], args.gpus)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    n_ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")  # RCCL over xGMI
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    from fatezero_amd import kernels as K
    if os.environ.get("FZ_BENCH_LAUNCHLOG"):
        install_launch_log(K, os.environ["FZ_BENCH_LAUNCHLOG"])
    timer = KernelTimer()
    timer.extra = False
    install_timers(K, timer)

    model_config = None
    if args.cfg1:
        global SRC_PROMPT, TGT_PROMPT, EDIT_KW
        args.latent_size, args.ddim_steps, args.no_n_edit2_probe = 32, 10, True
        model_config = {"lora": 160, "SparseCausalAttention_index": ["mid"], "least_sc_channel": 640}
        SRC_PROMPT = "a silver jeep driving down a curvy road in the countryside"
        TGT_PROMPT = "watercolor painting of a silver jeep driving down a curvy road in the countryside"
        EDIT_KW = dict(cross_replace_steps={"default_": 0.8}, self_replace_steps=0.8, use_inversion_attention=True,
                       is_replace_controller=False, eq_params={"words": ["watercolor"], "values": [10]}, save_self_attention=False,
                       guidance_scale=7.5)
    pipe = build_pipeline(device, seed=0, model_config=model_config)
    if args.issue_plans:
        pipe.unet.enable_issue_plans()
    by_frames = args.shard == "frames" and world > 1
    auto_frames = args.shard == "auto" and world > 1 and args.frames >= 2 * world  # the judged single clip: frames are the natural axis
    g = torch.Generator().manual_seed(1234 + (0 if by_frames else rank))  # frame-sharded: every rank holds the same clip
    L = args.latent_size
    z0 = torch.randn(1, 4, args.frames, L, L, generator=g).to(device)
    if dist is not None:  # every rank must have built the same model: compare a weight checksum over RCCL
        chk = torch.stack([p.float().sum() for p in list(pipe.unet.parameters())[:8]]).sum().reshape(1)
        ref = chk.clone()
        dist.broadcast(ref, 0)
        assert torch.allclose(ref, chk), "ranks built different weights"

    transport_used = None
    if by_frames:
        from fatezero_amd import dist as fz_dist
        pipe.frame_shard, transport_used = make_frame_shard(fz_dist, args.frames, args.transport, args.peer_heap_gb, device)
    box = measure_box(K, device) if not args.no_box else None  # BEFORE the warm-up: the chip as the job will find it
    wsteps = args.warmup_ddim_steps or args.ddim_steps
    for _ in range(args.warmup):
        run_job(pipe, z0, wsteps, device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    timer.enabled = True
    sampler = SmiSampler(local_rank).start() if (rank == 0 and not args.no_box) else None
    barrier()
    t0 = time.perf_counter()
    edited = None
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # job boundaries on the launch stream: no sync inside
    marks[0].record()
    for i in range(args.steps):
        edited = run_job(pipe, z0, args.ddim_steps, device, args.n_edit, blend_th=args.blend_th)  # only the judged flash launches carry event brackets here
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    smi = sampler.stop() if sampler is not None else None
    box_hot = measure_box(K, device, hot=True) if not args.no_box else None  # straight behind the timed jobs: the chip still in the job's state
    per_job_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    n_edit2 = None
    if args.n_edit == 1 and not args.no_n_edit2_probe and not by_frames:
        # the config-faithful job (1 inversion + BOTH prompts of jeep_posche.yaml), one run after the timed region, reported beside
        # the primary: frames/s = F * n_edit / (t_inversion + sum t_edit)  
        timer.enabled = False
        run_job(pipe, z0, min(2, args.ddim_steps), device, 2)  # the second controller's plans / constants: warm
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e2 = run_job(pipe, z0, args.ddim_steps, device, 2)
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t1
        n_edit2 = {"n_edit": 2, "ms_per_job": d2 * 1e3, "value": 2 * args.frames / d2, "unit": "frames/s",
                   "outputs_finite": bool(torch.isfinite(e2.float()).all()),
                   "what": "1 capture inversion + 2 CFG edits (p2p_config 0: Refine + Reweight + blend, p2p_config 1: Replace + blend)"}
        timer.enabled = True
    split = None
    if not args.no_kernel_breakdown and not args.cfg1 and not by_frames and not args.no_split_mask:
        # a blend threshold under which the masked-inject launches really read stored rows (pick_split_threshold), and ONE job with it,
        # timed like the primary (no event brackets besides the flash ones), beside the config-faithful th = 0.3 of the timed region
        timer.enabled = False
        th_split, seen = pick_split_threshold(pipe, z0, device, steps=args.split_steps, iters=args.split_iters)
        if th_split is not None:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            es = run_job(pipe, z0, args.ddim_steps, device, blend_th=th_split)
            torch.cuda.synchronize()
            ds = time.perf_counter() - t1
            split = {"blend_th": th_split, "ms_per_job": ds * 1e3, "value": args.frames / ds, "unit": "frames/s",
                     "stored_rows_fraction": stored_rows_fraction(pipe), "outputs_finite": bool(torch.isfinite(es.float()).all()),
                     "threshold_sweep_stored_rows_fraction": {str(k): v for k, v in seen.items()},
                     "what": "the primary job with blend_th raised so that the blend mask SPLITS the self-attention rows (the config's 0.3 "
                             "leaves ~all rows live with procedural weights: its masked-inject launches read no stored map)"}
        timer.enabled = True
    if not args.no_kernel_breakdown:  # (every rank: a frame-sharded job has collectives inside)
        # the other kernels' event brackets (22 k launches per job, one barrier packet each: ~3 % on the job) go on ONE extra job
        # after the timed region; its flash launches are not counted.  It runs with the split threshold: the inject entry then
        # measures launches that read stored rows (every other class launches the same kernels on the same shapes either way)
        n_flash = len(timer.events)
        timer.extra = True
        issuer, pipe.unet._issuer = pipe.unet._issuer, None  # (the brackets sit in the Python wrappers: this job is walked)
        run_job(pipe, z0, args.ddim_steps, device, blend_th=None if split is None else split["blend_th"])
        torch.cuda.synchronize()
        pipe.unet._issuer = issuer
        timer.extra = False
        timer.events = timer.events[:n_flash] + [ev for ev in timer.events[n_flash:] if ev[0][0] != "flash"]
    timer.enabled = False
    barrier()
    tmax = torch.tensor([dt], device=device)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if not by_frames:  # (a frame-sharded job already returned the all-gathered latents of its one clip)
            gathered = [torch.empty_like(edited) for _ in range(world)]
            dist.all_gather(gathered, edited.contiguous())  # edited latents -> rank 0 (and everyone) over xGMI
            edited = torch.cat(gathered, dim=0)
    dt = float(tmax.item())
    finite = bool(torch.isfinite(edited.float()).all())

    line = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = (1 if by_frames else world) * args.frames * args.n_edit * args.steps / dt
        roof, others = rooflines(timer.summary())
        px = 8 * L
        judged = args.frames == 8 and L == 64 and args.ddim_steps == 50 and not args.cfg1
        line = {"metric": "edited frames/sec (8f x 512^2 x 50 DDIM steps: capture inversion + 1 CFG edit, latents in/out)" if judged else
                          f"edited frames/sec ({args.frames}f x {px}^2 x {args.ddim_steps} DDIM steps: capture inversion + 1 CFG edit, latents in/out)",
                "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if by_frames else "weak",
                "vs_baseline": None,
                "dtype": "fp16", "data": "synthetic",
                "config": {"workload": ("config/low_resource_teaser (BASELINE configs[0]) shape: " if args.cfg1 else "config/teaser/jeep_posche.yaml shape: ") +
                                       f"{args.frames} x {px}x{px} (latents {args.frames}x{L}x{L}x4), "
                                       f"{args.ddim_steps}-step DDIM inversion with HBM map capture + {args.ddim_steps}-step "
                                       + ("CFG edit (Refine + Reweight x10, index ['mid'], least_sc_channel 640), " if args.cfg1 else
                                          "CFG edit (Replace, blend-masked self-attention), ") + "SD-1.x pseudo-3D UNet lora=160, "
                                       "random-init weights",
                           "frames": args.frames, "ddim_steps": args.ddim_steps, "n_edit": args.n_edit,
                           **({"blend_th_override": args.blend_th} if args.blend_th is not None else {}),
                           "parallelism": ("single GPU" if world == 1 else
                                           f"{world}-way frame-sharded clip" if by_frames else f"dp{world} over clips"),
                           "arena_GB": pipe.store_controller.arena_bytes / 1e9, "outputs_finite": finite,
                           "n_ranks_seen": n_ranks_seen, **({"transport": transport_used} if transport_used else {})},
                "roofline": roof, "rooflines": others, "cpu_baseline": None}
        n = len(per_job_ms)
        line["ms_per_step_spread"] = {"min": per_job_ms[0], "median": per_job_ms[n // 2], "max": per_job_ms[-1], "jobs": n,
                                      "how": "HIP events at the job boundaries of the timed region (launch stream, no sync inside)"}
        if box is not None:
            box["during_timed_region"] = smi
            box.update(box_hot or {})
            line["box"] = box
            if box.get("flash_calib_hot_us"):
                # the headline as it would read on a box whose hot calibration launch takes FLASH_CALIB_REF_US (first order: the job
                # follows the chip's sustained clock under the job's power draw, which is what the hot calibration launch measures)
                line["value_normalised"] = value * box["flash_calib_hot_us"] / FLASH_CALIB_REF_US
        if split is not None:
            line["split_mask_job"] = split
        if pipe.unet._issuer is not None:
            line["issue_plans"] = {k: (v if not isinstance(v, list) else v[:4]) for k, v in pipe.unet._issuer.stats.items()}
        if n_edit2 is not None:
            line["config_faithful_n_edit_2"] = n_edit2
        if not args.no_cpu_baseline and world == 1 and L == 64:  # (the oracle sample is a 512x512 clip)
            try:
                line["cpu_baseline"] = cpu_baseline(pipe, args.ddim_steps, args.frames, k=args.cpu_k)
                if args.cpu_cfg1:
                    line["cpu_baseline"]["cfg1_full"] = cpu_cfg1_full(pipe)
            except Exception as e:  # the baseline is a report, never a reason to lose the measurement
                line["cpu_baseline"] = {"error": repr(e)}

    # ---- N > 1, clips mode: one more job with ONE clip's frames split over the ranks (SURVEY 8e's natural split), guarded:
    #      if the exchange path stalls the clips line above is still printed (every rank runs the same watchdog)
    if dist is not None and not by_frames and not args.no_frame_shard_probe and args.frames >= world:
        import threading
        done = threading.Lock()

        insurance = None

        def bail():
            if done.acquire(blocking=False):
                if rank == 0:
                    line["frame_sharded"] = {"error": "frame-sharded probe exceeded its time limit"}
                    cancel_line_insurance(insurance)
                    print(json.dumps(line), flush=True)
                os._exit(0)
        dog = threading.Timer(120.0 + (12.0 * args.steps if auto_frames else 0.0), bail)
        dog.daemon = True
        dog.start()
        # the frame-sharded path has never run on more than one GPU: a rank that dies in it (GPU fault -> abort() from a runtime thread,
        # then the launcher's SIGTERM / SIGKILL to the others) must not take the finished clips measurement with it.  Python-level signal
        # handlers do not run while the main thread sits in a HIP call, so the insurance is a helper PROCESS that holds the clips line
        # and prints it when rank 0's end of the pipe closes without a "done" (start_line_insurance); handlers cover the polite cases.
        insurance = start_line_insurance(dict(line, frame_sharded={"error": "rank 0 died inside the frame-sharded probe"})) if rank == 0 else None
        import signal
        for sig in (signal.SIGTERM, signal.SIGABRT):
            try:
                signal.signal(sig, lambda *_: bail())
            except (ValueError, OSError):
                pass
        fs = None
        try:
            from fatezero_amd import dist as fz_dist
            zc = torch.randn(1, 4, args.frames, L, L, generator=torch.Generator().manual_seed(1234)).to(device)
            pipe.frame_shard, transport_used = make_frame_shard(fz_dist, args.frames, args.transport, args.peer_heap_gb, device)
            run_job(pipe, zc, 2, device)  # warm-up: RCCL channels, allocator
            njobs = args.steps if auto_frames else 1
            barrier()
            t0 = time.perf_counter()
            for _ in range(njobs):
                out = run_job(pipe, zc, args.ddim_steps, device)
            barrier()
            tf = torch.tensor([time.perf_counter() - t0], device=device)
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            st = pipe.frame_shard.stats
            fs = {"value": args.frames * njobs / float(tf.item()), "unit": "frames/s", "ms_per_job": float(tf.item()) * 1e3 / njobs,
                  "jobs_timed": njobs, "scaling": "strong", "parallelism": f"{world}-way frame-sharded clip ({args.frames} frames)",
                  "outputs_finite": bool(torch.isfinite(out.float()).all()), "transport": transport_used,
                  "exchanges": {"posted": st["posted"], "overlapped_with_compute": st["overlapped"], "blocking": st["blocking"],
                                "device_side": st.get("device_side", 0)}}
            if pipe.frame_shard.heap is not None:
                pipe.frame_shard.heap.check()
        except Exception as e:
            fs = {"error": repr(e)}
        if done.acquire(blocking=False):
            dog.cancel()
            if rank == 0:
                line["frame_sharded"] = fs
                line["value_clips_dp"] = line["value"]  # fixed definitions, whatever `value` ends up meaning (see `metric`)
                line["value_frame_sharded"] = fs.get("value")
                if auto_frames:
                    promote_frame_sharded(line, fs, world)
        else:
            return
    if rank == 0:
        cancel_line_insurance(locals().get("insurance"))
        print(json.dumps(line), flush=True)
    if dist is not None:
        # the line is out: a rank stuck in teardown (a peer that died in the probe) must not keep the launcher waiting
        import threading
        bye = threading.Timer(30.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
