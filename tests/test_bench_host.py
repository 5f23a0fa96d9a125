"""Host-side pieces of bench.py that do not need a GPU: the self-spawn command of a bare `python bench.py --gpus N`
(the driver's N = 1 invocation is bare; an N > 1 invocation of the same form must become N ranks on its own) and the
roofline bookkeeping."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_spawn_command_is_a_torchrun_of_this_file():
    cmd = bench.spawn_command(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    # a free port is picked when none is given
    p = int(bench.spawn_command([], 2)[bench.spawn_command([], 2).index("--master-port") + 1])
    assert 1024 < p < 65536


def test_spawned_ranks_see_world_size(tmp_path):
    """The spawn command really brings up N ranks with RANK / WORLD_SIZE / MASTER_* set (a stand-in script replaces bench.py:
    there is no GPU here)."""
    import subprocess
    probe = tmp_path / "probe.py"
    probe.write_text("import os\nopen(os.path.join(os.path.dirname(__file__), 'r' + os.environ['RANK']), 'w')"
                     ".write(os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_ADDR'])\n")
    cmd = bench.spawn_command([], 2)
    cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = str(probe)
    subprocess.run(cmd, check=True, timeout=120)
    assert (tmp_path / "r0").read_text() == "2 127.0.0.1" and (tmp_path / "r1").read_text() == "2 127.0.0.1"


def test_rooflines_bookkeeping():
    # 4th field: kv slots really contracted -- frames 0, 1 of every 8-frame clip read their single source frame once (14 of 16 slots)
    summ = {("flash", 8, 2, 14): {"launches": 10, "avg_ms": 0.5, "total_ms": 5.0},
            ("flash", 16, 2, 28): {"launches": 10, "avg_ms": 1.0, "total_ms": 10.0},
            ("conv3x3", 1e12, 5e8): {"launches": 4, "avg_ms": 1.0, "total_ms": 4.0},
            ("gemm_hbm", 6.4e7, 6.4e7): {"launches": 2, "avg_ms": 0.02, "total_ms": 0.04},
            ("gemm_mfma", 4e11, 1e8): {"launches": 2, "avg_ms": 0.5, "total_ms": 1.0},
            ("capture", 268435456, 268435456): {"launches": 2, "avg_ms": 0.1, "total_ms": 0.2},
            # a masked inject whose blend mask is all ones reads NO stored row (4th field 0): must not divide by zero anywhere
            ("inject", 268435456, 268435456, 0): {"launches": 2, "avg_ms": 0.1, "total_ms": 0.2},
            ("inject", 268435456, 268435456, 67108864): {"launches": 2, "avg_ms": 0.1, "total_ms": 0.2}}
    roof, others = bench.rooflines(summ)
    flops = 4.0 * 4096 * 8192 * 320 * (8 + 16) * 10
    assert abs(roof["achieved"] - flops / 15e-3 / 1e12) < 1e-6 and roof["peak"] == 2500.0 and roof["bound"] == "mfma"
    assert abs(roof["frac"] - roof["achieved"] / 2500.0) < 1e-12 and roof["launches"] == 20
    assert abs(roof["contracted_fraction"] - 0.875) < 1e-12
    assert abs(roof["achieved_over_contracted_tiles"] - 0.875 * roof["achieved"]) < 1e-9
    conv = [o for o in others if "MODE 1" in o["kernel"]][0]
    assert abs(conv["achieved"] - 1000.0) < 1e-6 and conv["bound"] == "mfma" and conv["algorithmic_bytes_per_launch"] == 5e8
    # projection GEMMs are reported per regime: the K <= 640 plain projections against the HBM roof, GEGLU / long-K against MFMA
    gh = [o for o in others if "HBM class" in o["kernel"]][0]
    gm = [o for o in others if "MFMA class" in o["kernel"]][0]
    assert gh["bound"] == "hbm" and gh["peak"] == 8000.0 and abs(gh["achieved"] - 2 * 6.4e7 / 0.04e-3 / 1e9) < 1e-6
    assert gm["bound"] == "mfma" and abs(gm["achieved"] - 2 * 4e11 / 1e-3 / 1e12) < 1e-6 and abs(gm["algorithmic_GBps"] - 2e8 / 1e-3 / 1e9) < 1e-6
    inj = [o for o in others if "INJECT" in o["kernel"]][0]
    # the inject entry is priced on the stored rows really read (mask 0); the whole-map figure rides beside it
    assert abs(inj["stored_rows_fraction"] - 0.125) < 1e-12 and abs(inj["achieved"] - 0.125 * inj["achieved_whole_map"]) < 1e-9
    assert abs(inj["achieved_whole_map"] - 4 * 268435456 / 0.4e-3 / 1e9) < 1e-6 and abs(inj["frac"] - inj["achieved"] / 8000.0) < 1e-12
    assert inj["algorithmic_bytes_per_launch"] == 2 * 67108864 / 4
    cap = [o for o in others if "CAPTURE" in o["kernel"]][0]
    assert abs(cap["achieved"] - 2 * 268435456 / 0.2e-3 / 1e9) < 1e-6 and cap["peak"] == 8000.0


def test_gemm_roofline_classes():
    """K <= 640 plain projections -> HBM class (algorithmic bytes incl. residuals); GEGLU and K >= 1280 -> MFMA class; < 1024 rows: none."""
    import torch
    x = torch.zeros(2048, 320, dtype=torch.float16)
    w = torch.zeros(640, 320, dtype=torch.float16)
    c = bench.gemm_class(x, w, {})
    assert c[0] == "gemm_hbm" and c[1] == c[2] == 2.0 * (2048 * 320 + 2048 * 640 + 320 * 640)
    c = bench.gemm_class(x, w, {"res": x})
    assert c[2] == 2.0 * (2048 * 320 + 2 * 2048 * 640 + 320 * 640)
    c = bench.gemm_class(x, torch.zeros(2560, 320, dtype=torch.float16), {"geglu": True})
    assert c[0] == "gemm_mfma" and c[1] == 2.0 * 2048 * 320 * 2560 and c[2] == 2.0 * (2048 * 320 + 2048 * 1280 + 320 * 2560)
    assert bench.gemm_class(torch.zeros(2048, 1280, dtype=torch.float16), torch.zeros(320, 1280, dtype=torch.float16), {})[0] == "gemm_mfma"
    assert bench.gemm_class(torch.zeros(512, 320, dtype=torch.float16), w, {}) is None


def test_flash_timer_counts_the_kv_slots_really_contracted(monkeypatch):
    """The roofline's `contracted_fraction`: under [-1, 'first'] frames 0 and 1 of every clip resolve both slots to frame 0 and are
    read once by the kernel (csrc/attn_flash.hip); ['mid'] and three-slot lists with a distinct member are read in full."""
    import types
    import torch
    from fatezero_amd import kernels as K
    tags = []

    class T(bench.KernelTimer):
        def wrap(self, module, fn_name, select):
            if fn_name == "attn_self":
                self.select = select
    fake = types.SimpleNamespace(FZ_ATTN_FLASH=K.FZ_ATTN_FLASH, FZ_ATTN_CAPTURE=K.FZ_ATTN_CAPTURE, kv_slots=K.kv_slots,
                                 attn_self=None, conv3x3=None, gemm=None)
    t = T()
    t.extra = False
    bench.install_timers(fake, t)
    q = torch.empty(16, 4096, 320, device="meta")
    assert t.select(q, None, None, None, clip_len=8, heads=8, index_list=[-1, "first"], n_frames=16) == ("flash", 16, 2, 28)
    assert t.select(q, None, None, None, clip_len=8, heads=8, index_list=[-1, "first"], frame0=8, n_frames=8) == ("flash", 8, 2, 14)
    assert t.select(q, None, None, None, clip_len=8, heads=8, index_list=[-1, "first"], frame0=2, n_frames=6) == ("flash", 6, 2, 12)
    assert t.select(q, None, None, None, clip_len=16, heads=8, index_list=["mid"], n_frames=16) == ("flash", 16, 1, 16)
    assert t.select(q, None, None, None, clip_len=4, heads=8, index_list=[-1, "first", 1], n_frames=4) == ("flash", 4, 3, 12)
    assert t.select(q, None, None, None, clip_len=4, heads=8, index_list=[-1, "first", "first"], n_frames=4) == ("flash", 4, 3, 8)


def _run_bench_ranks(world, extra, timeout=600):
    """bench.main() under torch.distributed.run with `world` CPU ranks (tests/bench_cpu_harness.py); returns rank 0's JSON line."""
    import json
    import subprocess
    cmd = bench.spawn_command(["--gpus", str(world), "--steps", "1", "--warmup", "0", "--ddim-steps", "2", "--frames", "2",
                               "--latent-size", "8", "--no-cpu-baseline"] + extra, world)
    cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = os.path.join(ROOT, "tests", "bench_cpu_harness.py")
    out = subprocess.run(cmd, check=True, timeout=timeout, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr  # ONE line, from rank 0 only
    return json.loads(lines[0])


def test_bench_main_two_ranks_clips_and_frame_shard_probe():
    """Default sharding (one clip per rank, weak scaling) + the frame-sharded probe that follows it, end to end over gloo."""
    line = _run_bench_ranks(2, ["--no-kernel-breakdown", "--transport", "rccl"])  # (the per-kernel event brackets are single-rank bookkeeping: test_rooflines_bookkeeping)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["n_ranks_seen"] == 2
    assert line["config"]["parallelism"] == "dp2 over clips" and line["config"]["outputs_finite"] is True
    assert line["value"] > 0 and abs(line["value"] - 2 * 2 * 1 / (line["ms_per_step"] / 1e3)) < 1e-6 * line["value"]  # whole-job frames/s
    assert line["steps"] == 1 and line["warmup"] == 0 and line["higher_is_better"] is True and line["cpu_baseline"] is None
    # the config-faithful job (1 inversion + both prompts of the YAML) is measured beside the primary
    # the normaliser and the spread of the timed jobs ride in the line (the harness stands in for the calibration launch: 910 us, cold and hot)
    assert line["box"]["flash_calib_us"] == 910.0 and line["box"]["flash_calib_hot_us"] == 910.0
    assert abs(line["value_normalised"] - line["value"] * 910.0 / bench.FLASH_CALIB_REF_US) < 1e-9 * line["value"]
    assert line["box"]["during_timed_region"]["samples"] >= 0
    sp = line["ms_per_step_spread"]
    assert sp["jobs"] == 1 and sp["min"] == sp["median"] == sp["max"] and 0.5 * line["ms_per_step"] < sp["min"] <= line["ms_per_step"] * 1.001
    ne2 = line["config_faithful_n_edit_2"]
    assert ne2["n_edit"] == 2 and ne2["outputs_finite"] is True and abs(ne2["value"] - 2 * 2 / (ne2["ms_per_job"] / 1e3)) < 1e-6 * ne2["value"]
    fs = line["frame_sharded"]
    assert "error" not in fs, fs
    assert fs["scaling"] == "strong" and fs["outputs_finite"] is True and fs["value"] > 0


def test_bench_main_two_ranks_transport_auto_is_the_peer_heaps():
    """`--transport auto` (opt-in: the default is rccl until the peer transport has met xGMI): the frame-sharded clip runs over the one-sided
    peer transport (csrc/peer.hip; shared-memory heaps under the CPU harness) once its self-test round trip succeeded ON EVERY RANK
    (agreed by an all_reduce) -- the line says so, and every exchange is device-side: no collective call and no blocking wait inside the UNet."""
    line = _run_bench_ranks(2, ["--frames", "4", "--no-kernel-breakdown", "--no-n-edit2-probe", "--peer-heap-gb", "0.05", "--transport", "auto"])
    fs = line["frame_sharded"]
    assert "error" not in fs and fs["outputs_finite"] is True and fs["transport"] == "peer", fs
    ex = fs["exchanges"]
    assert ex["device_side"] == ex["posted"] > 0 and ex["blocking"] == 0 and ex["overlapped_with_compute"] == 0, ex
    # frames >= 2 x ranks: `--shard auto` reports the frame-sharded clip as `value` (strong scaling, K timed jobs) and keeps the
    # one-clip-per-rank measurement beside it, under fixed-definition field names as well
    assert fs["jobs_timed"] == 1 and line["value"] == fs["value"] and line["ms_per_step"] == fs["ms_per_job"] and line["scaling"] == "strong"
    assert line["config"]["parallelism"] == "2-way frame-sharded clip" and "frame_sharded_not_promoted" not in line
    assert line["clips_dp"]["scaling"] == "weak" and line["clips_dp"]["parallelism"] == "dp2 over clips" and line["clips_dp"]["value"] > 0
    assert line["value_clips_dp"] == line["clips_dp"]["value"] and line["value_frame_sharded"] == fs["value"]


def test_frame_sharded_promotion_rule():
    def mk():
        return {"metric": "edited frames/sec", "value": 6.8, "ms_per_step": 2350.0, "scaling": "weak", "config": {"parallelism": "dp2 over clips"}}
    good = {"value": 4.7, "ms_per_job": 1700.0, "outputs_finite": True}
    line = mk()
    assert bench.promote_frame_sharded(line, good, 2) is True
    assert line["value"] == 4.7 and line["ms_per_step"] == 1700.0 and line["scaling"] == "strong"
    assert "frames sharded over 2 GPUs" in line["metric"] and "value_clips_dp" in line["metric"]  # the changed definition is named
    assert line["clips_dp"] == {"value": 6.8, "ms_per_step": 2350.0, "scaling": "weak", "parallelism": "dp2 over clips"}
    assert line["config"]["parallelism"] == "2-way frame-sharded clip"
    line = mk()  # two GPUs on one clip slower than one GPU on it: reported, not promoted
    assert bench.promote_frame_sharded(line, {"value": 3.0, "ms_per_job": 2666.0, "outputs_finite": True}, 2) is False
    assert line["value"] == 6.8 and line["scaling"] == "weak" and "slower than one GPU" in line["frame_sharded_not_promoted"]
    for bad in ({"error": "x"}, {"value": 9.0, "ms_per_job": 1.0, "outputs_finite": False}):
        line = mk()
        assert bench.promote_frame_sharded(line, bad, 2) is False and line["value"] == 6.8 and "clips_dp" not in line


def test_bench_main_two_ranks_frames_mode():
    line = _run_bench_ranks(2, ["--shard", "frames", "--no-kernel-breakdown"])
    assert line["scaling"] == "strong" and line["config"]["parallelism"] == "2-way frame-sharded clip"
    assert abs(line["value"] - 2 * 1 / (line["ms_per_step"] / 1e3)) < 1e-6 * line["value"]  # ONE clip's frames over the job time
    assert line["config"]["outputs_finite"] is True


def test_line_insurance_prints_the_finished_measurement_when_rank0_dies():
    """bench.py guards the never-measured multi-GPU probe with a helper process holding the finished clips line: it prints the line if
    rank 0 dies without a word (abort() from a GPU fault cannot be caught by Python signal handlers), and stays silent after `done`."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import bench
        p = bench.start_line_insurance({"value": 1.5, "who": "insured"})
        if sys.argv[1] == "die":
            os._exit(9)
        bench.cancel_line_insurance(p)
    ''' % ROOT)
    died = subprocess.run([sys.executable, "-c", code, "die"], capture_output=True, text=True, timeout=60)
    assert died.returncode == 9 and '"who": "insured"' in died.stdout, (died.stdout, died.stderr)
    ok = subprocess.run([sys.executable, "-c", code, "ok"], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and ok.stdout.strip() == "", (ok.stdout, ok.stderr)


def test_bench_main_single_rank_with_the_kernel_breakdown_job():
    """The default single-GPU path INCLUDING the extra job that brackets the conv / GEMM / capture / inject launches with events and
    the roofline bookkeeping behind it (the multi-rank tests above skip it): one rank, tiny model, CPU emulation."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_cpu_harness.py"), "--steps", "1", "--warmup", "0", "--ddim-steps", "2",
           "--frames", "2", "--latent-size", "8", "--no-cpu-baseline", "--split-steps", "2", "--split-iters", "3"]   # (the bisection: 3 short jobs, not 10)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, check=True, timeout=600, capture_output=True, text=True, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and isinstance(line["rooflines"], list) and line["config"]["outputs_finite"] is True
    assert line["config_faithful_n_edit_2"]["outputs_finite"] is True


def test_inject_timer_prices_the_rows_really_read():
    """A masked inject reads only the stored rows whose blend mask is 0: the timer tag carries the map bytes (SURVEY 8(d)'s figure) and,
    as a 4th field, the bytes of the rows really read -- 0 for an all-ones mask (procedural weights at th = 0.3), which must survive the
    roofline bookkeeping (it once divided by it)."""
    import types
    import torch
    from fatezero_amd import kernels as K

    class T(bench.KernelTimer):
        def wrap(self, module, fn_name, select):
            if fn_name == "attn_self":
                self.select = select
    fake = types.SimpleNamespace(FZ_ATTN_FLASH=K.FZ_ATTN_FLASH, FZ_ATTN_CAPTURE=K.FZ_ATTN_CAPTURE, FZ_ATTN_INJECT=K.FZ_ATTN_INJECT,
                                 kv_slots=K.kv_slots, attn_self=None, conv3x3=None, gemm=None, gemm_qkvt=None)
    t = T()
    t.extra = True
    bench.install_timers(fake, t)
    q = torch.empty(16, 1024, 640, device="meta")
    p = torch.empty(8, 8, 1024, 2048, dtype=torch.float16, device="meta")
    per_frame = 8 * 1024 * 2048 * 2
    ones = torch.ones(16, 1024)
    tag = t.select(q, None, None, None, mode=K.FZ_ATTN_INJECT, p=p, n_frames=8, frame0=8, row_mask=ones, mask_frame_off=8)
    assert tag == ("inject", 8 * per_frame, 8 * per_frame, 0)
    half = ones.clone()
    half[8:, :256] = 0.0
    tag = t.select(q, None, None, None, mode=K.FZ_ATTN_INJECT, p=p, n_frames=8, frame0=8, row_mask=half, mask_frame_off=8)
    assert tag == ("inject", 8 * per_frame, 8 * per_frame, round(8 * per_frame * 0.25))
    assert t.select(q, None, None, None, mode=K.FZ_ATTN_INJECT, p=p, n_frames=8, frame0=8)[3] == 8 * per_frame  # no mask: every row
    assert t.select(q, None, None, None, mode=K.FZ_ATTN_CAPTURE, p=p, n_frames=8) == ("capture", 8 * per_frame, 8 * per_frame)
    roof, others = bench.rooflines({tag: {"launches": 3, "avg_ms": 0.1, "total_ms": 0.3},
                                    ("inject", 8 * per_frame, 8 * per_frame, 0): {"launches": 3, "avg_ms": 0.1, "total_ms": 0.3}})
    assert roof is None and abs(others[0]["stored_rows_fraction"] - 0.125) < 1e-9
