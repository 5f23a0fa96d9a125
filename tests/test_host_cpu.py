"""oracle/host_cpu.py: the thread pool of the CPU oracle follows the cgroup CPU quota, not the logical CPUs torch sees."""
import builtins
import io

from oracle import host_cpu


def _fake_fs(monkeypatch, files, affinity=256):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup"):
            if path in files:
                return io.StringIO(files[path])
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(host_cpu.os, "sched_getaffinity", lambda pid: set(range(affinity)), raising=False)


def test_cgroup_v2_quota_of_the_gpu_boxes(monkeypatch):
    _fake_fs(monkeypatch, {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"})  # what the MI355X boxes report (profiles/r03_cpu_threads_probe.txt)
    assert host_cpu.cpu_budget() == 16


def test_cgroup_v2_without_a_quota_falls_back_to_the_affinity_mask(monkeypatch):
    _fake_fs(monkeypatch, {"/sys/fs/cgroup/cpu.max": "max 100000\n"}, affinity=24)
    assert host_cpu.cpu_budget() == 24


def test_cgroup_v1_quota_and_fractional_cores(monkeypatch):
    _fake_fs(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "250000\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})
    assert host_cpu.cpu_budget() == 3  # 2.5 cores -> 3 threads
    _fake_fs(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}, affinity=8)
    assert host_cpu.cpu_budget() == 8


def test_pool_never_grows(monkeypatch):
    import torch
    before = torch.get_num_threads()
    _fake_fs(monkeypatch, {"/sys/fs/cgroup/cpu.max": "100000000 100000\n"})
    assert host_cpu.size_torch_pool() == before == torch.get_num_threads()
