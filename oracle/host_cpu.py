"""TEST / BENCH INFRASTRUCTURE (like the rest of oracle/): how many host cores the CPU oracle may really use.

torch sizes its thread pool from the logical CPUs it can see.  On the MI355X boxes that is 256 logical CPUs (128 torch threads) behind a
cgroup CPU quota of 16 cores (`/sys/fs/cgroup/cpu.max` = "1600000 100000", profiles/r03_cpu_threads_probe.txt): 128 threads time-slicing
16 cores' worth of quota run the many small operators of the oracle UNet slower than 16 threads do.  `cpu_budget()` is the smaller of the
affinity mask and the quota; tests/conftest.py and bench.py's cpu_baseline leg size the pool with it and report it as `cores`.
"""
import math
import os


def _cgroup_quota():
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:  # cgroup v1
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def cpu_budget() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = _cgroup_quota()
    if quota is not None:
        n = min(n, max(1, int(math.ceil(quota))))
    return max(1, n)


def size_torch_pool() -> int:
    """Never more torch threads than the host lets this process run; returns the pool size in force."""
    import torch
    n = min(torch.get_num_threads(), cpu_budget())
    torch.set_num_threads(n)
    return n
