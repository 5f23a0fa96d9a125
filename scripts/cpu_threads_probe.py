#!/usr/bin/env python3
"""How many torch threads the host of the GPU box really has: affinity / cgroup quota, and the time of the two operators that dominate the
CPU oracle (softmax + baddbmm of the 64^2 self-attention level, a 3x3 convolution) at several thread counts."""
import os, time, torch
import torch.nn.functional as F
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
try:
    print(open("/proc/loadavg").read().strip())
except OSError:
    pass
q = torch.randn(24, 4096, 40); k = torch.randn(24, 8192, 40)
x = torch.randn(3, 320, 64, 64); w = torch.randn(320, 320, 3, 3)
for nt in (128, 64, 32, 16, 8):
    torch.set_num_threads(nt)
    for rep in range(2):
        t0 = time.time()
        s = torch.baddbmm(torch.empty(24, 4096, 8192), q, k.transpose(1, 2), beta=0, alpha=0.158)
        t1 = time.time()
        p = s.softmax(-1)
        t2 = time.time()
        y = F.conv2d(x, w, padding=1)
        t3 = time.time()
    print(f"threads {nt:3d}: baddbmm {t1 - t0:.2f} s, softmax {t2 - t1:.2f} s, conv {1e3 * (t3 - t2):.1f} ms")
    del s, p
