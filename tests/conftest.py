import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the CPU oracle must not run 128 threads against a 16-core cgroup quota (oracle/host_cpu.py)
    from oracle.host_cpu import size_torch_pool
    size_torch_pool()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_arena_pools():
    """The map arena recycles released blocks through process-wide pools (MapArena._pool / _host_pool): a test must not inherit what an
    earlier test released."""
    from fatezero_amd.video_diffusion.prompt_attention.attention_store import MapArena
    MapArena.reset_pools()
    yield
    MapArena.reset_pools()
