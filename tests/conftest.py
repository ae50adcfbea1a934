import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_threads() -> int:
    """Threads for the CPU oracle: the cgroup quota if there is one (a 256-thread pool on a 16-core quota thrashes)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return min(n, 64)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    torch.set_num_threads(_cpu_threads())


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")
