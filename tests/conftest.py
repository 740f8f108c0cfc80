import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch sizes its OpenMP pool by the machine, not by the container's CPU quota (128 threads for a 16-core quota on the GPU
    # boxes): the oracle runs get throttled by the cgroup instead of running faster
    import torch
    torch.set_num_threads(_usable_cores())
    # a fresh checkout has no built library (it is git-ignored): build it once, as __graft_entry__.build() does
    lib = os.path.join(ROOT, "ase_amd", "csrc", "libase_hip.so")
    if not os.path.exists(lib):
        import shutil
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            import __graft_entry__
            __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
