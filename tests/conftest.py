import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
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
