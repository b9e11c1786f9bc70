import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("utils_", "loss_", "producers_", "formats_", "harness_")))


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; building is a no-op when libgsr_hip.so is newer than its sources."""
    from gaustar_amd import _lib, build
    if os.path.isdir(os.path.join(ROOT, "gaustar_amd", "csrc")) and os.path.exists("/opt/rocm/bin/hipcc"):
        build.build()
    return _lib.load()
