import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the native pieces once per session if they are missing (cross-compiles without a GPU)."""
    import subprocess
    need = [os.path.join(ROOT, "gvpm_amd", "libgvpm_hip.so"), os.path.join(ROOT, "gvpm_amd", "host", "libgvpm_host.so"),
            os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "oracle", "liboracle_fast.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gvpm_amd", "csrc")])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    # torch ships its own HIP runtime: it must initialise before libgvpm_hip.so touches the GPU,
    # otherwise a later torch.cuda init in the same process finds no device
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield
