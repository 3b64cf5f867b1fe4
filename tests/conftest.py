"""pytest wiring: `gpu` marker, import paths (the package dir `rio-rs_amd/` has a hyphen)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


_MARKEXPR = ""


def pytest_configure(config):
    global _MARKEXPR
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    _MARKEXPR = config.getoption("-m") or ""


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """GPU runs only: the test process holds two HIP runtimes — the one bundled with torch (the sharded tests' streams and
    process groups) and /opt/rocm's under librio_gp.so.  torch's fails to find the device when it comes up second, so it is
    brought up before any test creates a handle, whatever subset of the tests was selected."""
    if "gpu" in _MARKEXPR and "not gpu" not in _MARKEXPR:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    yield


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle
