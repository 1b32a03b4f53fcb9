import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # Tests that hand device pointers over use torch for device memory.  torch must initialise its HIP runtime BEFORE
    # libdirect_ddp.so is loaded, or it reports no GPU (bench.py imports it first for the same reason): do it once
    # here, whatever order the test files run in.  Only when a GPU run was asked for (the import costs seconds).
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch
            torch.cuda.is_available()
        except Exception:
            pass


@pytest.fixture(scope="session")
def built():
    """Native pieces are built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True
