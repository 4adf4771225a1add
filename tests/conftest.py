import os
import sys

import pytest
import torch  # noqa: F401  first: pins ONE HIP runtime for the process (see nmfx/_lib.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("nmf.jl_amd", "oracle", "tests"):
    path = os.path.join(ROOT, sub)
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libnmfx.so and the C oracle exist (cross-compiles on CPU)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    return True
