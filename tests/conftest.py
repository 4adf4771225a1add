import os
import sys

import pytest

# Several ranks of the peer-to-peer transport inside ONE process (tests/test_gpu_peer.py, contexts on threads): a rank's wait kernel
# spins on the device until its peers' signal kernels have run, so no two ranks' streams may share a hardware queue (HIP multiplexes a
# process's streams onto GPU_MAX_HW_QUEUES = 4 queues by default; a signal queued behind a spinning wait would never run).  Read by
# the HIP runtime at initialisation, hence before torch.  One process per GPU -- the production layout -- needs nothing of the kind.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# the library's development switches (NMFX_POTRS, NMFX_K_GRANULE, NMFX_RS_FUSED, ...: alternative kernels the tests run against each
# other) are dead unless NMFX_DEV=1 is set as well (csrc/comm.hpp: dev_env)
os.environ.setdefault("NMFX_DEV", "1")

import torch  # noqa: E402,F401  first: pins ONE HIP runtime for the process (see nmfx/_lib.py)

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
