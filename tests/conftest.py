import ctypes
import os
import subprocess
import sys

import pytest

# The oracle is OpenMP code: pin its team size before libgomp starts, otherwise a box
# that reports more processors than its CPU quota oversubscribes and crawls.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, len(os.sched_getaffinity(0))))))
os.environ.setdefault("GOMP_SPINCOUNT", "100000")  # idle OpenMP threads spin ~0.1 ms, then sleep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def product():
    """The product library (host-side helpers work without a GPU)."""
    import fargocpt_amd
    return fargocpt_amd.load()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle: test infrastructure only (oracle/fargo_oracle.c)."""
    from fargocpt_amd.binding import Library
    so = os.path.join(ROOT, "oracle", "libfargo_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return Library(ctypes.CDLL(so), "orc_")
