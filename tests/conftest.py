import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# every sort of occurrence lists is verified ascending on the device (search.hip: lists_check_kernel after the sort)
os.environ.setdefault("VLG_CHECK_SORT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def refmod(oracle):
    """oracle/_ref (the reference's own code). Skips when it was never built."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libvlgref.so not built (needs /root/reference)")
    return oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_initialises_the_gpu_first():
    """torch.cuda.is_available() must be asked before any other library in the process has brought the HIP runtime up
    (asked afterwards it can report False on this image), so do it once at session start."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
