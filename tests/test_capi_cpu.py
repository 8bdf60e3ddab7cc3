"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/vlg_hip.h declares, and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def V():
    import vlg_matching_amd as v
    if not os.path.exists(v.library_path()):
        v.build_library()
    return v


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vlg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vlg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(V):
    L = C.CDLL(V.library_path())
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "library does not export " + n
    bound = {s[0] for s in V.capi.SYMBOLS}
    assert set(names) == bound, set(names) ^ bound


def test_no_gpu_means_loud_failure(V):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(V.VlgError) as e:
        V.VlgIndex.build(b"abracadabra")
    assert e.value.status == V.capi.E_NO_DEVICE
    with pytest.raises(V.VlgError):
        V.BitVector(np.zeros(1, np.uint64), 10)
    with pytest.raises(V.VlgError):
        V.index.Workspace()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under vlg_matching_amd/ or include/ may reference it."""
    bad = []
    for base in ("vlg_matching_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".c", "Makefile")):
                    s = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle|vlgo_|libvlgref|/root/reference", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
