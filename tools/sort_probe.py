#!/usr/bin/env python3
"""One long list of uniform 32-bit positions through vlg_sort_lists_u32 (K4 stand-alone), three times: run under
rocprofv3 --kernel-trace --stats to read the sort kernels' durations.  usage: tools/sort_probe.py [log2 keys per list] [position bits] [lists]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vlg_matching_amd as V

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = nl << lg
g = torch.Generator(device="cuda").manual_seed(1)
keys = torch.randint(0, 1 << bits, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
off = (np.arange(nl + 1, dtype=np.uint64) << np.uint64(lg))
L = V.lib()
for mode in (1, 1, 1, 2):
    d = keys.clone()
    nc = C.c_uint64(0)
    V.capi.check(L.vlg_sort_lists_u32(d.data_ptr(), off.ctypes.data, nl, bits, mode, C.byref(nc), None))
    if os.environ.get("VLG_SORT_PROBE_CHECK", "1") == "1":
        v = (d.to(torch.int64) & 0xFFFFFFFF).view(nl, -1)
        assert bool((v[:, 1:] >= v[:, :-1]).all()), mode
print("ok", n, bits)
