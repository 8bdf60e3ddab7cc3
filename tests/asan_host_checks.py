#!/usr/bin/env python3
"""Host code of the library under AddressSanitizer, no GPU needed (test infrastructure, run by hand):
    make -C vlg_matching_amd/csrc asan
    LD_PRELOAD=$(find /opt/rocm/lib/llvm/lib/clang -name libclang_rt.asan-x86_64.so | head -1) \\
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 python3 tests/asan_host_checks.py
Random strings through both dialects of the query parser; a reference-written csa_wt file and hundreds of truncated /
bit-flipped copies through the sdsl file reader.  (The image comes from the test-only reference glue.)"""
import sys, os, numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE)); sys.path.insert(0, _HERE)
import vlg_matching_amd.capi as capi
capi.library_path = lambda: os.environ.get("VLG_ASAN_LIB", "/tmp/vlg_asan/libvlg_hip.so")
import vlg_matching_amd as V
from oracle import oracle as O
from util import bwt_from_sa, dna_text, skewed_text
# parser: valid, invalid, weird
import random
rnd = random.Random(1)
alphabet = "abc.{}?,0123456789 -+x"
n_ok = n_bad = 0
for _ in range(20000):
    q = "".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 30)))
    for d in (0, 1):
        try:
            V.parse_query(q, d); n_ok += 1
        except V.VlgError:
            n_bad += 1
print("parser", n_ok, n_bad)
# reader on good and damaged files
text = dna_text(3000, 8).tobytes()
tz = np.frombuffer(text + b"\0", dtype=np.uint8)
sa = O.suffix_array(tz)
O.RefIndex(bwt_from_sa(tz, sa), sa, 0).write_csa_image("/tmp/vlg_asan/good.sdsl", sa)
raw = bytearray(open("/tmp/vlg_asan/good.sdsl", "rb").read())
rng = np.random.default_rng(5)
cases = [bytes(raw)] + [bytes(raw[:n]) for n in list(range(0, 64)) + [int(x) for x in rng.integers(64, len(raw), 200)]]
for _ in range(600):
    b = bytearray(raw)
    for pos in rng.integers(0, len(b), int(rng.integers(1, 4))):
        b[pos] = int(rng.integers(0, 256))
    cases.append(bytes(b))
ok = bad = 0
for c in cases:
    open("/tmp/vlg_asan/bad.sdsl", "wb").write(c)
    try:
        V.index.read_sdsl_file("/tmp/vlg_asan/bad.sdsl"); ok += 1
    except V.VlgError:
        bad += 1
print("reader", ok, bad)
