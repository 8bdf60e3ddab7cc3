"""vlg_matching_amd -- MI355X-native variable-length-gap matcher (host-side mirror of the reference interface).

The compute path is the HIP library `libvlg_hip.so` (hand-written gfx950 kernels behind the C-ABI of
include/vlg_hip.h).  There is no CPU fallback: importing works anywhere, but every compute call raises
VlgError when the library or a HIP device is missing.
"""
from .capi import VlgError, lib, library_path, build_library  # noqa: F401
from .index import VlgIndex, WtsaIndex, BitVector, RrrBitVector, SearchResult, SymbolMap, count, locate, parse_query  # noqa: F401
