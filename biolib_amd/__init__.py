"""biolib_amd — MI355X-native k-mer / minimizer streaming scan behind biolib's view interface.

The product is the HIP library (biolib_amd/csrc -> biolib_amd/lib/libbiolib_amd.so) and its C ABI
(include/biolib_amd.h); the C++ drop-in headers live in include/compat/.  This Python package is
the thin host-side binding used by the tests and bench.py (ctypes + torch for device memory).
"""
from .capi import BiolibError, FLAG_CANONICAL, FLAG_DROP_LAST, FLAG_SYNC, LIB_PATH, Result, lib  # noqa: F401
from .scan import Batch, Context, Reader, hash64  # noqa: F401

__all__ = ["Batch", "Context", "Reader", "BiolibError", "hash64", "lib", "LIB_PATH", "FLAG_CANONICAL", "FLAG_DROP_LAST", "FLAG_SYNC"]
