#!/usr/bin/env python3
"""Spill-format goldens: the bytes the REFERENCE writes for a set of k-mers —
one run file of emem::external_memory_vector<uint64_t> (sorted, raw little-endian elements through io::basic_store)
and one io::basic_store(std::vector<uint64_t>) file (size_t length prefix + elements).  Build container only."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

R = O.ref()
R.ref_emv_create.restype = C.c_void_p
R.ref_emv_create.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p, C.c_void_p]
R.ref_emv_read.restype = C.c_uint64
R.ref_emv_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
R.ref_emv_destroy.argtypes = [C.c_void_p]
R.ref_store_vector_u64.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
OUT = os.path.join(HERE, "spill")
os.makedirs(OUT, exist_ok=True)

seq = O.synth(99, 3000)
val, ok = O.units(seq, np.array([0, 3000], np.uint64), 21, True)
keys = val[ok.astype(bool)]  # unsorted, with duplicates possible
np.save(os.path.join(OUT, "keys.npy"), keys)
with tempfile.TemporaryDirectory() as d:
    runs = C.c_uint64()
    h = R.ref_emv_create(O._ptr(keys), len(keys), 1 << 30, d.encode(), b"first", C.byref(runs))
    assert runs.value == 1
    data = open(os.path.join(d, "tmp.run_first_0.bin"), "rb").read()
    open(os.path.join(OUT, "tmp.run_first_0.bin"), "wb").write(data)
    back = np.zeros(len(keys), np.uint64)
    assert R.ref_emv_read(h, O._ptr(back), len(back)) == len(keys) and np.array_equal(back, np.sort(keys))
    R.ref_emv_destroy(h)
    R.ref_store_vector_u64(os.path.join(OUT, "vector.bin").encode(), O._ptr(np.sort(keys)), len(keys))
print("keys", len(keys), "run bytes", len(data), "vector bytes", os.path.getsize(os.path.join(OUT, "vector.bin")))
