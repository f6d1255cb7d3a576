"""ctypes access to the CPU oracle (oracle/_build/libbl_oracle.so) and, when it has been
built in the container that holds /root/reference, to the reference itself
(oracle/_ref/libbiolib_ref.so).  TEST INFRASTRUCTURE: imported only by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg — never by biolib_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libbl_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libbiolib_ref.so")

u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
vp = C.c_void_p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.blo_splitmix64.restype = C.c_uint64
        L.blo_splitmix64.argtypes = [C.c_uint64]
        L.blo_synth.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp]
        L.blo_murmur3_x64_128.argtypes = [vp, C.c_int, C.c_uint32, vp]
        L.blo_hash64_u64.restype = C.c_uint64
        L.blo_hash64_u64.argtypes = [C.c_uint64, C.c_uint64]
        L.blo_hash64_u64_general.restype = C.c_uint64
        L.blo_hash64_u64_general.argtypes = [C.c_uint64, C.c_uint64]
        L.blo_hash64_bytes.restype = C.c_uint64
        L.blo_hash64_bytes.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.blo_remix.restype = C.c_uint64
        L.blo_remix.argtypes = [C.c_uint64]
        L.blo_nt4.restype = C.c_uint8
        L.blo_nt4.argtypes = [C.c_uint8]
        L.blo_kmer_items.restype = C.c_size_t
        L.blo_kmer_items.argtypes = [vp, C.c_size_t, C.c_uint, C.c_int, C.c_int, vp, vp, vp, vp, C.c_size_t]
        L.blo_units.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.c_int, vp, vp]
        L.blo_kmer_digest.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.c_int, C.c_uint64, C.c_int, C.c_int, vp]
        L.blo_minimizers.restype = C.c_size_t
        L.blo_minimizers.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, C.c_size_t]
        L.blo_minimizer_digest.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint64, C.c_int, C.c_int, vp]
        L.blo_super_kmers.restype = C.c_size_t
        L.blo_super_kmers.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint64, C.c_int, vp, vp, vp, vp, vp, C.c_size_t]
        L.blo_minimizer_position.restype = C.c_uint
        L.blo_minimizer_position.argtypes = [C.c_uint64, C.c_uint, C.c_uint]
        L.blo_syncmers.restype = C.c_size_t
        L.blo_syncmers.argtypes = [vp, vp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The unmodified reference behind oracle/ref_shim.cpp (None if not built)."""
    global _ref
    if _ref is None and have_ref():
        L = C.CDLL(REF_SO)
        L.ref_hash64_u64.restype = C.c_uint64
        L.ref_hash64_u64.argtypes = [C.c_uint64, C.c_uint64]
        L.ref_double_hash64_u64.argtypes = [C.c_uint64, C.c_uint64, vp]
        L.ref_hash64_bytes.restype = C.c_uint64
        L.ref_hash64_bytes.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.ref_hash64_u128.restype = C.c_uint64
        L.ref_hash64_u128.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.ref_hash64_u32.restype = C.c_uint64
        L.ref_hash64_u32.argtypes = [C.c_uint32, C.c_uint64]
        L.ref_remix.restype = C.c_uint64
        L.ref_remix.argtypes = [C.c_uint64]
        L.ref_kmer_view.restype = C.c_size_t
        L.ref_kmer_view.argtypes = [vp, C.c_size_t, C.c_uint8, C.c_int, C.c_int, vp, vp, vp, vp, C.c_size_t]
        L.ref_minpos.restype = C.c_size_t
        L.ref_minpos.argtypes = [vp, C.c_size_t, C.c_uint8, C.c_uint8, C.c_int, C.c_int, vp, C.c_size_t]
        L.ref_syncmer_count.restype = C.c_uint64
        L.ref_syncmer_count.argtypes = [vp, C.c_size_t, C.c_uint8, C.c_uint8, C.c_uint16, C.c_uint16, C.c_int]
        L.ref_scan_kmers_xor.restype = C.c_uint64
        L.ref_scan_kmers_xor.argtypes = [vp, C.c_size_t, C.c_uint8, C.c_int]
        L.ref_scan_kmer_hash_xor.restype = C.c_uint64
        L.ref_scan_kmer_hash_xor.argtypes = [vp, C.c_size_t, C.c_uint8, C.c_int, C.c_uint64]
        _ref = L
    return _ref


# ----------------------------------------------------------------------------- helpers

def as_bytes(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8).copy()
    return np.ascontiguousarray(seq, dtype=np.uint8)


def synth(seed, n, first=0):
    out = np.empty(n, dtype=np.uint8)
    oracle().blo_synth(seed, first, n, _ptr(out))
    return out


def fixed_offsets(n, read_len):
    """consecutive fixed-length slices; a trailing partial read is kept as a short read"""
    offs = list(range(0, n, read_len)) + [n]
    return np.asarray(offs, dtype=np.uint64)


def kmer_items(seq, k, canonical, complete, lib=None):
    """[(position, id, value-or-None)] through the oracle (or, with lib=ref(), the reference)."""
    s = as_bytes(seq)
    cap = len(s) + 2
    vals = np.zeros(cap, np.uint64)
    nul = np.zeros(cap, np.uint8)
    pos = np.zeros(cap, np.uint64)
    ids = np.zeros(cap, np.uint64)
    if lib is None:
        n = oracle().blo_kmer_items(_ptr(s), len(s), k, int(canonical), int(complete), _ptr(vals), _ptr(nul), _ptr(pos), _ptr(ids), cap)
    else:
        n = lib.ref_kmer_view(_ptr(s), len(s), k, int(canonical), int(complete), _ptr(vals), _ptr(nul), _ptr(pos), _ptr(ids), cap)
    assert n <= cap
    return [(int(pos[i]), int(ids[i]), None if nul[i] else int(vals[i])) for i in range(n)]


def units(seq, offsets, k, canonical):
    s = as_bytes(seq)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    vals = np.zeros(len(s), np.uint64)
    valid = np.zeros(len(s), np.uint8)
    oracle().blo_units(_ptr(s), _ptr(offsets), len(offsets) - 1, k, int(canonical), _ptr(vals), _ptr(valid))
    return vals, valid


def kmer_digest(seq, offsets, k, canonical, seed, drop_last=False, threads=1):
    s = as_bytes(seq)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    d = np.zeros(4, np.uint64)
    oracle().blo_kmer_digest(_ptr(s), _ptr(offsets), len(offsets) - 1, k, int(canonical), seed, int(drop_last), threads, _ptr(d))
    return dict(count=int(d[0]), xor_value=int(d[1]), xor_hash=int(d[2]), sum_hash=int(d[3]))


def minimizers(seq, offsets, unit, w, seed, canonical, brute=True):
    s = as_bytes(seq)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    cap = len(s) + 1
    v = np.zeros(cap, np.uint64)
    p = np.zeros(cap, np.uint64)
    h = np.zeros(cap, np.uint64)
    n = oracle().blo_minimizers(_ptr(s), _ptr(offsets), len(offsets) - 1, unit, w, seed, int(canonical), int(brute), _ptr(v), _ptr(p), _ptr(h), cap)
    return v[:n].copy(), p[:n].copy(), h[:n].copy()


def minimizer_digest(seq, offsets, unit, w, seed, canonical, threads=1):
    s = as_bytes(seq)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    d = np.zeros(4, np.uint64)
    oracle().blo_minimizer_digest(_ptr(s), _ptr(offsets), len(offsets) - 1, unit, w, seed, int(canonical), threads, _ptr(d))
    return dict(count=int(d[0]), xor_value=int(d[1]), xor_hash=int(d[2]), xor_pos=int(d[3]))


def super_kmers(seq, offsets, k, m, seed, canonical):
    s = as_bytes(seq)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    cap = len(s) + 1
    mn = np.zeros(cap, np.uint64)
    fp = np.zeros(cap, np.uint64)
    mp = np.zeros(cap, np.uint8)
    sz = np.zeros(cap, np.uint8)
    hs = np.zeros(cap, np.uint64)
    n = oracle().blo_super_kmers(_ptr(s), _ptr(offsets), len(offsets) - 1, k, m, seed, int(canonical), _ptr(mn), _ptr(fp), _ptr(mp), _ptr(sz), _ptr(hs), cap)
    return mn[:n].copy(), fp[:n].copy(), mp[:n].copy(), sz[:n].copy(), hs[:n].copy()


def syncmers(seq, offsets, k, m, soff, eoff, canonical, drop_last=False, threads=1, positions=True):
    s = as_bytes(seq)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    cap = len(s) + 1 if positions else 0
    p = np.zeros(max(cap, 1), np.uint64)
    n = oracle().blo_syncmers(_ptr(s), _ptr(offsets), len(offsets) - 1, k, m, soff, eoff, int(canonical), int(drop_last), threads, _ptr(p) if positions else None, cap)
    return (n, p[:n].copy()) if positions else (n, None)


def xor_reduce(a):
    return int(np.bitwise_xor.reduce(a)) if len(a) else 0


def hash64_np(values, seed):
    """vectorised hash64 of uint64 keys (numpy restatement of blo_hash64_u64, pinned against it in
    tests/test_shard_gloo.py): low half of MurmurHash3_x64_128 over the 8 key bytes, 32-bit seed."""
    M = np.uint64
    with np.errstate(over="ignore"):
        k = np.asarray(values, dtype=np.uint64).copy()
        s = M(int(seed) & 0xFFFFFFFF)
        rotl = lambda x, r: (x << M(r)) | (x >> M(64 - r))
        k *= M(0x87C37B91114253D5)
        k = rotl(k, 31)
        k *= M(0x4CF5AD432745937F)
        h1 = (s ^ k) ^ M(8)
        h2 = np.full_like(k, s ^ M(8))
        h1 = h1 + h2
        h2 = h2 + h1

        def fmix(x):
            x = x ^ (x >> M(33))
            x = x * M(0xFF51AFD7ED558CCD)
            x = x ^ (x >> M(33))
            x = x * M(0xC4CEB9FE1A85EC53)
            return x ^ (x >> M(33))

        h1, h2 = fmix(h1), fmix(h2)
        return h1 + h2
