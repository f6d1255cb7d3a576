"""CPU-only: the HIP kernels' phase functions, compiled for the host (tests/emu/), against the oracle.
 - emu_selftest: -fsanitize=address,undefined build, randomised batches (breaks, ragged/empty/short
   sequences, fixed reads, sub-range unions, tie-heavy DNA, tile-boundary sizes), all four scans.
 - libbl_emu.so: the golden fixtures the reference produced, through the emulated kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    L = C.CDLL(os.path.join(EMU_DIR, "_build", "libbl_emu.so"))
    vp, u64, u = C.c_void_p, C.c_uint64, C.c_uint
    L.emu_batch.restype = vp
    L.emu_batch.argtypes = [vp, u64, vp, u64, u64]
    L.emu_batch_free.argtypes = [vp]
    L.emu_minimizers.argtypes = [vp, u64, u64, u, u, u64, u, vp, vp, vp, u64, vp]
    L.emu_super_kmers.argtypes = [vp, u64, u64, u, u, u64, u, vp, vp, vp, vp, vp, u64, vp]
    L.emu_syncmers.argtypes = [vp, u64, u64, u, u, u, u, u64, u, vp, u64, vp]
    L.emu_kmers.argtypes = [vp, u64, u64, u, u64, u, vp, vp, vp, vp]
    return L


def test_sanitizer_selftest(emu):
    out = subprocess.run([os.path.join(EMU_DIR, "_build", "emu_selftest"), "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "emu_selftest: OK" in out.stdout


def test_top_dword_used_by_pass_one_is_the_hash_or_one_below(emu):
    """murmur64_top (bl_scan_core.hpp) leaves out the carry of the two low dwords: T - S is 0 or 1, for every key and seed; the
    window code of the approximate pass 1 relies on exactly that (ties: keys less than two prefixes apart)"""
    rng = np.random.default_rng(5)
    emu.emu_top_check.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_void_p, C.c_void_p]
    for seed in (0, 42, 0xffffffff, 7):
        keys = rng.integers(0, 2**64, 400_000, dtype=np.uint64)
        keys[:4] = (0, 1, 2**64 - 1, 2**62 - 1)
        bad, below = C.c_uint64(), C.c_uint64()
        emu.emu_top_check(O._ptr(keys), len(keys), seed, C.byref(bad), C.byref(below))
        assert bad.value == 0
        assert 0.4 < below.value / len(keys) < 0.6  # the carry is a coin toss


def _batch(emu, seq, offs=None, read_len=0):
    seq = O.as_bytes(seq)
    if offs is not None:
        offs = np.ascontiguousarray(offs, np.uint64)
        return emu.emu_batch(O._ptr(seq), len(seq), O._ptr(offs), len(offs) - 1, 0)
    return emu.emu_batch(O._ptr(seq), len(seq), None, 0, read_len)


def test_golden_minimizers_and_super_kmers(emu, golden_arrays):
    A = golden_arrays
    for name in ("clean_one", "broken_one", "clean_reads150", "broken_reads150", "clean_ragged", "broken_ragged"):
        seq = A["small_clean" if name.startswith("clean") else "small_broken"]
        kind = name.split("_", 1)[1]
        b = _batch(emu, seq, A["ragged_offsets"]) if kind == "ragged" else _batch(emu, seq, None, 150 if kind == "reads150" else 0)
        cap = len(seq) + 1
        res = np.zeros(8, np.uint64)
        for (unit, w, seed, canon) in ((31, 11, 42, 1), (15, 17, 42, 1), (11, 21, 0, 0), (5, 4, 1, 1), (32, 2, 9, 1), (8, 1, 3, 0)):
            exp = A[f"mm_{name}_u{unit}_w{w}_s{seed}_c{canon}"]
            v, p, h = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
            emu.emu_minimizers(b, 0, 0, unit, w, seed, canon, O._ptr(v), O._ptr(p), O._ptr(h), cap, O._ptr(res))
            n = int(res[0])
            assert n == len(exp), (name, unit, w)
            assert np.array_equal(v[:n], exp[:, 0]) and np.array_equal(p[:n], exp[:, 1]) and np.array_equal(h[:n], exp[:, 2])
        for (k, m, seed, canon) in ((31, 15, 42, 1), (21, 8, 0, 0), (31, 31, 5, 1)):
            exp = A[f"sk_{name}_k{k}_m{m}_s{seed}_c{canon}"]
            mn, fp, hs = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
            mp, sz = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
            emu.emu_super_kmers(b, 0, 0, k, m, seed, canon, O._ptr(mn), O._ptr(fp), O._ptr(mp), O._ptr(sz), O._ptr(hs), cap, O._ptr(res))
            n = int(res[0])
            assert n == len(exp) and int(res[4]) == n
            assert np.array_equal(mn[:n], exp[:, 0]) and np.array_equal(fp[:n], exp[:, 1]) and np.array_equal(hs[:n], exp[:, 4])
            assert np.array_equal(mp[:n].astype(np.uint64), exp[:, 2]) and np.array_equal(sz[:n].astype(np.uint64), exp[:, 3])
        emu.emu_batch_free(b)


def test_golden_units_and_one_mib_digests(emu, golden_arrays, golden_kats):
    A = golden_arrays
    res = np.zeros(8, np.uint64)
    for name in ("clean", "broken"):
        seq = A[f"small_{name}"]
        b = _batch(emu, seq)
        for k in (5, 15, 21, 31, 32):
            for canon in (0, 1):
                v, h, ok = np.zeros(len(seq), np.uint64), np.zeros(len(seq), np.uint64), np.zeros(len(seq), np.uint8)
                emu.emu_kmers(b, 0, 0, k, 0, canon, O._ptr(v), O._ptr(h), O._ptr(ok), O._ptr(res))
                assert np.array_equal(ok, A[f"units_{name}_k{k}_c{canon}_ok"])
                assert np.array_equal(v, A[f"units_{name}_k{k}_c{canon}_val"])
        emu.emu_batch_free(b)
    D = golden_kats["digests_1MiB_seed42"]
    n = 1 << 20
    s = O.synth(42, n)
    b = _batch(emu, s)
    emu.emu_kmers(b, 0, 0, 31, 0, 1 | 2, None, None, None, O._ptr(res))
    d = D["k31_canon1_idiom_seed0"]
    assert [int(x) for x in res[:4]] == [d["count"], d["xor_value"], d["xor_hash"], d["sum_hash"]]
    emu.emu_kmers(b, 0, 0, 21, 0, 2, None, None, None, O._ptr(res))
    assert (int(res[0]), int(res[1])) == (D["k21_canon0_idiom"]["count"], D["k21_canon0_idiom"]["xor_value"])
    emu.emu_syncmers(b, 0, 0, 31, 11, 0, 20, 0, 1 | 2, None, 0, O._ptr(res))
    assert int(res[0]) == D["syncmer_k31_s11_0_20_canon1_idiom"]
    emu.emu_syncmers(b, 0, 0, 31, 11, 0, 20, 0, 2, None, 0, O._ptr(res))
    assert int(res[0]) == D["syncmer_k31_s11_0_20_canon0_idiom"]
    emu.emu_batch_free(b)
    c3 = D["C3_like_reads150_unit31_w11_seed42"]
    b = _batch(emu, s[:c3["n_bases"]], None, 150)
    emu.emu_minimizers(b, 0, 0, 31, 11, 42, 1, None, None, None, 0, O._ptr(res))
    assert [int(x) for x in res[:4]] == [c3["count"], c3["xor_value"], c3["xor_hash"], c3["xor_pos"]]
    emu.emu_batch_free(b)
    c4 = D["C4_like_reads10k_k31_m15_seed42"]
    b = _batch(emu, s[:c4["n_bases"]], None, 10000)
    emu.emu_super_kmers(b, 0, 0, 31, 15, 42, 1, None, None, None, None, None, 0, O._ptr(res))
    assert (int(res[0]), int(res[1]), int(res[2]), int(res[4])) == (c4["count"], c4["xor_minimizer"], c4["xor_hash"], c4["count"])
    emu.emu_batch_free(b)
