"""The CPU oracle (oracle/bl_oracle.c) against the golden vectors that the unmodified
reference produced (tests/golden/, generator make_golden.py) and, when the reference
library was built in this container, live against the reference itself.
CPU-only: runs under -m "not gpu"."""
import numpy as np
import pytest

import oracle_lib as O

GOLD_ONE = ("clean_one", "broken_one")
SEQ_OF = {"clean": "small_clean", "broken": "small_broken"}


def _seq_offs(A, name):
    seq = A[SEQ_OF[name.split("_")[0]]]
    kind = name.split("_", 1)[1]
    if kind == "one":
        offs = np.array([0, len(seq)], np.uint64)
    elif kind == "reads150":
        offs = np.arange(0, len(seq) + 1, 150, dtype=np.uint64)
    else:
        offs = A["ragged_offsets"]
    return seq, offs


def test_synth_generator(golden_kats):
    s = O.synth(42, 64)
    assert bytes(s[:32]).decode() == golden_kats["synth_seed42_first32"]
    # windows of the stream agree with the stream
    assert np.array_equal(O.synth(42, 1000, first=37), O.synth(42, 1037)[37:])


def test_nt4_table():
    L = O.oracle()
    exp = {ord(c): v for c, v in zip("ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3])}
    for c in range(256):
        assert L.blo_nt4(c) == exp.get(c, 4)


def test_hash_kats(golden_kats):
    L = O.oracle()
    for v, s, h in golden_kats["hash64_u64"]:
        assert L.blo_hash64_u64(v, s) == h
    out = np.zeros(2, np.uint64)
    for v, s, h0, h1 in golden_kats["double_hash64_u64"]:
        key = np.array([v], np.uint64)
        L.blo_murmur3_x64_128(O._ptr(key), 8, s & 0xFFFFFFFF, O._ptr(out))
        assert (int(out[0]), int(out[1])) == (h0, h1)
    for hexkey, s, h in golden_kats["hash64_bytes"]:
        key = np.frombuffer(bytes.fromhex(hexkey), np.uint8).copy()
        assert L.blo_hash64_bytes(O._ptr(key) if len(key) else None, len(key), s) == h
    lo, hi, s, h = golden_kats["hash64_u128"][0]
    key = np.array([lo, hi], np.uint64)
    assert L.blo_hash64_bytes(O._ptr(key), 16, s) == h
    v, s, h = golden_kats["hash64_u32"][0]
    key = np.array([v], np.uint32)
    assert L.blo_hash64_bytes(O._ptr(key), 4, s) == h
    for z, r in golden_kats["remix"]:
        assert L.blo_remix(z) == r
    # SURVEY.md §8a-a4 spot values
    assert L.blo_hash64_u64(0, 0) == 0x28DF63B7CC57C3CB
    assert L.blo_hash64_u64(0x0123456789ABCDEF, 0x10000002A) == L.blo_hash64_u64(0x0123456789ABCDEF, 0x2A)


def test_kmer_item_protocol(golden_kats):
    for sect in ("kmer_items_tiny", "kmer_items_k21", "kmer_items_k32"):
        for e in golden_kats[sect]:
            got = O.kmer_items(e["seq"], e["k"], e["canonical"], e["complete"])
            exp = [tuple(x) for x in e["items"]]
            assert got == exp, (e["seq"], e["k"], e["canonical"], e["complete"])


def test_kmer_items_outside_reference_domain_terminate():
    # Q2: the reference reads out of bounds here; the contract is "terminate cleanly"
    assert O.kmer_items("AC", 3, 0, 1) == []
    assert O.kmer_items("", 3, 0, 1) == []
    assert O.kmer_items("ACGTNN", 3, 0, 0) == [(0, 0, 6), (1, 1, 27), (2, 2, None)]
    assert O.kmer_items("ACNGT", 3, 0, 1) == []
    assert O.kmer_items("ACGTNAC", 3, 0, 1) == [(0, 0, 6), (1, 1, 27), (2, 2, None)]


@pytest.mark.parametrize("name", ["clean", "broken"])
def test_units_vs_reference_arrays(golden_arrays, name):
    A = golden_arrays
    seq = A[SEQ_OF[name]]
    offs = np.array([0, len(seq)], np.uint64)
    for k in (5, 15, 21, 31, 32):
        for canon in (0, 1):
            val, ok = O.units(seq, offs, k, canon)
            assert np.array_equal(ok, A[f"units_{name}_k{k}_c{canon}_ok"])
            assert np.array_equal(val, A[f"units_{name}_k{k}_c{canon}_val"])


@pytest.mark.parametrize("name", ["clean", "broken"])
def test_minimizer_position_extractor(golden_arrays, golden_kats, name):
    A = golden_arrays
    seq = A[SEQ_OF[name]]
    L = O.oracle()
    for (k, m) in ((31, 11), (21, 8), (7, 4), (15, 15)):
        for canon in (0, 1):
            items = O.kmer_items(seq, k, canon, True)
            got = [k + 1 if v is None else L.blo_minimizer_position(v, k, m) for _, _, v in items]
            assert np.array_equal(np.array(got, np.uint16), A[f"minpos_{name}_k{k}_m{m}_c{canon}"])
    e = golden_kats["minpos_example"]
    items = O.kmer_items(e["seq"], e["k"], e["canonical"], e["complete"])
    assert [L.blo_minimizer_position(v, e["k"], e["m"]) for _, _, v in items] == e["minpos"]
    n, pos = O.syncmers(e["seq"], np.array([0, len(e["seq"])], np.uint64), 7, 4, 0, 3, True, drop_last=True)
    assert n == 8


@pytest.mark.parametrize("name", ["clean_one", "broken_one", "clean_reads150", "broken_reads150", "clean_ragged", "broken_ragged"])
def test_minimizers_and_super_kmers_vs_composition(golden_arrays, name):
    A = golden_arrays
    seq, offs = _seq_offs(A, name)
    for (unit, w, seed, canon) in ((31, 11, 42, 1), (15, 17, 42, 1), (11, 21, 0, 0), (5, 4, 1, 1), (32, 2, 9, 1), (8, 1, 3, 0)):
        exp = A[f"mm_{name}_u{unit}_w{w}_s{seed}_c{canon}"]
        for brute in (True, False):
            v, p, h = O.minimizers(seq, offs, unit, w, seed, canon, brute=brute)
            assert len(v) == len(exp)
            assert np.array_equal(v, exp[:, 0]) and np.array_equal(p, exp[:, 1]) and np.array_equal(h, exp[:, 2])
        d = O.minimizer_digest(seq, offs, unit, w, seed, canon, threads=2)
        assert d["count"] == len(exp) and d["xor_hash"] == O.xor_reduce(exp[:, 2]) and d["xor_pos"] == O.xor_reduce(exp[:, 1])
    for (k, m, seed, canon) in ((31, 15, 42, 1), (21, 8, 0, 0), (31, 31, 5, 1)):
        exp = A[f"sk_{name}_k{k}_m{m}_s{seed}_c{canon}"]
        mn, fp, mp, sz, hs = O.super_kmers(seq, offs, k, m, seed, canon)
        assert len(mn) == len(exp)
        assert np.array_equal(mn, exp[:, 0]) and np.array_equal(fp, exp[:, 1])
        assert np.array_equal(mp.astype(np.uint64), exp[:, 2]) and np.array_equal(sz.astype(np.uint64), exp[:, 3])
        assert np.array_equal(hs, exp[:, 4])


def test_one_mib_digests(golden_kats):
    D = golden_kats["digests_1MiB_seed42"]
    n = 1 << 20
    s = O.synth(42, n)
    one = np.array([0, n], np.uint64)
    for canon in (0, 1):
        it = O.kmer_items(s, 21, canon, False)
        assert len(it) == D[f"k21_canon{canon}_idiom"]["count"]
        assert O.xor_reduce(np.array([v for _, _, v in it], np.uint64)) == D[f"k21_canon{canon}_idiom"]["xor_value"]
    assert O.kmer_digest(s, one, 31, True, 0, drop_last=False, threads=4) == D["k31_canon1_complete_seed0"]
    assert O.kmer_digest(s, one, 31, True, 0, drop_last=True, threads=4) == D["k31_canon1_idiom_seed0"]
    assert O.syncmers(s, one, 31, 11, 0, 20, True, drop_last=True, threads=8, positions=False)[0] == D["syncmer_k31_s11_0_20_canon1_idiom"]
    assert O.syncmers(s, one, 31, 11, 0, 20, False, drop_last=True, threads=8, positions=False)[0] == D["syncmer_k31_s11_0_20_canon0_idiom"]
    assert O.syncmers(s, one, 21, 8, 0, 13, True, drop_last=True, threads=8, positions=False)[0] == D["syncmer_k21_s8_0_13_canon1_idiom"]
    c3 = D["C3_like_reads150_unit31_w11_seed42"]
    d = O.minimizer_digest(s[:c3["n_bases"]], O.fixed_offsets(c3["n_bases"], 150), 31, 11, 42, True, threads=4)
    assert (d["count"], d["xor_value"], d["xor_hash"], d["xor_pos"]) == (c3["count"], c3["xor_value"], c3["xor_hash"], c3["xor_pos"])
    c4 = D["C4_like_reads10k_k31_m15_seed42"]
    mn, fp, mp, sz, hs = O.super_kmers(s[:c4["n_bases"]], O.fixed_offsets(c4["n_bases"], 10000), 31, 15, 42, True)
    assert (len(mn), O.xor_reduce(mn), O.xor_reduce(hs), int(sz.sum()), O.xor_reduce(fp)) == (
        c4["count"], c4["xor_minimizer"], c4["xor_hash"], c4["sum_size"], c4["xor_first_pos"])


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference; build container only)")
def test_live_against_reference_library():
    R = O.ref()
    L = O.oracle()
    rng = np.random.default_rng(1)
    for v in rng.integers(0, 2**63, 2000, dtype=np.uint64):
        for s in (0, 42, 2**40 + 5):
            assert L.blo_hash64_u64(int(v), s) == R.ref_hash64_u64(int(v), s)
    for seed in range(6):
        n = 3000 + 17 * seed
        s = O.synth(100 + seed, n)
        if seed % 2:  # breaks kept away from the tail (reference-defined domain)
            for p in rng.integers(0, n - 200, 6):
                s[p] = ord("N")
        for k in (1, 2, 3, 8, 16, 21, 31, 32):
            for canon in (0, 1):
                for complete in (0, 1):
                    assert O.kmer_items(s, k, canon, complete) == O.kmer_items(s, k, canon, complete, lib=R)
        one = np.array([0, n], np.uint64)
        for (k, m, a, b) in ((31, 11, 0, 20), (21, 8, 0, 13), (9, 9, 0, 0), (15, 4, 3, 7)):
            for canon in (0, 1):
                got = O.syncmers(s, one, k, m, a, b, canon, drop_last=True, positions=False)[0]
                assert got == R.ref_syncmer_count(O._ptr(s), n, k, m, a, b, canon)


def test_closed_form_hash_equals_the_general_murmur3():
    """blo_hash64_u64 (MurmurHash3_x64_128 written out for an 8-byte key, what the timed CPU baseline runs) against the byte-wise
    general form on random keys and seeds, seeds beyond 32 bits among them (hash.hpp:16,50 truncates them)"""
    rng = np.random.default_rng(11)
    L = O.oracle()
    for v, s in zip(rng.integers(0, 2**64, 20000, dtype=np.uint64), rng.integers(0, 2**40, 20000, dtype=np.uint64)):
        assert L.blo_hash64_u64(int(v), int(s)) == L.blo_hash64_u64_general(int(v), int(s))
    for v in (0, 1, 2**64 - 1, 0x0123456789abcdef):
        for s in (0, 42, 2**32 - 1, 2**32 + 42):
            assert L.blo_hash64_u64(v, s) == L.blo_hash64_u64_general(v, s)
