"""GPU parity tests (run with -m gpu on an MI355X): every scan of the C ABI against
  (1) the golden fixtures the unmodified reference produced (tests/golden/),
  (2) the CPU oracle on seeded inputs with breaks, ragged / empty / short sequences and fixed reads,
  (3) size-independent properties at sizes the oracle cannot finish quickly (range unions,
      sortedness, digest-of-digests).
Bit-exact everywhere: integer / index work has no tolerance."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import biolib_amd

    c = biolib_amd.Context(0)
    yield c
    c.close()


MM_CASES = ((31, 11, 42, 1), (15, 17, 42, 1), (11, 21, 0, 0), (5, 4, 1, 1), (32, 2, 9, 1), (8, 1, 3, 0))
SK_CASES = ((31, 15, 42, 1), (21, 8, 0, 0), (31, 31, 5, 1))


def _golden_batch(ctx, A, name):
    seq = A["small_clean" if name.startswith("clean") else "small_broken"]
    kind = name.split("_", 1)[1]
    if kind == "ragged":
        return seq, ctx.upload(seq, A["ragged_offsets"])
    if kind == "reads150":
        return seq, ctx.upload(seq, np.arange(0, len(seq) + 1, 150, dtype=np.uint64))
    return seq, ctx.upload(seq)


def test_library_loaded_is_the_hip_build():
    import biolib_amd

    assert biolib_amd.lib().bl_version() == 100
    with open("/proc/self/maps") as f:
        assert "libbiolib_amd.so" in f.read()


def test_device_synth_matches_generator(ctx, golden_kats):
    b = ctx.synth(42, 100_003, 150)
    got = b.download()
    assert bytes(got[:32]).decode() == golden_kats["synth_seed42_first32"]
    assert np.array_equal(got, O.synth(42, 100_003))
    assert b.n_seqs == (100_003 + 149) // 150


def test_hash_kats_on_device(ctx, golden_kats):
    # a 1-mer .. 32-mer scan exposes the device hash of arbitrary packed values; here: the KAT values as 32-mers
    import biolib_amd

    for v, s, h in golden_kats["hash64_u64"]:
        assert biolib_amd.hash64(v, s) == h
    vals = [v for v, s, h in golden_kats["hash64_u64"] if s == 42]
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for v in vals:
        seq = acgt[[(v >> (62 - 2 * i)) & 3 for i in range(32)]]
        out = ctx.upload(seq).kmers(32, seed=42, canonical=False)
        assert out["count"] == 1 and int(out["values"][0]) == v
        assert int(out["hashes"][0]) == O.oracle().blo_hash64_u64(v, 42)


@pytest.mark.parametrize("name", ["clean", "broken"])
def test_kmers_dense_vs_reference_arrays(ctx, golden_arrays, name):
    A = golden_arrays
    seq = A[f"small_{name}"]
    b = ctx.upload(seq)
    for k in (5, 15, 21, 31, 32):
        for canon in (0, 1):
            out = b.kmers(k, seed=7, canonical=bool(canon))
            ok = A[f"units_{name}_k{k}_c{canon}_ok"]
            val = A[f"units_{name}_k{k}_c{canon}_val"]
            assert np.array_equal(out["valid"], ok)
            assert np.array_equal(out["values"], val)
            exp_h = np.array([O.oracle().blo_hash64_u64(int(v), 7) if o else 0 for v, o in zip(val, ok)], np.uint64)
            assert np.array_equal(out["hashes"], exp_h)
            assert out["count"] == int(ok.sum()) and out["xor_hash"] == O.xor_reduce(exp_h)


def _all_bytes_sequence(spacing):
    """random ACGT with every byte value 0..255 planted once at every position of a 16-byte load (position mod 16), `spacing`
    bases apart (a multiple of 16): the encoder's whole input domain (reference constants.hpp:12-21; the reference indexes its
    table with a signed char, kmer_view.hpp:191, so bytes >= 0x80 lie outside it: this build's contract for them is "break")"""
    assert spacing % 16 == 0
    seq = O.synth(77, 256 * 16 * spacing).copy()
    i = np.arange(256 * 16)
    seq[i * spacing + 16 + (i % 16)] = (i // 16).astype(np.uint8)
    return seq


def test_encoder_over_the_whole_byte_domain(ctx):
    """a1 on the device: the SWAR encoder (encode4 / encode16) sees all 256 byte values in all 16 lanes of its loads, in both
    staging paths (position-tiled and read-tiled), and inside k-mers that straddle lane, wave-tile and workgroup-tile edges"""
    # dense k-mers (kmer_kernel: stage_chunk): validity and values per position for short and long k
    seq = _all_bytes_sequence(48)
    offs = np.array([0, len(seq)], np.uint64)
    b = ctx.upload(seq)
    letters = set(b"ACGTUacgtu")
    for k in (1, 4, 31, 32):
        for canon in (False, True):
            val, ok = O.units(seq, offs, k, canon)
            out = b.kmers(k, seed=3, canonical=canon)
            assert np.array_equal(out["valid"], ok), (k, canon)
            assert np.array_equal(out["values"], val), (k, canon)
            if k == 1:  # the table itself: a position is valid iff its byte is one of the ten letters
                assert np.array_equal(ok.astype(bool), np.isin(seq, list(letters)))
    # window scans: position-tiled (one sequence), read-tiled (fixed 192-bp reads: stage_chunk_frl) and ragged
    for kw, batch in (("one", b), ("reads192", ctx.upload(seq, read_len=192)), ("reads150", ctx.upload(seq[: len(seq) // 150 * 150], read_len=150))):
        s = seq if kw != "reads150" else seq[: len(seq) // 150 * 150]
        o = offs if kw == "one" else np.arange(0, len(s) + 1, 192 if kw == "reads192" else 150, dtype=np.uint64)
        for (unit, w, canon) in ((31, 11, True), (5, 10, False), (15, 17, True)):
            v, p, h = O.minimizers(s, o, unit, w, 42, canon, brute=False)
            got = batch.minimizers(unit, w, seed=42, canonical=canon)
            assert got["count"] == len(v), (kw, unit, w)
            assert np.array_equal(got["values"], v) and np.array_equal(got["positions"], p) and np.array_equal(got["hashes"], h), (kw, unit, w)
        n, pos = O.syncmers(s, o, 31, 11, 0, 20, True)
        got = batch.syncmers(31, 11, 0, 20, canonical=True)
        assert got["count"] == n and np.array_equal(got["positions"], pos), kw


@pytest.mark.parametrize("name", ["clean_one", "broken_one", "clean_reads150", "broken_reads150", "clean_ragged", "broken_ragged"])
def test_minimizers_and_super_kmers_vs_golden(ctx, golden_arrays, name):
    A = golden_arrays
    seq, b = _golden_batch(ctx, A, name)
    for (unit, w, seed, canon) in MM_CASES:
        exp = A[f"mm_{name}_u{unit}_w{w}_s{seed}_c{canon}"]
        got = b.minimizers(unit, w, seed=seed, canonical=bool(canon))
        assert got["count"] == len(exp), (name, unit, w)
        assert np.array_equal(got["values"], exp[:, 0]) and np.array_equal(got["positions"], exp[:, 1]) and np.array_equal(got["hashes"], exp[:, 2])
        assert got["xor_hash"] == O.xor_reduce(exp[:, 2]) and got["xor_pos"] == O.xor_reduce(exp[:, 1]) and got["xor_value"] == O.xor_reduce(exp[:, 0])
    for (k, m, seed, canon) in SK_CASES:
        exp = A[f"sk_{name}_k{k}_m{m}_s{seed}_c{canon}"]
        got = b.super_kmers(k, m, seed=seed, canonical=bool(canon))
        assert got["count"] == len(exp) and got["aux"] == len(exp)
        assert np.array_equal(got["minimizers"], exp[:, 0]) and np.array_equal(got["first_pos"], exp[:, 1])
        assert np.array_equal(got["mm_pos"].astype(np.uint64), exp[:, 2]) and np.array_equal(got["sizes"].astype(np.uint64), exp[:, 3])
        assert np.array_equal(got["hashes"], exp[:, 4])


def test_one_mib_reference_digests(ctx, golden_kats):
    D = golden_kats["digests_1MiB_seed42"]
    n = 1 << 20
    b = ctx.synth(42, n)
    for canon in (0, 1):
        d = b.kmers(21, canonical=bool(canon), drop_last=True, arrays=False)
        assert (d["count"], d["xor_value"]) == (D[f"k21_canon{canon}_idiom"]["count"], D[f"k21_canon{canon}_idiom"]["xor_value"])
    for key, drop in (("k31_canon1_complete_seed0", False), ("k31_canon1_idiom_seed0", True)):
        d = b.kmers(31, seed=0, canonical=True, drop_last=drop, arrays=False)
        assert (d["count"], d["xor_value"], d["xor_hash"], d["sum_hash"]) == (D[key]["count"], D[key]["xor_value"], D[key]["xor_hash"], D[key]["sum_hash"])
    assert b.syncmers(31, 11, 0, 20, canonical=True, drop_last=True, positions=False)["count"] == D["syncmer_k31_s11_0_20_canon1_idiom"]
    assert b.syncmers(31, 11, 0, 20, canonical=False, drop_last=True, positions=False)["count"] == D["syncmer_k31_s11_0_20_canon0_idiom"]
    assert b.syncmers(21, 8, 0, 13, canonical=True, drop_last=True, positions=False)["count"] == D["syncmer_k21_s8_0_13_canon1_idiom"]
    c3 = D["C3_like_reads150_unit31_w11_seed42"]
    r = ctx.synth(42, c3["n_bases"], 150).minimizers(31, 11, seed=42, canonical=True)
    assert (r["count"], r["xor_value"], r["xor_hash"], r["xor_pos"]) == (c3["count"], c3["xor_value"], c3["xor_hash"], c3["xor_pos"])
    c4 = D["C4_like_reads10k_k31_m15_seed42"]
    g = ctx.synth(42, c4["n_bases"], 10000).super_kmers(31, 15, seed=42, canonical=True)
    assert (g["count"], g["xor_value"], g["xor_hash"], int(g["sizes"].sum(dtype=np.uint64)), O.xor_reduce(g["first_pos"])) == (
        c4["count"], c4["xor_minimizer"], c4["xor_hash"], c4["sum_size"], c4["xor_first_pos"])


def _random_case(rng, n, flavour):
    seq = O.synth(int(rng.integers(1, 10**6)), n)
    if flavour == "lowcomplexity":
        motif = np.frombuffer([b"A", b"AC", b"AAAT", b"GATTACA"][int(rng.integers(4))], np.uint8)
        seq = np.resize(motif, n).copy()
        for p in rng.integers(0, max(n, 1), n // 300 + 1):
            if n:
                seq[p] = ord("ACGT"[int(rng.integers(4))])
    if flavour == "mixed_repeats" and n:  # random sequence with tandem-repeat islands: some waves hit hash ties, most do not
        for p in rng.integers(0, n, n // 2000 + 2):
            motif = np.frombuffer([b"A", b"CA", b"TTG", b"ACGTACGA", b"GATTACAGATTACC"][int(rng.integers(5))], np.uint8)
            ln = min(int(rng.integers(20, 3000)), n - int(p))
            seq[p:p + ln] = np.resize(motif, ln)
    if flavour in ("breaks", "ragged_breaks") and n:
        for p in rng.integers(0, n, n // 150 + 1):
            seq[p] = ord("NnRY-"[int(rng.integers(5))])
    if flavour == "case" and n:
        idx = rng.integers(0, n, n // 3)
        seq[idx] = np.frombuffer(b"acgtuU", np.uint8)[rng.integers(0, 6, len(idx))]
    if flavour.startswith("ragged"):
        cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, n // 400 + 3)]))
        offs = np.sort(np.concatenate([cuts, cuts[1:3]])).astype(np.uint64)  # a few empty sequences too
    elif flavour == "reads":
        offs = O.fixed_offsets(n, int(rng.choice([150, 41, 250, 1000])))
    else:
        offs = np.array([0, n], np.uint64)
    return seq, offs


@pytest.mark.parametrize("flavour", ["plain", "breaks", "ragged", "ragged_breaks", "reads", "lowcomplexity", "mixed_repeats", "case"])
def test_all_scans_vs_oracle_random(ctx, flavour):
    import zlib
    rng = np.random.default_rng(zlib.crc32(flavour.encode()))
    for n in (0, 1, 40, 41, 4079, 4081, 20_000, 131_072 + 17):
        seq, offs = _random_case(rng, n, flavour)
        b = ctx.upload(seq, offs)
        for (unit, w, seed, canon) in ((31, 11, 42, 1), (15, 17, 1, 0), (11, 21, 0, 1), (20, 7, 5, 1), (32, 64, 2, 1), (4, 1, 3, 0), (15, 10, 6, 1), (19, 19, 7, 1),
                                       (21, 5, 8, 0)):
            v, p, h = O.minimizers(seq, offs, unit, w, seed, canon, brute=False)
            got = b.minimizers(unit, w, seed=seed, canonical=bool(canon))
            assert got["count"] == len(v), (flavour, n, unit, w)
            assert np.array_equal(got["values"], v) and np.array_equal(got["positions"], p) and np.array_equal(got["hashes"], h)
        for (k, m, seed, canon) in ((31, 15, 42, 1), (21, 8, 0, 0), (40, 9, 4, 1), (24, 15, 6, 1), (27, 9, 7, 0), (19, 15, 8, 1)):
            mn, fp, mp, sz, hs = O.super_kmers(seq, offs, k, m, seed, canon)
            got = b.super_kmers(k, m, seed=seed, canonical=bool(canon))
            assert got["count"] == len(mn) == got["aux"], (flavour, n, k, m)
            assert np.array_equal(got["minimizers"], mn) and np.array_equal(got["first_pos"], fp) and np.array_equal(got["hashes"], hs)
            assert np.array_equal(got["mm_pos"], mp) and np.array_equal(got["sizes"], sz)
        # (31, 11) canonical with offsets other than {0, 20}: scan_count_kernel<SYNCMER,21,11,1,SY=2> (argmins, exact form in scan_redo_kernel);
        # (31, 11) not canonical and the other shapes: the closed form where the offsets are {0, k - s}, else the general argmin kernels
        for (k, s, a, e, canon) in ((31, 11, 0, 20, 1), (31, 11, 0, 20, 0), (21, 8, 0, 13, 1), (15, 15, 0, 0, 1), (32, 12, 3, 9, 1),
                                    (31, 11, 3, 9, 1), (31, 11, 0, 0, 1), (31, 11, 20, 20, 1), (31, 11, 7, 20, 1), (31, 11, 3, 9, 0)):
            for drop in (False, True):
                cnt, pos = O.syncmers(seq, offs, k, s, a, e, canon, drop_last=drop)
                got = b.syncmers(k, s, a, e, canonical=bool(canon), drop_last=drop)
                assert got["count"] == cnt, (flavour, n, k, s, drop)
                assert np.array_equal(got["positions"], pos) and got["xor_pos"] == O.xor_reduce(pos)
        for k in (1, 16, 21, 31, 32):
            val, ok = O.units(seq, offs, k, 1)
            got = b.kmers(k, seed=9, canonical=True)
            assert np.array_equal(got["valid"], ok) and np.array_equal(got["values"], val)
            d = O.kmer_digest(seq, offs, k, True, 9, drop_last=True)
            g2 = b.kmers(k, seed=9, canonical=True, drop_last=True, arrays=False)
            assert (g2["count"], g2["xor_value"], g2["xor_hash"], g2["sum_hash"]) == (d["count"], d["xor_value"], d["xor_hash"], d["sum_hash"])


def test_capacity_error_reports_the_count(ctx):
    import biolib_amd

    b = ctx.synth(3, 50_000, 150)
    full = b.minimizers(31, 11, seed=1, canonical=True)
    r = biolib_amd.Result()
    small = ctx.empty_u64(100)
    with pytest.raises(biolib_amd.BiolibError) as e:
        b.minimizers_raw(31, 11, 1, biolib_amd.FLAG_CANONICAL | biolib_amd.FLAG_SYNC, values=small, positions=None, hashes=None, capacity=100, result=r)
    assert e.value.code == -4 and r.count == full["count"]
    # the first 100 records were still written, in order
    assert np.array_equal(small.cpu().numpy().view(np.uint64)[:100], full["values"][:100])
    # invalid arguments fail loudly
    with pytest.raises(biolib_amd.BiolibError):
        b.minimizers(33, 11)
    with pytest.raises(biolib_amd.BiolibError):
        b.minimizers(31, 65)
    with pytest.raises(biolib_amd.BiolibError):
        b.syncmers(33, 11, 0, 22)


def test_range_union_equals_whole(ctx):
    n = 3_000_000
    b = ctx.synth(11, n, 150)
    whole = b.minimizers(31, 11, seed=42, canonical=True)
    rng = np.random.default_rng(5)
    cuts = [0] + sorted(int(x) for x in rng.integers(1, n, 5)) + [n]
    parts = [b.minimizers(31, 11, seed=42, canonical=True, first=a, n=e - a) for a, e in zip(cuts[:-1], cuts[1:])]
    assert sum(p["count"] for p in parts) == whole["count"]
    assert np.array_equal(np.concatenate([p["positions"] for p in parts]), whole["positions"])
    assert np.array_equal(np.concatenate([p["hashes"] for p in parts]), whole["hashes"])
    x = 0
    for p in parts:
        x ^= p["xor_hash"]
    assert x == whole["xor_hash"]
    sw = b.syncmers(31, 11, 0, 20, canonical=True)
    sp = [b.syncmers(31, 11, 0, 20, canonical=True, first=a, n=e - a) for a, e in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate([p["positions"] for p in sp]), sw["positions"])


def test_large_batch_properties_and_oracle_digest(ctx):
    """64 Mbp of 150-bp reads (>15,000 tiles: exercises the inter-tile look-back): ordered output,
    per-read structure, digest equal to the multi-threaded CPU oracle."""
    n, L = 64_000_050, 150
    b = ctx.synth(42, n, L)
    got = b.minimizers(31, 11, seed=42, canonical=True)
    pos = got["positions"].astype(np.int64)
    assert np.all(np.diff(pos) > 0), "records must be strictly position-ordered"
    # every read of 150 bp has 110 windows, hence at least one record, and all units lie inside their read
    reads = pos // L
    assert np.array_equal(np.unique(reads), np.arange(n // L))
    assert np.all(pos % L <= L - 31)
    # hashes are the hashes of the values
    import biolib_amd
    idx = np.random.default_rng(0).integers(0, got["count"], 2000)
    for i in idx:
        assert biolib_amd.hash64(int(got["values"][i]), 42) == int(got["hashes"][i])
    seq = b.download()
    d = O.minimizer_digest(seq, O.fixed_offsets(n, L), 31, 11, 42, True, threads=16)
    assert (got["count"], got["xor_value"], got["xor_hash"], got["xor_pos"]) == (d["count"], d["xor_value"], d["xor_hash"], d["xor_pos"])
    # digest-only run (no output arrays) agrees
    import biolib_amd as B
    r = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
    assert (r.count, r.xor_hash) == (got["count"], got["xor_hash"])
    # super-k-mers: sizes sum to the number of valid k-mers (every k-mer belongs to exactly one group)
    b2 = ctx.synth(43, 20_000_000, 10_000)
    g = b2.super_kmers(31, 15, seed=42, canonical=True)
    assert int(g["sizes"].sum(dtype=np.uint64)) == (20_000_000 // 10_000) * (10_000 - 30)
    assert g["aux"] == g["count"] and np.all(np.diff(g["first_pos"].astype(np.int64)) > 0)
    assert np.all(g["mm_pos"] <= 16) and np.all(g["sizes"] >= 1) and np.all(g["sizes"] <= 17)
    sy = b2.syncmers(31, 11, 0, 20, canonical=True, positions=False)
    seq2 = b2.download()
    cnt, _ = O.syncmers(seq2[:2_000_000], O.fixed_offsets(2_000_000, 10_000), 31, 11, 0, 20, True, threads=16, positions=False)
    part = b2.syncmers(31, 11, 0, 20, canonical=True, positions=False, first=0, n=2_000_000)
    assert part["count"] == cnt and sy["count"] > cnt


@pytest.mark.parametrize("lanes", [1, 2])
def test_async_scans_on_internal_lanes(lanes):
    """With the context's own streams and two lanes, consecutive asynchronous scans alternate between two streams
    (staggered: pass 2 of one beside pass 1 of the next); results and outputs are valid after ctx.sync() and
    equal the synchronous ones.  Mixed scan kinds share the lanes."""
    import biolib_amd as B

    c = B.Context(0, torch_stream=False, lanes=lanes)
    n = 12_000_000
    b = c.synth(5, n, 150)
    cuts = [(i * 2_000_000, 2_000_000) for i in range(6)]
    cap = 400_000
    outs = [(c.empty_u64(cap), c.empty_u64(cap), c.empty_u64(cap)) for _ in cuts]
    res = [B.Result() for _ in cuts]
    for (a, m), (v, p, h), r in zip(cuts, outs, res):
        b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL, first=a, n=m, values=v, positions=p, hashes=h, capacity=cap, result=r)
    sk = B.Result()
    b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL, first=0, n=n, result=sk)
    # other kernels in flight on the same lanes: a runtime window size, super-k-mers, a runtime-width syncmer scan
    gm, gs, gy = B.Result(), B.Result(), B.Result()
    b.minimizers_raw(19, 13, 7, B.FLAG_CANONICAL, first=0, n=n, result=gm)
    b.super_kmers_raw(27, 12, 5, B.FLAG_CANONICAL, first=0, n=n, result=gs)
    b.syncmers_raw(25, 12, 0, 13, 0, B.FLAG_CANONICAL, first=0, n=n, result=gy)
    c.sync()
    whole = b.minimizers(31, 11, seed=42, canonical=True)
    assert sum(int(r.count) for r in res) == whole["count"] and all(r.status == 0 for r in res)
    pos = np.concatenate([p[: int(r.count)].cpu().numpy().view(np.uint64) for (v, p, h), r in zip(outs, res)])
    hsh = np.concatenate([h[: int(r.count)].cpu().numpy().view(np.uint64) for (v, p, h), r in zip(outs, res)])
    assert np.array_equal(pos, whole["positions"]) and np.array_equal(hsh, whole["hashes"])
    assert int(sk.count) == b.syncmers(31, 11, 0, 20, canonical=True, positions=False)["count"]
    seq = O.synth(5, n)
    offs = O.fixed_offsets(n, 150)
    d = O.minimizer_digest(seq, offs, 19, 13, 7, True, threads=8)
    assert (int(gm.count), int(gm.xor_hash), int(gm.xor_pos)) == (d["count"], d["xor_hash"], d["xor_pos"])
    assert int(gs.count) == len(O.super_kmers(seq[:3_000_000], offs[:20_001], 27, 12, 5, True)[0]) + b.super_kmers_raw(
        27, 12, 5, B.FLAG_CANONICAL | B.FLAG_SYNC, first=3_000_000, n=n - 3_000_000).count
    assert int(gy.count) == O.syncmers(seq, offs, 25, 12, 0, 13, True, positions=False, threads=8)[0]
    with pytest.raises(B.BiolibError):
        B.capi.check(c._lib.bl_ctx_set_lanes(c._h, 3))
    c.close()


def test_random_parameters_vs_oracle(ctx):
    """random (unit, w, seed, strand, read layout) combinations on small inputs with breaks and repeats: minimizers,
    super-k-mers, syncmers and hash samples bit-identical to the oracle (covers the generic runtime-w kernels as well
    as the templated window sizes)"""
    rng = np.random.default_rng(20260)
    for it in range(60 * int(os.environ.get("BL_FUZZ_ROUNDS", "1"))):  # more rounds for an exploratory run
        n = int(rng.integers(50, 20_000))
        flavour = ["plain", "breaks", "ragged", "reads", "mixed_repeats", "lowcomplexity"][int(rng.integers(6))]
        seq, offs = _random_case(rng, n, flavour)
        b = ctx.upload(seq, offs)
        unit, w = int(rng.integers(1, 33)), int(rng.integers(1, 65))
        seed, canon = int(rng.integers(0, 2**40)), bool(rng.integers(2))
        v, p, h = O.minimizers(seq, offs, unit, w, seed, canon, brute=False)
        got = b.minimizers(unit, w, seed=seed, canonical=canon)
        assert got["count"] == len(v), (it, flavour, n, unit, w, canon)
        assert np.array_equal(got["positions"], p) and np.array_equal(got["values"], v) and np.array_equal(got["hashes"], h), (it, flavour, unit, w)
        m = int(rng.integers(1, 33))
        k = int(min(m + rng.integers(0, 40), 95, m + 63))
        mn, fp, mp, sz, hs = O.super_kmers(seq, offs, k, m, seed, canon)
        g = b.super_kmers(k, m, seed=seed, canonical=canon)
        assert g["count"] == len(mn), (it, flavour, n, k, m, canon)
        assert np.array_equal(g["first_pos"], fp) and np.array_equal(g["sizes"], sz) and np.array_equal(g["mm_pos"], mp) and np.array_equal(g["minimizers"], mn)
        kk = int(rng.integers(2, 33))
        s = int(rng.integers(max(1, kk - 31), kk + 1))
        a, e = int(rng.integers(0, kk - s + 1)), int(rng.integers(0, kk - s + 1))
        drop = bool(rng.integers(2))
        cnt, pos = O.syncmers(seq, offs, kk, s, a, e, canon, drop_last=drop)
        gs = b.syncmers(kk, s, a, e, canonical=canon, drop_last=drop)
        assert gs["count"] == cnt and np.array_equal(gs["positions"], pos), (it, flavour, n, kk, s, a, e, canon, drop)
        # the closed shape of the same (k, s) — offsets {0, k - s}, either way round: the run-time width kernels with ties decided per k-mer
        a, e = (0, kk - s) if rng.integers(2) else (kk - s, 0)
        cnt, pos = O.syncmers(seq, offs, kk, s, a, e, canon, drop_last=drop)
        gs = b.syncmers(kk, s, a, e, canonical=canon, drop_last=drop)
        assert gs["count"] == cnt and np.array_equal(gs["positions"], pos), (it, flavour, n, kk, s, a, e, canon, drop, "closed")
        b.close()


@pytest.mark.parametrize("read_len", [100, 101, 125, 150, 151, 250, 300, 76, 10_000])
def test_shape_sweep_shapes_vs_oracle(ctx, read_len):
    """the shapes of tests/perf/shape_sweep.py (profiles/r04_shape_sweep.json) against the oracle, record for record: the BASELINE
    minimizer shape on every fixed read length (the read-tiled kernels for 14 / 15 / 16 units per lane, windows decided on murmur64_top),
    every window width from 2 to 32 (a kernel each) and beyond, closed syncmers of any (k, s) (run-time width kernels) and open ones —
    on clean reads, reads with breaks and reads of repeats (where the approximate forms hand every tile to the exact kernels)"""
    rng = np.random.default_rng(read_len)
    n_reads = max(3, 60_000 // read_len)
    n = n_reads * read_len
    for flavour in ("plain", "breaks", "repeats"):
        seq = O.synth(900 + read_len, n).copy()
        if flavour == "breaks":
            seq[rng.integers(0, n, n // 400 + 1)] = ord("N")
        if flavour == "repeats":
            for p0 in rng.integers(0, n, 12):
                ln = min(int(rng.integers(50, 2500)), n - int(p0))
                seq[p0:p0 + ln] = np.resize(np.frombuffer([b"A", b"CA", b"ACGTTACA", b"GATTACAGATTACC"][int(rng.integers(4))], np.uint8), ln)
        offs = O.fixed_offsets(n, read_len)
        b = ctx.upload(seq, read_len=read_len)
        widths = [(31, 11), (15, 10), (25, 5)] + [(int(rng.integers(1, 33)), w) for w in rng.permutation(np.arange(2, 33))[:6]] + [(15, 33), (31, 64)]
        for (unit, w) in widths:
            if unit + w - 1 > read_len:
                continue
            v, p, h = O.minimizers(seq, offs, unit, w, 42, True, brute=False)
            got = b.minimizers(unit, w, seed=42, canonical=True)
            assert got["count"] == len(v), (read_len, flavour, unit, w)
            assert np.array_equal(got["positions"], p) and np.array_equal(got["values"], v) and np.array_equal(got["hashes"], h), (read_len, flavour, unit, w)
        for (k, s, a, e) in ((31, 11, 0, 20), (31, 15, 0, 16), (21, 11, 0, 10), (31, 8, 0, 23), (31, 8, 23, 0), (25, 12, 0, 13), (20, 16, 0, 4), (15, 5, 0, 10),
                             (32, 1, 0, 31), (12, 11, 0, 1), (31, 11, 3, 9), (21, 8, 2, 5)):
            for canon in (True, False):
                cnt, pos = O.syncmers(seq, offs, k, s, a, e, canon)
                gs = b.syncmers(k, s, a, e, canonical=canon)
                assert gs["count"] == cnt and np.array_equal(gs["positions"], pos), (read_len, flavour, k, s, a, e, canon)
        b.close()


@pytest.mark.gpu
def test_read_tiled_scan_on_tiles_of_repeats(ctx):
    """C3's kernel pair on a batch in which whole tiles are low-complexity reads — nearly every window starts an occurrence,
    3,520 records in a tile instead of ~614 — beside ordinary tiles, with breaks, against the oracle."""
    rng = np.random.default_rng(77)
    L, n_reads = 150, 32 * 9 + 5  # nine full tiles of 32 reads and a partial one
    n = L * n_reads
    seq = O.synth(7, n).copy()
    for t in (1, 4, 5, 8):  # whole tiles of repeats: 32 reads x 110 windows > 1024 records
        motif = np.frombuffer([b"A", b"AC", b"AAAT", b"ACGTT"][t % 4], np.uint8)
        seq[t * 32 * L:(t + 1) * 32 * L] = np.resize(motif, 32 * L)
    seq[rng.integers(0, n, 25)] = ord("N")
    offs = O.fixed_offsets(n, L)
    b = ctx.upload(seq, offs)  # equal-length offsets: a fixed-length batch, scanned by the read-tiled kernels
    for canonical in (True, False):
        v, p, h = O.minimizers(seq, offs, 31, 11, 42, canonical, brute=False)
        got = b.minimizers(31, 11, seed=42, canonical=canonical)
        assert got["count"] == len(v)
        assert np.array_equal(got["values"], v) and np.array_equal(got["positions"], p) and np.array_equal(got["hashes"], h)
    b.close()


@pytest.mark.gpu
def test_windows_decided_on_the_approximate_dword_give_the_records_of_the_hashes(ctx):
    """C3's pass 1 and C5's closed form look at murmur64_top (the hash's high dword without the carry of the low ones, DESIGN.md
    §5.1b) and have the tiles they cannot decide counted again on the hashes.  Against the same scans with
    bl_ctx_set_exact_windows: the same records, array for array, on 0.6 Gbp of random reads (where about one tile in a thousand
    is decided again: Result.redone says how many, and must not be 0 or the second path went untested) and on reads of repeats
    (where every tile is)."""
    import biolib_amd as B

    n = 600_000_000
    cap = n // 6
    flags = B.FLAG_CANONICAL | B.FLAG_SYNC
    for kind in ("minimizers", "closed_syncmers"):
        b = ctx.synth(31 if kind == "minimizers" else 32, n, 150 if kind == "minimizers" else 10_000)
        out = {}
        for exact in (False, True):
            ctx.set_exact_windows(exact)
            v, p, h = ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u64(cap)
            if kind == "minimizers":
                r = b.minimizers_raw(31, 11, 42, flags, values=v, positions=p, hashes=h, capacity=cap)
            else:
                r = b.syncmers_raw(31, 11, 0, 20, 0, flags, positions=p, capacity=cap)
            cnt = int(r.count)
            out[exact] = (r.as_dict(), int(r.redone), p[:cnt].clone(), (v[:cnt].clone(), h[:cnt].clone()) if kind == "minimizers" else None)
        ctx.set_exact_windows(False)
        b.close()
        assert out[False][0] == out[True][0], kind
        assert bool((out[False][2] == out[True][2]).all()), kind
        if kind == "minimizers":
            assert bool((out[False][3][0] == out[True][3][0]).all()) and bool((out[False][3][1] == out[True][3][1]).all())
            assert out[True][1] == 0  # the hashes themselves leave nothing undecided in this kernel (its exact form is inline)
        n_tiles = n // (4800 if kind == "minimizers" else 3700)
        assert 0 < out[False][1] < n_tiles // 50, (kind, out[False][1], n_tiles)  # a few tiles in a thousand
    # reads of repeats: every window ties
    L, n_reads = 150, 32 * 40
    seq = np.resize(np.frombuffer(b"ACGTTACA", np.uint8), L * n_reads).copy()
    b = ctx.upload(seq, O.fixed_offsets(L * n_reads, L))
    v, p, h = O.minimizers(seq, O.fixed_offsets(L * n_reads, L), 31, 11, 42, True, brute=False)
    for exact in (False, True):
        ctx.set_exact_windows(exact)
        got = b.minimizers(31, 11, seed=42, canonical=True)
        assert got["count"] == len(v) and np.array_equal(got["positions"], p) and np.array_equal(got["hashes"], h) and np.array_equal(got["values"], v)
    ctx.set_exact_windows(False)
    b.close()


@pytest.mark.gpu
def test_baseline_kernels_on_random_layouts_and_cuts(ctx):
    """the specialised kernels of the BASELINE configurations (C3 read-tiled and position-tiled, C4, C5 closed syncmers with its
    redo kernel) and the minimap2 widths on random sizes, layouts (one sequence, 150-bp / 10-kbp / ragged reads), breaks and
    repeat islands: whole scans against the oracle, and two ranges cut at a random position against the whole
    (BL_FUZZ_ROUNDS scales the number of cases for an exploratory run)"""
    rng = np.random.default_rng(777)
    for it in range(12 * int(os.environ.get("BL_FUZZ_ROUNDS", "1"))):
        n = int(rng.integers(2_000, 400_000))
        flavour = ["plain", "breaks", "mixed_repeats", "lowcomplexity"][int(rng.integers(4))]
        seq, _ = _random_case(rng, n, flavour)
        layout = int(rng.integers(4))
        if layout == 0:
            offs = np.array([0, n], np.uint64)
        elif layout == 1:
            n = n // 150 * 150
            seq = seq[:n]
            offs = O.fixed_offsets(n, 150)
        elif layout == 2:
            offs = O.fixed_offsets(n, 10_000)
        else:
            cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, n // 3000 + 2)]))
            offs = cuts.astype(np.uint64)
        b = ctx.upload(seq, offs)
        cut = int(rng.integers(1, n)) if layout != 1 else int(rng.integers(1, n // 150)) * 150  # (read-aligned for the read-tiled kernels half of the time)
        if layout == 1 and rng.integers(2):
            cut = int(rng.integers(1, n))
        for unit, w in ((31, 11), (15, 10), (21, 5)):
            v, p, h = O.minimizers(seq, offs, unit, w, 42, True, brute=False)
            got = b.minimizers(unit, w, seed=42, canonical=True)
            assert got["count"] == len(v) and np.array_equal(got["positions"], p) and np.array_equal(got["values"], v) and np.array_equal(got["hashes"], h), (it, flavour, layout, n, unit, w)
            g1, g2 = b.minimizers(unit, w, seed=42, canonical=True, first=0, n=cut), b.minimizers(unit, w, seed=42, canonical=True, first=cut, n=n - cut)
            assert np.array_equal(np.concatenate([g1["positions"], g2["positions"]]), p) and np.array_equal(np.concatenate([g1["hashes"], g2["hashes"]]), h), (it, flavour, layout, n, cut, unit, w)
        mn, fp, mp, sz, hs = O.super_kmers(seq, offs, 31, 15, 42, True)
        g = b.super_kmers(31, 15, seed=42, canonical=True)
        assert g["count"] == len(mn) and np.array_equal(g["first_pos"], fp) and np.array_equal(g["sizes"], sz) and np.array_equal(g["mm_pos"], mp) and np.array_equal(g["minimizers"], mn), (it, flavour, layout, n)
        for drop in (False, True):
            cnt, pos = O.syncmers(seq, offs, 31, 11, 0, 20, True, drop_last=drop)
            gs = b.syncmers(31, 11, 0, 20, canonical=True, drop_last=drop)
            assert gs["count"] == cnt and np.array_equal(gs["positions"], pos), (it, flavour, layout, n, drop)
            s1, s2 = b.syncmers(31, 11, 0, 20, canonical=True, drop_last=drop, first=0, n=cut), b.syncmers(31, 11, 0, 20, canonical=True, drop_last=drop, first=cut, n=n - cut)
            assert np.array_equal(np.concatenate([s1["positions"], s2["positions"]]), pos), (it, flavour, layout, n, cut, drop)
        b.close()
