#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (it needs oracle/_ref/libbiolib_ref.so, which
`make -C oracle ref` compiles from the unmodified sources under /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Everything written here is data (inputs + expected outputs).  Values come from
  * reference hash::hash64 / double_hash64 / remix            (include/hash.hpp)
  * reference wrapper::kmer_view iteration                     (include/kmer_view.hpp)
  * reference hash::minimizer_position_extractor               (include/kmer_view.hpp:250-283)
  * reference sampler::syncmer_sampler iteration counts        (include/syncmer_sampler.hpp)
and, for minimizer_view / super_kmer_view (which yield nothing / do not compile in the
reference snapshot), from the documented composition
  reference kmer_view(k := unit, complete) + reference hash64 + the brute-force leftmost
  argmin written below in numpy — deliberately NOT the C oracle, so the oracle is checked
  against an independent statement of the same rule (minimizer_view.hpp:283,374).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

R = O.ref()
if R is None:
    sys.exit("oracle/_ref/libbiolib_ref.so missing: run `make -C oracle ref` where /root/reference exists")


def synth(seed, n):
    # SURVEY.md §8d generator, restated in numpy (independent of the C oracle)
    i = np.arange(n, dtype=np.uint64)
    x = (np.uint64(seed) + (i >> np.uint64(5)))
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    code = (x >> (np.uint64(2) * (i & np.uint64(31)))) & np.uint64(3)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[code.astype(np.int64)].copy()


def ref_items(seq, k, canonical, complete):
    return O.kmer_items(seq, k, canonical, complete, lib=R)


def ref_hash(vals, seed):
    return np.array([R.ref_hash64_u64(int(v), seed) for v in vals], dtype=np.uint64)


def ref_units(seq, offsets, unit, canonical):
    """per global position: (value, valid) from reference kmer_view run per sequence, complete set"""
    n = len(seq)
    val = np.zeros(n, np.uint64)
    ok = np.zeros(n, np.uint8)
    for q in range(len(offsets) - 1):
        b, e = int(offsets[q]), int(offsets[q + 1])
        if e - b < unit:
            continue
        for pos, _id, v in ref_items(seq[b:e], unit, canonical, True):
            if v is not None:
                val[b + pos] = v
                ok[b + pos] = 1
    return val, ok


def compose_minimizers(seq, offsets, unit, w, seed, canonical):
    """brute-force leftmost argmin per window of w consecutive valid units; record per change"""
    val, ok = ref_units(seq, offsets, unit, canonical)
    hsh = np.zeros(len(seq), np.uint64)
    idx = np.nonzero(ok)[0]
    hsh[idx] = ref_hash(val[idx], seed)
    recs = []
    for q in range(len(offsets) - 1):
        b, e = int(offsets[q]), int(offsets[q + 1])
        prev = None
        for j in range(b, e - unit - w + 2):
            if not ok[j:j + w].all():
                prev = None
                continue
            win = hsh[j:j + w]
            arg = j + int(np.argmin(win))  # numpy argmin returns the first (leftmost) minimum
            if prev is None or arg != prev:
                recs.append((int(val[arg]), arg, int(hsh[arg]), j))
            prev = arg
    return recs


def compose_super_kmers(seq, offsets, k, m, seed, canonical):
    w = k - m + 1
    val, ok = ref_units(seq, offsets, m, canonical)
    hsh = np.zeros(len(seq), np.uint64)
    idx = np.nonzero(ok)[0]
    hsh[idx] = ref_hash(val[idx], seed)
    groups = []
    for q in range(len(offsets) - 1):
        b, e = int(offsets[q]), int(offsets[q + 1])
        cur = None  # [arg, first, size]
        for j in range(b, e - k + 1):
            valid = bool(ok[j:j + w].all())
            arg = j + int(np.argmin(hsh[j:j + w])) if valid else None
            if cur is not None and (not valid or arg != cur[0]):
                groups.append((int(val[cur[0]]), cur[1], cur[0] - cur[1], cur[2], int(hsh[cur[0]])))
                cur = None
            if valid:
                if cur is None:
                    cur = [arg, j, 0]
                cur[2] += 1
        if cur is not None:
            groups.append((int(val[cur[0]]), cur[1], cur[0] - cur[1], cur[2], int(hsh[cur[0]])))
    return groups


def xor_all(a):
    a = np.asarray(a, dtype=np.uint64)
    return int(np.bitwise_xor.reduce(a)) if len(a) else 0


def with_breaks(seq, positions, chars=b"N"):
    s = seq.copy()
    for i, p in enumerate(positions):
        s[p] = chars[i % len(chars)]
    return s


G = {}

# ---------------------------------------------------------------- a4: hash KATs
kat_vals = [0, 1, 27, 0x0123456789ABCDEF, 2**64 - 1, 0x3FFFFFFFFFFFFFFF, 0x8000000000000000, 6, 36, 0xDEADBEEFCAFEF00D]
kat_seeds = [0, 42, 0x10000002A, 0xFFFFFFFF, 7]
G["hash64_u64"] = [[v, s, R.ref_hash64_u64(v, s)] for v in kat_vals for s in kat_seeds]
dh = np.zeros(2, np.uint64)
G["double_hash64_u64"] = []
for v in kat_vals[:5]:
    for s in (0, 42):
        R.ref_double_hash64_u64(v, s, O._ptr(dh))
        G["double_hash64_u64"].append([v, s, int(dh[0]), int(dh[1])])
rng = np.random.default_rng(20241218)
G["hash64_bytes"] = []
for ln in list(range(0, 41)) + [63, 64, 65, 100]:
    key = rng.integers(0, 256, ln, dtype=np.uint8)
    for s in (0, 42):
        G["hash64_bytes"].append([bytes(key).hex(), s, R.ref_hash64_bytes(O._ptr(key) if ln else None, ln, s)])
G["hash64_u128"] = [[0xFEDCBA9876543210, 0x0123456789ABCDEF, 0, R.ref_hash64_u128(0xFEDCBA9876543210, 0x0123456789ABCDEF, 0)]]
G["hash64_u32"] = [[0xDEADBEEF, 7, R.ref_hash64_u32(0xDEADBEEF, 7)]]
G["remix"] = [[z, R.ref_remix(z)] for z in (0, 1, 42, 2**64 - 1)]

# ---------------------------------------------------------------- a2/a3: tiny strings inside the reference's defined domain
tiny = ["ACGTTGCA", "ACGNTGCAT", "ACGTN", "NACGT", "NNACGTA", "ACNACGTA", "ACGNNNTGCAT", "ACGTNACNGTACGA",
        "acgtUuTgca", "ACGTTGCAGGATCCATTTACGGCA", "AAAAAAAAAAAA", "ACGTACGTACGTACGT", "ACG", "TTTNTTTT"]
G["kmer_items_tiny"] = []
for s in tiny:
    for k in (3,):
        for canon in (0, 1):
            for complete in (0, 1):
                G["kmer_items_tiny"].append(dict(seq=s, k=k, canonical=canon, complete=complete, items=ref_items(s, k, canon, complete)))
s21 = "ACGTTGCAGGATCCATTTACGGCA"
G["kmer_items_k21"] = [dict(seq=s21, k=21, canonical=c, complete=1, items=ref_items(s21, 21, c, 1)) for c in (0, 1)]
s32 = bytes(synth(5, 80)).decode()
G["kmer_items_k32"] = [dict(seq=s32, k=kk, canonical=c, complete=1, items=ref_items(s32, kk, c, 1)) for c in (0, 1) for kk in (31, 32)]

# ---------------------------------------------------------------- a5: syncmer predicate on the survey's example
sx = "ACGTTGCAGGATCCATTTACGGCATTAGC"
mp = np.zeros(64, np.uint64)
n = R.ref_minpos(O._ptr(O.as_bytes(sx)), len(sx), 7, 4, 1, 0, O._ptr(mp), 64)
G["minpos_example"] = dict(seq=sx, k=7, m=4, canonical=1, complete=0, minpos=[int(x) for x in mp[:n]])

# ---------------------------------------------------------------- 1 MiB digests
N1 = 1 << 20
big = synth(42, N1)
G["synth_seed42_first32"] = bytes(big[:32]).decode()
dig = {}
for canon in (0, 1):
    it = ref_items(big, 21, canon, False)
    dig[f"k21_canon{canon}_idiom"] = dict(count=len(it), xor_value=xor_all([v for _, _, v in it]))
it31 = ref_items(big, 31, 1, True)
v31 = np.array([v for _, _, v in it31], dtype=np.uint64)
h31 = ref_hash(v31, 0)
dig["k31_canon1_complete_seed0"] = dict(count=len(v31), xor_value=xor_all(v31), xor_hash=xor_all(h31), sum_hash=int(h31.sum(dtype=np.uint64)))
dig["k31_canon1_idiom_seed0"] = dict(count=len(v31) - 1, xor_value=xor_all(v31[:-1]), xor_hash=xor_all(h31[:-1]), sum_hash=int(h31[:-1].sum(dtype=np.uint64)))
assert dig["k31_canon1_idiom_seed0"]["xor_hash"] == R.ref_scan_kmer_hash_xor(O._ptr(big), N1, 31, 1, 0)
dig["syncmer_k31_s11_0_20_canon1_idiom"] = int(R.ref_syncmer_count(O._ptr(big), N1, 31, 11, 0, 20, 1))
dig["syncmer_k31_s11_0_20_canon0_idiom"] = int(R.ref_syncmer_count(O._ptr(big), N1, 31, 11, 0, 20, 0))
dig["syncmer_k21_s8_0_13_canon1_idiom"] = int(R.ref_syncmer_count(O._ptr(big), N1, 21, 8, 0, 13, 1))
# C3-like: 6,990 reads x 150 bp, unit 31, w 11, seed 42
n3 = 6990 * 150
offs3 = np.arange(0, n3 + 1, 150, dtype=np.uint64)
rec3 = compose_minimizers(big[:n3], offs3, 31, 11, 42, 1)
dig["C3_like_reads150_unit31_w11_seed42"] = dict(n_bases=n3, read_len=150, count=len(rec3), xor_value=xor_all([r[0] for r in rec3]),
                                                 xor_hash=xor_all([r[2] for r in rec3]), xor_pos=xor_all([r[1] for r in rec3]))
# C4-like: 104 reads x 10 kbp, k 31, m 15, seed 42
n4 = 104 * 10000
offs4 = np.arange(0, n4 + 1, 10000, dtype=np.uint64)
grp4 = compose_super_kmers(big[:n4], offs4, 31, 15, 42, 1)
dig["C4_like_reads10k_k31_m15_seed42"] = dict(n_bases=n4, read_len=10000, count=len(grp4), xor_minimizer=xor_all([g[0] for g in grp4]),
                                              xor_hash=xor_all([g[4] for g in grp4]), sum_size=int(sum(g[3] for g in grp4)),
                                              xor_first_pos=xor_all([g[1] for g in grp4]))
G["digests_1MiB_seed42"] = dig

with open(os.path.join(HERE, "kats.json"), "w") as f:
    json.dump(G, f, indent=0, separators=(",", ":"))

# ---------------------------------------------------------------- full arrays on small inputs (npz)
A = {}
small = synth(7, 6000)
ragged_offs = np.array([0, 10, 40, 41, 200, 1500, 1530, 1561, 3000, 3000, 6000], dtype=np.uint64)  # incl. empty + short seqs
broken = with_breaks(small, [100, 101, 777, 2000, 2031, 2062, 2500, 4500, 4533, 5200], b"NnRYK-")
A["small_clean"] = small
A["small_broken"] = broken
A["ragged_offsets"] = ragged_offs
for name, seq in (("clean", small), ("broken", broken)):
    for k in (5, 15, 21, 31, 32):
        for canon in (0, 1):
            val, ok = ref_units(seq, np.array([0, len(seq)], np.uint64), k, canon)
            A[f"units_{name}_k{k}_c{canon}_val"] = val
            A[f"units_{name}_k{k}_c{canon}_ok"] = ok
    # syncmer predicate per yielded item (idiom) from the reference extractor
    for (k, m) in ((31, 11), (21, 8), (7, 4), (15, 15)):
        for canon in (0, 1):
            out = np.zeros(len(seq) + 2, np.uint64)
            n = R.ref_minpos(O._ptr(seq), len(seq), k, m, canon, 1, O._ptr(out), len(out))
            A[f"minpos_{name}_k{k}_m{m}_c{canon}"] = out[:n].astype(np.uint16)
one = np.array([0, len(small)], np.uint64)
reads150 = np.arange(0, 6001, 150, dtype=np.uint64)
for name, seq, offs in (("clean_one", small, one), ("broken_one", broken, one), ("clean_reads150", small, reads150),
                        ("broken_reads150", broken, reads150), ("clean_ragged", small, ragged_offs), ("broken_ragged", broken, ragged_offs)):
    for (unit, w, seed, canon) in ((31, 11, 42, 1), (15, 17, 42, 1), (11, 21, 0, 0), (5, 4, 1, 1), (32, 2, 9, 1), (8, 1, 3, 0)):
        rec = compose_minimizers(seq, offs, unit, w, seed, canon)
        A[f"mm_{name}_u{unit}_w{w}_s{seed}_c{canon}"] = np.array([[r[0], r[1], r[2]] for r in rec], dtype=np.uint64).reshape(-1, 3)
    for (k, m, seed, canon) in ((31, 15, 42, 1), (21, 8, 0, 0), (31, 31, 5, 1)):
        grp = compose_super_kmers(seq, offs, k, m, seed, canon)
        A[f"sk_{name}_k{k}_m{m}_s{seed}_c{canon}"] = np.array([[g[0], g[1], g[2], g[3], g[4]] for g in grp], dtype=np.uint64).reshape(-1, 5)
np.savez_compressed(os.path.join(HERE, "arrays.npz"), **A)
print("wrote kats.json and arrays.npz:", len(G), "json sections,", len(A), "arrays")
