"""GPU tests for SURVEY.md §8f rank 2: hash_sampler over kmer_view, sort + unique, Jaccard of two k-mer sets —
the pipeline of the reference's tests/test_jaccard.cpp — against numpy on the oracle's k-mers."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    import biolib_amd

    c = biolib_amd.Context(0)
    yield c
    c.close()


def _idiom_kmers(seq, offs, k, canonical):
    """what the reference loop `for (it = cbegin(); it != cend(); ++it) if ((*it).value) push_back` collects"""
    val, ok = O.units(seq, offs, k, canonical)
    keep = ok.astype(bool)
    ends = (np.asarray(offs[1:], dtype=np.int64) - k)  # the k-mer that ends each sequence is never visited (Q1)
    ends = ends[(ends >= 0)]
    keep[ends[ends < len(keep)]] = False
    return val, keep


@pytest.mark.parametrize("flavour", ["plain", "breaks", "ragged"])
def test_hash_sample_vs_oracle(ctx, flavour):
    rng = np.random.default_rng(11)
    n = 50_000
    seq = O.synth(77, n)
    if flavour == "breaks":
        seq[rng.integers(0, n, 200)] = ord("N")
    offs = np.array([0, n], np.uint64) if flavour != "ragged" else np.unique(np.concatenate([[0, n], rng.integers(0, n, 60)])).astype(np.uint64)
    b = ctx.upload(seq, offs)
    L = O.oracle()
    for k, canon in ((21, True), (31, False), (5, True)):
        val, ok = O.units(seq, offs, k, canon)
        hsh = np.array([L.blo_hash64_u64(int(v), 42) for v in val], np.uint64)
        for rate in (1.0, 0.25, 0.0):
            thr = 2**64 - 1 if rate >= 1.0 else int(rate * float(2**64 - 1))
            exp = np.nonzero(ok.astype(bool) & (hsh < np.uint64(thr)))[0]
            got = b.hash_sample(k, seed=42, threshold=thr, canonical=canon)
            assert got["count"] == len(exp), (flavour, k, rate)
            assert np.array_equal(got["positions"], exp.astype(np.uint64)) and np.array_equal(got["values"], val[exp]) and np.array_equal(got["hashes"], hsh[exp])
        vals, keep = _idiom_kmers(seq, offs, k, canon)
        got = b.hash_sample(k, seed=42, canonical=canon, drop_last=True)
        assert np.array_equal(got["values"], vals[keep])
        same = b.minimizers(k, 1, seed=42, canonical=canon)  # w = 1 is the plain unit list
        assert np.array_equal(same["values"], vals[ok.astype(bool)])


def test_sort_unique_and_jaccard(ctx):
    import torch

    rng = np.random.default_rng(5)
    a = rng.integers(0, 5000, 40_000).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    bb = rng.integers(2500, 9000, 70_000).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    ta = torch.from_numpy(a.view(np.int64)).cuda()
    tb = torch.from_numpy(bb.view(np.int64)).cuda()
    na, nb = ctx.sort_unique(ta), ctx.sort_unique(tb)
    ua, ub = np.unique(a), np.unique(bb)
    assert na == len(ua) and nb == len(ub)
    assert np.array_equal(ta[:na].cpu().numpy().view(np.uint64), ua) and np.array_equal(tb[:nb].cpu().numpy().view(np.uint64), ub)
    inter, uni = ctx.jaccard(ta, na, tb, nb)
    assert inter == len(np.intersect1d(ua, ub)) and uni == len(np.union1d(ua, ub))
    assert ctx.jaccard(tb, nb, ta, na) == (inter, uni)
    assert ctx.jaccard(ta, na, ta, na) == (na, na) and ctx.jaccard(ta, 0, tb, nb) == (0, nb)


def test_jaccard_of_two_fasta_files_like_the_reference_tool(ctx):
    """tests/test_jaccard.cpp:55-130 of the reference: canonical k-mers of two files (idiom loop), sorted, unique, merged."""
    import biolib_amd

    exp = json.load(open(os.path.join(HERE, "golden", "ingest", "expected.json")))
    k = 15
    sets = []
    for fn in ("many.fa.gz", "mixed.fa"):
        parts = []
        for batch, names, offs in biolib_amd.Reader(os.path.join(HERE, "golden", "ingest", fn)).batches(ctx, 20_000):
            r = batch.hash_sample(k, canonical=True, drop_last=True, device=True)
            parts.append(r["values_device"][: r["n"]])
        import torch
        keys = torch.cat(parts)
        n = ctx.sort_unique(keys)
        sets.append((keys, n))
        # oracle: the same set from the sequences the reference reader returned
        seqs = exp[fn]["seqs"]
        seq = np.frombuffer("".join(seqs).encode("latin1"), np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
        vals, keep = _idiom_kmers(seq, offs, k, True)
        assert np.array_equal(keys[:n].cpu().numpy().view(np.uint64), np.unique(vals[keep]))
    inter, uni = ctx.jaccard(sets[0][0], sets[0][1], sets[1][0], sets[1][1])
    a = sets[0][0][: sets[0][1]].cpu().numpy().view(np.uint64)
    b = sets[1][0][: sets[1][1]].cpu().numpy().view(np.uint64)
    assert (inter, uni) == (len(np.intersect1d(a, b)), len(np.union1d(a, b)))


# ---- §8f rank 4: partitioned counting (bucket split -> exchange -> sort -> run-length count)
@pytest.mark.parametrize("parts", [1, 2, 8, 64])
def test_partition_by_owner(ctx, parts):
    import torch

    rng = np.random.default_rng(parts)
    keys = rng.integers(0, 2**63, 300_001).astype(np.uint64)
    keys[::7] = keys[0]  # heavy duplicates land in one bucket
    t = torch.from_numpy(keys.view(np.int64).copy()).cuda()
    out, counts = ctx.partition(t, parts, seed=5)
    got = out.cpu().numpy().view(np.uint64)
    owner = O.hash64_np(keys, 5) % np.uint64(parts)
    assert counts == np.bincount(owner.astype(np.int64), minlength=parts).tolist()
    edges = np.concatenate([[0], np.cumsum(counts)])
    for b in range(parts):
        seg = got[edges[b]:edges[b + 1]]
        assert np.all(O.hash64_np(seg, 5) % np.uint64(parts) == b)
        assert np.array_equal(np.sort(seg), np.sort(keys[owner == b]))
    with pytest.raises(Exception):
        ctx.partition(t, 65)


def test_sort_count_vs_numpy(ctx):
    import torch

    seq = O.synth(3, 400_000)
    vals, ok = O.units(seq, O.fixed_offsets(len(seq), 150), 9, True)   # 4^9 keys: plenty of repeats
    keys = vals[ok != 0]
    t = torch.from_numpy(keys.view(np.int64).copy()).cuda()
    u, c = ctx.sort_count(t)
    eu, ec = np.unique(keys, return_counts=True)
    assert np.array_equal(u.cpu().numpy().view(np.uint64), eu) and np.array_equal(c.cpu().numpy().astype(np.int64), ec)
    assert np.array_equal(t.cpu().numpy().view(np.uint64), np.sort(keys))
    e = ctx.sort_count(ctx.empty_u64(0), n=0)
    assert e[0].numel() == 0 and e[1].numel() == 0


def test_exchange_and_count_single_rank_rccl(ctx):
    """the whole distributed counter on one GPU: scan -> partition -> all-to-all (RCCL, world 1) -> sort -> count"""
    import torch
    import torch.distributed as dist

    from biolib_amd.shard import exchange_and_count

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        n = 2_000_000
        b = ctx.synth(12, n, 150)
        import biolib_amd as B

        span = n
        v, ok_t = ctx.empty_u64(span), ctx.empty_u8(span)
        b.kmers_raw(11, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, values=v, valid=ok_t)
        vals = v[:span][ok_t[:span].bool()].contiguous()               # device-resident k-mers straight from the scan
        seq = O.synth(12, n)
        ev, ok = O.units(seq, O.fixed_offsets(n, 150), 11, True)
        u, c = exchange_and_count(vals, lambda k, parts: ctx.partition(k, parts), lambda k: ctx.sort_count(k))
        eu, ec = np.unique(ev[ok != 0], return_counts=True)
        assert np.array_equal(u.cpu().numpy().view(np.uint64), eu) and np.array_equal(c.cpu().numpy().astype(np.int64), ec)
    finally:
        if created:
            dist.destroy_process_group()


def _sorted_counts(u, c):
    """(keys ascending, multiplicities) from the unordered output of the bucketed counter"""
    u = u.cpu().numpy().view(np.uint64)
    c = c.cpu().numpy().astype(np.int64)
    order = np.argsort(u, kind="stable")
    return u[order], c[order]


def _np_pack(seq, fp, sz, k, mp=None):
    """numpy restatement of the 16-byte packed super-k-mer record"""
    code = np.zeros(256, np.uint64)
    for ch, c in zip(b"ACGTUacgtu", (0, 1, 2, 3, 3, 0, 1, 2, 3, 3)):
        code[ch] = c
    out = np.zeros((len(fp), 2), np.uint64)
    for g, (p, s) in enumerate(zip(fp.tolist(), sz.tolist())):
        nb = s + k - 1
        c = code[seq[p:p + nb]]
        hi = 0
        for i in range(min(nb, 32)):
            hi |= int(c[i]) << (62 - 2 * i)
        lo = (s - 1) | ((int(mp[g]) if mp is not None else 0) << 5)
        for i in range(32, nb):
            lo |= int(c[i]) << (62 - 2 * (i - 32))
        out[g] = (hi, lo)
    return out


@pytest.mark.parametrize("k,m,canon", [(31, 15, True), (31, 15, False), (32, 5, True), (21, 11, True), (5, 5, True), (16, 1, False)])
def test_super_kmer_records_pack_and_expand(ctx, k, m, canon):
    n, L = 120_000, 1500
    seq = O.synth(31, n)
    seq[np.random.default_rng(k).integers(0, n, 25)] = ord("N")
    offs = O.fixed_offsets(n, L)
    b = ctx.upload(seq, offs)
    mn, fp, mp, sz, hs = O.super_kmers(seq, offs, k, m, 9, canon)
    recs, hashes = b.super_kmer_records(k, m, seed=9, canonical=canon)
    assert recs.shape[0] == len(mn) and np.array_equal(hashes.cpu().numpy().view(np.uint64), hs)
    assert np.array_equal(recs.cpu().numpy().view(np.uint64), _np_pack(seq, fp, sz, k, mp))
    # expansion = the k-mers of each group, group after group = the oracle's units at first_pos .. first_pos+size-1
    vals, ok = O.units(seq, offs, k, canon)
    idx = np.concatenate([np.arange(p, p + s) for p, s in zip(fp.tolist(), sz.tolist())]) if len(fp) else np.zeros(0, np.int64)
    assert np.all(ok[idx] != 0)
    got = ctx.expand_super_kmers(recs, k, canonical=canon)
    assert np.array_equal(got.cpu().numpy().view(np.uint64), vals[idx])
    # routing: buckets are the records whose minimizer hash % parts == b
    for parts in (1, 3, 8):
        out, counts = ctx.partition_records(hashes, recs, parts)
        got_r = out.cpu().numpy().view(np.uint64)
        owner = hs % np.uint64(parts)
        assert counts == np.bincount(owner.astype(np.int64), minlength=parts).tolist()
        edges = np.concatenate([[0], np.cumsum(counts)])
        exp = _np_pack(seq, fp, sz, k, mp)
        for bkt in range(parts):
            a = got_r[edges[bkt]:edges[bkt + 1]]
            e = exp[owner == bkt]
            assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], e[np.lexsort((e[:, 1], e[:, 0]))])


@pytest.mark.parametrize("layout", ["reads150", "ragged", "one_contig"])
@pytest.mark.parametrize("k,m,canon", [(31, 15, True), (31, 15, False), (32, 5, True), (21, 11, True), (9, 9, True), (27, 3, False)])
def test_super_kmer_records_from_the_scan_equal_scan_then_pack(ctx, layout, k, m, canon):
    """bl_scan_super_kmer_records (records built inside the scan from its 2-bit codes) against bl_scan_super_kmers +
    bl_pack_super_kmers and against the oracle's groups, on the read-tiled layout, on ragged reads with breaks and on one sequence"""
    rng = np.random.default_rng(k * 100 + m)
    if layout == "reads150":
        n = 150 * 40_000
        offs = O.fixed_offsets(n, 150)
    elif layout == "ragged":
        lens = rng.integers(1, 700, 9000)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        n = int(offs[-1])
    else:
        n = 3_000_017
        offs = np.array([0, n], np.uint64)
    seq = O.synth(77, n)
    seq[rng.integers(0, n, n // 3000)] = ord("N")
    b = ctx.upload(seq, offs)
    fused, h1 = b.super_kmer_records(k, m, seed=5, canonical=canon)
    plain, h2 = b.super_kmer_records(k, m, seed=5, canonical=canon, fused=False)
    assert fused.shape == plain.shape and bool((fused == plain).all()) and bool((h1 == h2).all())
    mn, fp, mp, sz, hs = O.super_kmers(seq, offs, k, m, 5, canon)
    assert fused.shape[0] == len(mn) and np.array_equal(h1.cpu().numpy().view(np.uint64), hs)
    assert np.array_equal(fused.cpu().numpy().view(np.uint64)[:2000], _np_pack(seq, fp[:2000], sz[:2000], k, mp[:2000]))
    # a sub-range, and a capacity that is too small
    first, cnt = (150 * 1000, 150 * 5000) if layout == "reads150" else (int(offs[len(offs) // 3]), int(offs[-1] - offs[len(offs) // 3]) // 2 if layout != "one_contig" else n // 2)
    if layout != "one_contig":
        cnt = int(offs[np.searchsorted(offs, first + cnt)] - first)  # (ranges end where a sequence ends)
    f2, _ = b.super_kmer_records(k, m, seed=5, canonical=canon, first=first, n=cnt)
    p2, _ = b.super_kmer_records(k, m, seed=5, canonical=canon, first=first, n=cnt, fused=False)
    assert f2.shape == p2.shape and bool((f2 == p2).all())
    # a batch that is a piece of a longer whole (bl_batch_set_origin): the scan reports origin + position, and the unfused path packs
    # from those positions — the same records as without an origin (the records hold bases, not positions)
    b.set_origin(10**12 + 7)
    f3, _ = b.super_kmer_records(k, m, seed=5, canonical=canon, first=first, n=cnt)
    p3, _ = b.super_kmer_records(k, m, seed=5, canonical=canon, first=first, n=cnt, fused=False)
    assert f3.shape == f2.shape and bool((f3 == f2).all()) and bool((p3 == f2).all())
    b.close()


@pytest.mark.parametrize("copies,distinct", [(40, 4000), (3, 4000), (40, 40_000)])
def test_count_super_kmers_on_reads_of_high_coverage(ctx, copies, distinct):
    """reads that repeat (sequencing coverage): few distinct minimizers, each many times over, so that most buckets hold more
    records than one wave's table takes and go to the sort-and-count path — every one of them, however many there are"""
    k, m, L = 31, 15, 150  # (the largest case lists some 10^5 buckets: more than the list of oversized buckets used to hold)
    base = O.synth(123, distinct * L).reshape(distinct, L)
    order = np.random.default_rng(5).permutation(distinct * copies) % distinct
    seq = np.ascontiguousarray(base[order]).reshape(-1)
    offs = O.fixed_offsets(seq.size, L)
    b = ctx.upload(seq, offs)
    recs, _ = b.super_kmer_records(k, m, seed=42, canonical=True)
    u, c = ctx.count_super_kmers(recs, k, m, seed=42, canonical=True)
    vals, ok = O.units(base.reshape(-1), O.fixed_offsets(distinct * L, L), k, True)
    eu, ec = np.unique(vals[ok != 0], return_counts=True)
    got_u = u.cpu().numpy().view(np.uint64)
    at = np.argsort(got_u)
    assert np.array_equal(got_u[at], eu) and np.array_equal(c.cpu().numpy().astype(np.int64)[at], ec * copies)
    b.close()


def test_super_kmer_record_limits(ctx):
    import biolib_amd as B

    b = ctx.synth(1, 10_000, 100)
    with pytest.raises(B.BiolibError):
        b.super_kmer_records(32, 4)  # 2k - m = 60 bases do not fit a record (59 do)


def test_count_kmers_via_super_kmers_single_gpu(ctx):
    """scan -> pack -> (route) -> expand -> sort -> count equals the multiset of canonical k-mers of the oracle;
    with a world-1 RCCL group the all-to-all path runs too"""
    import torch
    import torch.distributed as dist

    from biolib_amd.shard import count_kmers_via_super_kmers, exchange

    n, L = 3_000_000, 150
    b = ctx.synth(77, n, L)
    seq = O.synth(77, n)
    vals, ok = O.units(seq, O.fixed_offsets(n, L), 15, True)          # 4^15/2 canonical 15-mers over 2.7 M: repeats exist
    eu, ec = np.unique(vals[ok != 0], return_counts=True)
    u, c = _sorted_counts(*count_kmers_via_super_kmers(ctx, b, 15, 9, seed=42, canonical=True))
    assert np.array_equal(u, eu) and np.array_equal(c, ec) and ec.max() > 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        recs, hashes = b.super_kmer_records(15, 9, seed=42, canonical=True)
        bucketed, counts = ctx.partition_records(hashes, recs, 1)
        inbox = exchange(bucketed, counts)
        assert inbox.shape == recs.shape
        u2, c2 = ctx.sort_count(ctx.expand_super_kmers(inbox, 15, canonical=True))
        assert np.array_equal(u2.cpu().numpy().view(np.uint64), u) and np.array_equal(c2.cpu().numpy().astype(np.int64), c)
    finally:
        if created:
            dist.destroy_process_group()


def test_exchange_large_payload_rccl_own_streams():
    """A context on its OWN (non-blocking) streams chained behind a large RCCL all-to-all: the library must not read the
    inbox before the collective has delivered it (the wrapper synchronises torch's stream first).  60 Mbp -> ~100 MB of
    16-byte records through all_to_all_single, then expand + sort + count, against the borrowed-stream context's result."""
    import torch
    import torch.distributed as dist

    import biolib_amd as B
    from biolib_amd.shard import count_kmers_via_super_kmers

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29536")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        n, L = 60_000_000, 150
        own = B.Context(0, torch_stream=False, lanes=2)
        ref = B.Context(0)
        u1, c1 = count_kmers_via_super_kmers(own, own.synth(9, n, L), 31, 15, seed=42, canonical=True, group=dist.group.WORLD, force_exchange=True)
        u2, c2 = count_kmers_via_super_kmers(ref, ref.synth(9, n, L), 31, 15, seed=42, canonical=True)
        a, b2 = _sorted_counts(u1, c1), _sorted_counts(u2, c2)
        assert u1.numel() > 40_000_000 and np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])
        own.close()
        ref.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("k,m,canon,kind", [(31, 15, True, "random"), (21, 11, False, "random"), (31, 15, True, "repeats"), (32, 9, True, "random"), (15, 9, True, "breaks")])
def test_count_super_kmers_bucketed(ctx, k, m, canon, kind):
    """bl_count_super_kmers (minimizer buckets counted in LDS hash tables, no global sort) against np.unique of the oracle's k-mers:
    random reads; low-complexity reads whose few minimizers make buckets far too big for a table (the sort fallback counts
    them); reads with breaks; capacity too small -> the exact need comes back"""
    import biolib_amd as B

    n, L = 1_200_000, 150
    seq = O.synth(61 + k, n)
    if kind == "repeats":  # a third of the reads are poly-A / dinucleotide repeats: thousands of copies of very few k-mers
        r = seq.reshape(-1, L)
        r[::3] = np.frombuffer(b"A" * L, np.uint8)
        r[1::9] = np.frombuffer((b"AC" * L)[:L], np.uint8)
    if kind == "breaks":
        seq[np.random.default_rng(k).integers(0, n, n // 500)] = ord("N")
    offs = O.fixed_offsets(n, L)
    b = ctx.upload(seq, offs)
    recs, _ = b.super_kmer_records(k, m, seed=7, canonical=canon)
    u, c = _sorted_counts(*ctx.count_super_kmers(recs, k, m, seed=7, canonical=canon))
    vals, ok = O.units(seq, offs, k, canon)
    eu, ec = np.unique(vals[ok != 0], return_counts=True)
    assert np.array_equal(u, eu) and np.array_equal(c, ec)
    if kind == "repeats":
        assert ec.max() > 10_000
    # too small an output: the need is reported, nothing is written past the end
    import ctypes as C

    from biolib_amd import capi

    need = C.c_uint64()
    small = ctx.empty_u64(100)
    import torch

    cs = torch.empty(100, dtype=torch.int32, device="cuda")
    rc = capi.lib().bl_count_super_kmers(ctx._h, C.c_void_p(recs.data_ptr()), recs.shape[0], k, m, 7, B.FLAG_CANONICAL if canon else 0, C.c_void_p(small.data_ptr()),
                                         C.c_void_p(cs.data_ptr()), 100, C.byref(need))
    assert rc == capi.BL_ERR_CAPACITY and need.value == len(eu)


TWO_RANK_COUNTER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import biolib_amd as B
from biolib_amd.shard import shard_reads, count_kmers_via_super_kmers
dist.init_process_group("gloo")                      # two ranks on ONE GPU: RCCL refuses that, gloo stages through the host
rank, world = dist.get_rank(), dist.get_world_size()
ctx = B.Context(0)
L, n_reads, k, m = 150, 40_000, 21, 11
first, cnt = shard_reads(n_reads, world, rank)
import oracle_lib as O
whole = O.synth(5, n_reads * L)
whole[::997] = ord("N")
seq = whole[first * L:(first + cnt) * L]
b = ctx.upload(seq, O.fixed_offsets(len(seq), L))
u, c = count_kmers_via_super_kmers(ctx, b, k, m, seed=3, canonical=True)
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), u=u.cpu().numpy().view(np.uint64), c=c.cpu().numpy())  # unordered: the test sorts
ctx.close()
dist.destroy_process_group()
"""


def test_two_rank_counter_on_one_gpu(tmp_path):
    """the whole distributed counter with world_size 2 (both ranks on this GPU, gloo for the exchange): scan -> pack ->
    route by minimizer hash -> all-to-all -> expand -> sort -> count; the union over ranks is the exact global count and
    no k-mer appears on two ranks"""
    import subprocess
    import sys

    root = os.path.dirname(HERE)
    script = tmp_path / "counter.py"
    script.write_text(TWO_RANK_COUNTER.format(root=root, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    L, n_reads, k = 150, 40_000, 21
    whole = O.synth(5, n_reads * L)
    whole[::997] = ord("N")
    vals, ok = O.units(whole, O.fixed_offsets(len(whole), L), k, True)
    eu, ec = np.unique(vals[ok != 0], return_counts=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    assert len(np.intersect1d(parts[0]["u"], parts[1]["u"])) == 0
    gu = np.concatenate([p["u"] for p in parts])
    gc = np.concatenate([p["c"] for p in parts]).astype(np.int64)
    order = np.argsort(gu)
    assert np.array_equal(gu[order], eu) and np.array_equal(gc[order], ec)
    assert min(len(p["u"]) for p in parts) > 0.3 * len(eu)   # both ranks own a fair share
