"""GPU edge-case and property tests: limits of the C ABI, alternative batch constructors, extremes of k / w,
super-k-mer and syncmer range unions, multi-context use from threads, size-independent properties at 1 Gbp."""
import os
import threading

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


def test_batch_from_device_tensor_offsets_and_fixed_reads(ctx):
    import torch

    n = 300_007
    seq = O.synth(21, n)
    seq[[5, 77_777, 150_000, n - 1]] = ord("N")
    t = torch.from_numpy(seq.copy()).cuda()
    rng = np.random.default_rng(2)
    offs = np.unique(np.concatenate([[0, n], rng.integers(0, n, 500)])).astype(np.uint64)
    for batch, o in ((ctx.from_tensor(t, offsets=offs), offs), (ctx.from_tensor(t, read_len=151), O.fixed_offsets(n, 151)), (ctx.from_tensor(t), np.array([0, n], np.uint64))):
        assert batch.n_bases == n and batch.n_seqs == len(o) - 1
        v, p, h = O.minimizers(seq, o, 31, 11, 42, True, brute=False)
        got = batch.minimizers(31, 11, seed=42, canonical=True)
        assert np.array_equal(got["positions"], p) and np.array_equal(got["values"], v) and np.array_equal(got["hashes"], h)
    with pytest.raises(Exception):
        ctx.from_tensor(t[1:])  # not 16-byte aligned


def test_extreme_parameters_vs_oracle(ctx):
    n = 60_000
    seq = O.synth(8, n)
    seq[np.random.default_rng(8).integers(0, n, 40)] = ord("N")
    offs = O.fixed_offsets(n, 777)
    b = ctx.upload(seq, offs)
    for (unit, w) in ((1, 1), (1, 64), (32, 1), (32, 64), (2, 33), (17, 48), (31, 16), (31, 17), (16, 32)):
        for canon in (0, 1):
            v, p, h = O.minimizers(seq, offs, unit, w, 7, canon, brute=False)
            got = b.minimizers(unit, w, seed=7, canonical=bool(canon))
            assert got["count"] == len(v), (unit, w, canon)
            assert np.array_equal(got["positions"], p) and np.array_equal(got["values"], v) and np.array_equal(got["hashes"], h)
    for (k, m) in ((32, 32), (64, 1), (95, 32), (33, 2), (48, 17)):
        mn, fp, mp, sz, hs = O.super_kmers(seq, offs, k, m, 3, 1)
        got = b.super_kmers(k, m, seed=3, canonical=True)
        assert got["count"] == len(mn) == got["aux"], (k, m)
        assert np.array_equal(got["first_pos"], fp) and np.array_equal(got["sizes"], sz) and np.array_equal(got["mm_pos"], mp) and np.array_equal(got["minimizers"], mn)
    for (k, s, a, e) in ((32, 1, 0, 31), (32, 32, 0, 0), (2, 1, 0, 1), (31, 30, 0, 1), (20, 5, 7, 7)):
        for canon in (0, 1):
            cnt, pos = O.syncmers(seq, offs, k, s, a, e, canon)
            got = b.syncmers(k, s, a, e, canonical=bool(canon))
            assert got["count"] == cnt and np.array_equal(got["positions"], pos), (k, s, canon)


def test_range_unions_for_super_kmers_and_syncmers(ctx):
    """ranges aligned to read boundaries compose exactly for super-k-mers (a range cuts groups otherwise, by contract)"""
    L, n_reads = 1000, 3000
    n = L * n_reads
    b = ctx.synth(17, n, L)
    whole = b.super_kmers(31, 15, seed=42, canonical=True)
    cuts = [0, 700 * L, 701 * L, 2000 * L, n]
    parts = [b.super_kmers(31, 15, seed=42, canonical=True, first=a, n=e - a) for a, e in zip(cuts[:-1], cuts[1:])]
    assert sum(p["count"] for p in parts) == whole["count"]
    for key in ("first_pos", "minimizers", "sizes", "mm_pos", "hashes"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), whole[key]), key
    # unaligned cut: group boundaries are cut at the range boundary, k-mers are still covered exactly once
    a = b.super_kmers(31, 15, seed=42, canonical=True, first=0, n=12_345)
    c = b.super_kmers(31, 15, seed=42, canonical=True, first=12_345, n=n - 12_345)
    assert int(a["sizes"].sum(dtype=np.uint64)) + int(c["sizes"].sum(dtype=np.uint64)) == int(whole["sizes"].sum(dtype=np.uint64))
    sw = b.syncmers(31, 11, 0, 20, canonical=True, drop_last=True)
    sp = [b.syncmers(31, 11, 0, 20, canonical=True, drop_last=True, first=x, n=y - x) for x, y in zip([0, 999, 50_001, 2_222_222], [999, 50_001, 2_222_222, n])]
    assert np.array_equal(np.concatenate([p["positions"] for p in sp]), sw["positions"])


def test_limits_and_errors(ctx):
    import biolib_amd as B

    b = ctx.synth(1, 10_000, 100)
    with pytest.raises(B.BiolibError):
        b.minimizers_raw(31, 11, 0, B.FLAG_SYNC, first=20_000)  # range starts beyond the batch
    with pytest.raises(B.BiolibError):
        b.super_kmers(31, 0)
    with pytest.raises(B.BiolibError):
        b.super_kmers(10, 11)
    with pytest.raises(B.BiolibError):
        b.kmers(0)
    with pytest.raises(B.BiolibError):
        b.kmers(33)
    # capacity errors report the count for every compacting scan
    r = B.Result()
    small = ctx.empty_u64(8)
    with pytest.raises(B.BiolibError) as e:
        b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, positions=small, capacity=8, result=r)
    assert e.value.code == -4 and r.count == b.syncmers(31, 11, 0, 20, canonical=True, positions=False)["count"]
    r = B.Result()
    with pytest.raises(B.BiolibError) as e:
        b.super_kmers_raw(31, 15, 0, B.FLAG_SYNC, minimizers=small, capacity=8, result=r)
    assert e.value.code == -4 and r.count == b.super_kmers(31, 15)["count"]
    # an empty range and an empty batch are fine
    assert b.minimizers(31, 11, first=5_000, n=0)["count"] > 0  # n = 0 means "to the end"
    empty = ctx.upload(b"")
    assert empty.minimizers(31, 11)["count"] == 0 and empty.kmers(5)["count"] == 0 and empty.syncmers(31, 11, 0, 20)["count"] == 0
    # more than 2^31 positions in one range is refused (split it)
    big = ctx.synth(1, (1 << 31) + 4096)
    with pytest.raises(B.BiolibError):
        big.minimizers_raw(31, 11, 0, B.FLAG_SYNC)
    r = big.minimizers_raw(31, 11, 0, B.FLAG_SYNC | B.FLAG_CANONICAL, first=(1 << 31) - 1000, n=5000)  # a range straddling 2^31 works
    assert r.count > 0
    big.close()


def test_two_contexts_in_two_threads():
    import biolib_amd as B

    results = {}

    def work(seed):
        c = B.Context(0, torch_stream=False)
        b = c.synth(seed, 3_000_000, 150)
        results[seed] = b.minimizers(31, 11, seed=42, canonical=True)["xor_hash"]
        c.close()

    ts = [threading.Thread(target=work, args=(s,)) for s in (101, 102)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for s in (101, 102):
        seq = O.synth(s, 3_000_000)
        assert results[s] == O.minimizer_digest(seq, O.fixed_offsets(3_000_000, 150), 31, 11, 42, True, threads=8)["xor_hash"]


def test_one_gbp_properties(ctx):
    """size-independent properties at 1 Gbp of 10-kbp reads (the oracle would need minutes): chunk composition,
    every k-mer in exactly one super-k-mer, sorted outputs, syncmer count sub-additivity across ranges."""
    L = 10_000
    n = 1_000_000_000
    b = ctx.synth(7, n, L)
    import biolib_amd as B

    half = (n // 2) // L * L
    r1 = b.super_kmers_raw(31, 15, 42, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=half)
    r2 = b.super_kmers_raw(31, 15, 42, B.FLAG_CANONICAL | B.FLAG_SYNC, first=half, n=n - half)
    rw = b.super_kmers_raw(31, 15, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
    assert r1.count + r2.count == rw.count and (r1.xor_hash ^ r2.xor_hash) == rw.xor_hash and r1.aux + r2.aux == rw.aux == rw.count
    g = b.super_kmers(31, 15, seed=42, canonical=True, first=0, n=100 * L)
    assert int(g["sizes"].sum(dtype=np.uint64)) == 100 * (L - 30)
    m1 = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=half)
    m2 = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC, first=half, n=n - half)
    mw = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
    assert m1.count + m2.count == mw.count and (m1.xor_pos ^ m2.xor_pos) == mw.xor_pos
    # density of random minimizers is 2/(w+1) per window
    windows = (n // L) * (L - 41 + 1)
    assert abs(mw.count / windows - 2 / 12) < 0.002
    s1 = b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=half).count
    s2 = b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=half, n=n - half).count
    assert s1 + s2 == b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL | B.FLAG_SYNC).count
    k1 = b.kmers_raw(31, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=half)
    k2 = b.kmers_raw(31, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=half, n=n - half)
    kw = b.kmers_raw(31, 0, B.FLAG_CANONICAL | B.FLAG_SYNC)
    assert k1.count + k2.count == kw.count == (n // L) * (L - 30) and (k1.xor_hash ^ k2.xor_hash) == kw.xor_hash


def test_full_baseline_size_composition():
    """BASELINE C3 at its full size (333,333,333 reads x 150 bp = 50 Gbp on one GPU): the scan of the whole shard in
    34 ranges and in 26 differently cut ranges must agree on count and on all three XOR digests (range composition is
    exact on read boundaries), with two lanes in flight; density and k-mer totals match the closed forms."""
    import biolib_amd as B

    c = B.Context(0, torch_stream=False, lanes=2)
    L, n_reads = 150, 333_333_333
    n = L * n_reads
    b = c.synth(42, n, L)

    def scan(reads_per_range):
        res = []
        for a in range(0, n_reads, reads_per_range):
            m = min(reads_per_range, n_reads - a)
            res.append(b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL, first=a * L, n=m * L))
        c.sync()
        assert all(r.status == 0 for r in res)
        cnt = sum(int(r.count) for r in res)
        xv = xh = xp = 0
        for r in res:
            xv ^= int(r.xor_value); xh ^= int(r.xor_hash); xp ^= int(r.xor_pos)
        return cnt, xv, xh, xp

    a = scan(10_000_000)
    d = scan(13_000_001)  # 1.95e9 positions per range: just under the 2^31 limit
    assert a == d
    windows = n_reads * (L - 41 + 1)
    assert abs(a[0] / windows - 2 / 12) < 0.01          # random minimizers: 2/(w+1) per window
    k = b.kmers_raw(31, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=13_333_333 * L)
    assert k.count == 13_333_333 * (L - 30)
    c.close()


def test_bench_two_rank_flow_on_one_gpu():
    """bench.py under torch.distributed.run with two ranks (rehearsal mode: both on this GPU, gloo): one JSON line from
    rank 0, whole-job aggregate over both shards, digests reduced across ranks"""
    import json
    import os
    import subprocess
    import sys

    import biolib_amd as B

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BL_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gbp", "1.5"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and "rehearsal" in d and d["value"] > 0
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"
    # the same line WITHOUT a launcher: `python bench.py --gpus 2` starts its two ranks itself (what the driver's N > 1 call
    # looks like when it does not go through torch.distributed.run); n_gpus is what was asked for, digests as above
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gbp", "1.5"],
                          capture_output=True, text=True, timeout=900, env=env2, cwd=root)
    assert out2.returncode == 0, out2.stdout[-2000:] + out2.stderr[-3000:]
    lines2 = [x for x in out2.stdout.splitlines() if x.startswith("{")]
    assert len(lines2) == 1
    d2 = json.loads(lines2[0])
    assert d2["n_gpus"] == 2 and d2["rccl_ranks"] == 2 and d2["collective_backend"] == "gloo"
    assert (d2["records_per_step"], d2["xor_hash"]) == (d["records_per_step"], d["xor_hash"])
    # the reduced digest = the two shards (seed 42 and 43) scanned directly
    c = B.Context(0)
    n = d["config"]["bases_per_gpu"]
    cnt, xh = 0, 0
    for seed in (42, 43):
        b = c.synth(seed, n, 150)
        r = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
        cnt += int(r.count)
        xh ^= int(r.xor_hash)
        b.close()
    assert d["records_per_step"] == cnt and d["xor_hash"] == xh
    c.close()


# ------------------------------------------------------------------------------------------------------------------
# The other BASELINE.json configurations at the sizes they are stated at (one GPU).  Parity at that size comes from
# size-independent properties — two different cuttings of the shard must give the same count and digests — plus the
# CPU oracle on a 256-Mbp prefix.

def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _fold(results):
    cnt, xv, xh, xp, aux = 0, 0, 0, 0, 0
    for r in results:
        assert r.status == 0
        cnt += int(r.count); xv ^= int(r.xor_value); xh ^= int(r.xor_hash); xp ^= int(r.xor_pos); aux += int(r.aux)
    return cnt, xv, xh, xp, aux


def test_full_baseline_size_c2_kmers_10gbp():
    """BASELINE C2 as stated: canonical 31-mer 2-bit encode + hash64 over 10 Gbp of synthetic DNA on one GPU (one sequence).
    Two cuttings agree on count / XOR of values / XOR of hashes / wrapping sum of hashes; the count has its closed form;
    the first 256 Mbp match the CPU oracle's digest."""
    import biolib_amd as B

    c = B.Context(0, torch_stream=False, lanes=2)
    n = 10_000_000_000
    b = c.synth(42, n)

    def scan(step):
        res = [b.kmers_raw(31, 0, B.FLAG_CANONICAL, first=a, n=min(step, n - a)) for a in range(0, n, step)]
        c.sync()
        cnt, xv, xh = 0, 0, 0
        sm = 0
        for r in res:
            assert r.status == 0
            cnt += int(r.count); xv ^= int(r.xor_value); xh ^= int(r.xor_hash); sm = (sm + int(r.xor_pos)) & (2**64 - 1)
        return cnt, xv, xh, sm

    a = scan(1_500_000_000)
    d = scan(2_147_483_648)  # the largest range the ABI takes
    assert a == d and a[0] == n - 30
    s = 256_000_000
    o = O.kmer_digest(b.download(0, s + 30), np.array([0, s + 30], np.uint64), 31, True, 0, threads=_cores())
    r = b.kmers_raw(31, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=s)
    assert (int(r.count), int(r.xor_value), int(r.xor_hash)) == (o["count"], o["xor_value"], o["xor_hash"])
    c.close()


def test_full_baseline_size_c4_super_kmers_50gbp():
    """BASELINE C4 as stated: super_kmer_view k=31 m=15 over 50 Gbp of 10-kbp reads on one GPU, records materialised.
    Two cuttings agree; every group start has its end; group sizes sum to the k-mer total; 256 Mbp against the oracle."""
    import biolib_amd as B

    c = B.Context(0, torch_stream=False, lanes=2)
    L, n_reads = 10_000, 5_000_000
    n = L * n_reads
    b = c.synth(42, n, L)
    cap = int(2_147_480_000 * 2.3 / 18) + 65536
    bufs = [(c.empty_u64(cap), c.empty_u64(cap), c.empty_u8(cap), c.empty_u8(cap), c.empty_u64(cap)) for _ in range(2)]

    def scan(reads_per_range):
        res = []
        for i, a in enumerate(range(0, n_reads, reads_per_range)):
            m = min(reads_per_range, n_reads - a)
            mn, fp, mp, sz, hs = bufs[i & 1]
            res.append(b.super_kmers_raw(31, 15, 42, B.FLAG_CANONICAL, first=a * L, n=m * L, minimizers=mn, first_pos=fp, mm_pos=mp, sizes=sz,
                                         hashes=hs, capacity=cap))
        c.sync()
        return _fold(res)

    a = scan(150_000)
    d = scan(214_748)  # 2,147,480,000 positions per range: just under the 2^31 limit
    assert a == d and a[0] == a[4]                      # every start has its end
    windows = n_reads * (L - 31 + 1)
    assert abs(a[0] / windows - 2 / 18) < 0.005         # random minimizers: 2/(w+1) groups per k-mer, w = 17
    s = 25_600 * L
    g = b.super_kmers(31, 15, seed=42, canonical=True, first=0, n=s)
    assert int(g["sizes"].sum(dtype=np.uint64)) == 25_600 * (L - 30)
    seq = b.download(0, s)
    mn, fp, mp, sz, hs = O.super_kmers(seq, O.fixed_offsets(s, L), 31, 15, 42, True)
    assert g["count"] == len(mn) and np.array_equal(g["first_pos"], fp) and np.array_equal(g["sizes"], sz)
    assert np.array_equal(g["minimizers"], mn) and np.array_equal(g["mm_pos"], mp) and np.array_equal(g["hashes"], hs)
    c.close()


def test_full_baseline_size_c5_syncmers_with_rccl_count_reduce():
    """BASELINE C5, one GPU's share as stated: syncmer_sampler k=31 s=11 offsets {0, 20} over a 50-Gbp shard of 10-kbp reads,
    then the count reduction over RCCL (backend nccl, world 1 — the 8-GPU leg needs hardware this box does not have).
    Two cuttings of the shard agree on count and position digest; 256 Mbp match the oracle's count."""
    import torch
    import torch.distributed as dist

    import biolib_amd as B
    from biolib_amd.shard import reduce_digests

    c = B.Context(0, torch_stream=False, lanes=2)
    L, n_reads = 10_000, 5_000_000
    n = L * n_reads
    b = c.synth(42, n, L)

    def scan(reads_per_range):
        res = [b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL, first=a * L, n=min(reads_per_range, n_reads - a) * L)
               for a in range(0, n_reads, reads_per_range)]
        c.sync()
        return _fold(res)

    a = scan(150_000)
    d = scan(214_748)
    assert a == d
    kmers = n_reads * (L - 30)
    assert abs(a[0] / kmers - 2 / 21) < 0.003           # open syncmers at two offsets of 21: 2/(k-s+1) of the k-mers
    s = 25_600 * L
    cnt, _ = O.syncmers(b.download(0, s), O.fixed_offsets(s, L), 31, 11, 0, 20, True, threads=_cores(), positions=False)
    assert int(b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=s).count) == cnt
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        tot = reduce_digests(dict(count=a[0], xor_pos=a[3]), device="cuda")  # all-reduce (sum) + all-gather (xor fold) on the device
        assert tot == dict(count=a[0], xor_pos=a[3])
    finally:
        if created:
            dist.destroy_process_group()
    c.close()


def test_upload_reads_origin_and_markers(ctx):
    """round-3 entry points: bl_batch_upload_reads (reads of one length, no offsets) scans like the same reads uploaded with
    offsets and like the device generator's batch; bl_batch_set_origin shifts every reported position and the position digest,
    nothing else; bl_ctx_mark / bl_ctx_mark_times give non-decreasing device times, one per marker"""
    import biolib_amd as B

    L, n_reads = 150, 20_001
    n = L * n_reads - 37  # a shorter last read is kept
    seq = O.synth(42, n)
    a = ctx.upload(seq, read_len=L)
    b = ctx.upload(seq, np.minimum(np.arange(0, n + L, L, dtype=np.uint64), n))
    assert a.n_seqs == b.n_seqs == n_reads and a.n_bases == n
    ra, rb = a.minimizers(31, 11, seed=42, canonical=True), b.minimizers(31, 11, seed=42, canonical=True)
    v, p, h = O.minimizers(seq, np.minimum(np.arange(0, n + L, L, dtype=np.uint64), n), 31, 11, 42, True, brute=False)
    for r in (ra, rb):
        assert r["count"] == len(v) and np.array_equal(r["values"], v) and np.array_equal(r["positions"], p) and np.array_equal(r["hashes"], h)
    one = ctx.upload(seq[:1000], read_len=5000)  # a read length beyond the batch: one sequence
    assert one.n_seqs == 1
    one.close()
    # origin: positions and xor_pos move, values / hashes / counts do not
    origin = (1 << 40) + 12345
    a.set_origin(origin)
    for scan, key in ((lambda: a.minimizers(31, 11, seed=42, canonical=True), "positions"), (lambda: a.super_kmers(31, 15, seed=42, canonical=True), "first_pos"),
                      (lambda: a.syncmers(31, 11, 0, 20, canonical=True), "positions")):
        moved = scan()
        a.set_origin(0)
        plain = scan()
        a.set_origin(origin)
        assert moved["count"] == plain["count"] and np.array_equal(moved[key], plain[key] + np.uint64(origin))
    r0 = a.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
    assert int(r0.xor_pos) == O.xor_reduce(p + np.uint64(origin)) and int(r0.xor_hash) == O.xor_reduce(h)
    a.close(); b.close()
    # markers
    c = B.Context(0, torch_stream=False, lanes=2)
    big = c.synth(1, 150 * 2_000_000, 150)
    c.mark()
    for _ in range(3):
        big.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL)
        c.mark()
    t = c.mark_times()
    assert len(t) == 4 and t[0] >= 0 and all(y >= x for x, y in zip(t[:-1], t[1:])) and t[-1] > 0
    assert c.mark_times() == []  # forgotten
    big.close(); c.close()


def test_approximate_pass1_soak_ten_batches():
    """the decisions pass 1 makes on an approximation of the hash's high dword (murmur64_top) and the closed-syncmer form, against the
    exact forms, on 10 batches of 1.5 Gbp per scan kind (tests/perf/approx_soak.py): every digest identical, and the shipped form did
    meet — and hand to the exact kernels — tiles it could not decide.  The per-batch lines are kept (gpurun_out/r4/approx_soak.jsonl,
    copied to profiles/ when a round's numbers are committed)."""
    import json
    import os
    import sys

    import biolib_amd as B

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "perf"))
    import approx_soak

    c = B.Context(0, torch_stream=False)
    lines, bad, redone = approx_soak.soak(c, 10, 1_500_000_000)
    c.close()
    out = os.path.join(root, "gpurun_out", "r4")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "approx_soak.jsonl"), "w") as f:
        for d in lines:
            f.write(json.dumps(d) + "\n")
        f.write(json.dumps({"batches": 10, "bases_per_batch": 1_500_000_000, "mismatches": bad, "tiles_decided_again": redone}) + "\n")
    assert bad == 0 and len(lines) == 20
    assert redone > 0
    assert all(d["count"] > 0 and d["status"] == 0 for d in lines)


def test_context_options_and_marker_limits():
    """round-4 ABI: bl_ctx_set_option (names the scan path used to read from the environment) and the bounds of the marker calls"""
    import ctypes as C

    import biolib_amd as B

    c = B.Context(0, torch_stream=False)
    lib = c._lib
    L, n = 150, 150 * 30_000
    b = c.synth(9, n, L)
    ref = b.minimizers(31, 11, seed=42, canonical=True)
    assert ref["count"] > 0
    # every switch leaves the records alone
    for name, value in (("position_tiled", 1), ("exact_windows", 1), ("lanes", 2), ("emit_lds_bytes", 40960), ("emit_lds_bytes", 0), ("lanes", 1),
                        ("exact_windows", 0), ("position_tiled", 0)):
        c.set_option(name, value)
        got = b.minimizers(31, 11, seed=42, canonical=True)
        assert got["count"] == ref["count"] and np.array_equal(got["positions"], ref["positions"]) and np.array_equal(got["hashes"], ref["hashes"]), (name, value)
    # unknown names and values out of range are refused, with a message
    for name, value in (("no_such_switch", 1), ("lanes", 3), ("exact_windows", 2), ("emit_lds_bytes", -1), ("emit_lds_bytes", 1 << 20)):
        with pytest.raises(B.BiolibError) as e:
            c.set_option(name, value)
        assert e.value.code == -1  # BL_ERR_INVALID
    # markers: a size query keeps them, a buffer too small is refused and keeps them, the 4,097th is refused
    for _ in range(5):
        c.mark()
    k = C.c_uint32()
    assert lib.bl_ctx_mark_times(c._h, None, 0, C.byref(k)) == 0 and k.value == 5
    small = (C.c_double * 2)()
    assert lib.bl_ctx_mark_times(c._h, small, 2, C.byref(k)) == -4 and k.value == 5  # BL_ERR_CAPACITY
    assert len(c.mark_times()) == 5 and c.mark_times() == []
    for _ in range(4096):
        c.mark()
    assert lib.bl_ctx_mark(c._h) == -4
    assert len(c.mark_times()) == 4096
    b.close(); c.close()


@pytest.mark.parametrize("case", ["reads151_minimizers", "reads10k_minimizers", "closed_syncmers_25_12", "super_kmers_31_9"])
def test_round4_kernels_at_size(case):
    """the kernels round 4 added, at sizes the oracle cannot follow whole: 20 Gbp each, scanned in two different cuttings (which must
    agree on count and on every digest: ranges compose exactly), once more with bl_ctx_set_exact_windows where the default decides on an
    approximation (same digests), a closed-form density, and the first 64 Mbp against the oracle record for record.
      reads151_minimizers     (31, 11) on 151-bp reads: read-tiled murmur64_top kernel with 16 units per lane
      reads10k_minimizers     (31, 11) on 10-kbp reads: position-tiled murmur64_top kernel + scan_redo_kernel
      closed_syncmers_25_12   closed syncmers with the width at run time, ties decided per k-mer in the kernel
      super_kmers_31_9        w = 23: one of the per-width kernels of the super-k-mer scan"""
    import biolib_amd as B

    c = B.Context(0, torch_stream=False, lanes=2)
    L = 151 if case == "reads151_minimizers" else 10_000
    n_reads = 20_000_000_000 // L
    n = n_reads * L
    b = c.synth(4242, n, L)
    cap = 420_000_000
    bufs = [tuple(c.empty_u64(cap) for _ in range(3)) for _ in range(2)]
    u8s = [tuple(c.empty_u8(cap) for _ in range(2)) for _ in range(2)] if case == "super_kmers_31_9" else None
    flags = B.FLAG_CANONICAL

    def one(i, first, m):
        v, p, h = bufs[i & 1]
        if case.endswith("minimizers"):
            return b.minimizers_raw(31, 11, 42, flags, first=first, n=m, values=v, positions=p, hashes=h, capacity=cap)
        if case == "closed_syncmers_25_12":
            return b.syncmers_raw(25, 12, 0, 13, 0, flags, first=first, n=m, positions=p, capacity=cap)
        mp, sz = u8s[i & 1]
        return b.super_kmers_raw(31, 9, 42, flags, first=first, n=m, minimizers=v, first_pos=p, mm_pos=mp, sizes=sz, hashes=h, capacity=cap)

    def scan(reads_per_range):
        res = [one(i, a * L, min(reads_per_range, n_reads - a) * L) for i, a in enumerate(range(0, n_reads, reads_per_range))]
        c.sync()
        return _fold(res)

    per = 1_900_000_000 // L
    a = scan(per)
    d = scan(per * 2 // 3 + 1)
    assert a == d, (a, d)
    if case != "super_kmers_31_9":  # (that kernel decides on the hashes themselves already)
        c.set_exact_windows(True)
        e = scan(per)
        c.set_exact_windows(False)
        assert e[:4] == a[:4]
    if case.endswith("minimizers"):
        assert abs(a[0] / (n_reads * (L - 41 + 1)) - 2 / 12) < 0.01
    elif case == "closed_syncmers_25_12":
        assert abs(a[0] / (n_reads * (L - 25 + 1)) - 2 / 14) < 0.01   # closed syncmers: 2 / (k - s + 1) of the k-mers
    else:
        assert a[0] == a[4] and abs(a[0] / (n_reads * (L - 31 + 1)) - 2 / 24) < 0.005
    s = 64_000_000 // L * L
    seq = b.download(0, s)
    offs = O.fixed_offsets(s, L)
    if case.endswith("minimizers"):
        ev, ep, eh = O.minimizers(seq, offs, 31, 11, 42, True, brute=False)
        g = b.minimizers(31, 11, seed=42, canonical=True, first=0, n=s)
        assert g["count"] == len(ev) and np.array_equal(g["positions"], ep) and np.array_equal(g["hashes"], eh) and np.array_equal(g["values"], ev)
    elif case == "closed_syncmers_25_12":
        cnt, pos = O.syncmers(seq, offs, 25, 12, 0, 13, True, threads=_cores())
        g = b.syncmers(25, 12, 0, 13, canonical=True, first=0, n=s)
        assert g["count"] == cnt and np.array_equal(g["positions"], pos)
    else:
        mn, fp, mp, sz, hs = O.super_kmers(seq, offs, 31, 9, 42, True)
        g = b.super_kmers(31, 9, seed=42, canonical=True, first=0, n=s)
        assert g["count"] == len(mn) and np.array_equal(g["first_pos"], fp) and np.array_equal(g["sizes"], sz) and np.array_equal(g["mm_pos"], mp) and np.array_equal(g["hashes"], hs)
    b.close()
    c.close()
