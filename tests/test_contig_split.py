"""The contig split of SURVEY.md §8(e): ONE concatenation cut across ranks by bases, (unit-1)+(w-1) bases of halo, windows owned
by the rank that holds their first base, positions global (biolib_amd/shard.py).  CPU tests: the host logic over an oracle-backed
stand-in for the batch (tests/shard_oracle_ctx.py) — in-process for many world sizes, and world 2 over gloo; the GPU test runs
the same functions on real batches, two ranks on one GPU, against the single-GPU scan and the oracle."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import oracle_lib as O
from biolib_amd import shard
from shard_oracle_ctx import OracleContext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def contig_with_breaks(seed, n, n_breaks):
    rng = np.random.default_rng(seed)
    seq = O.synth(seed, n).copy()
    if n_breaks:
        seq[rng.integers(0, n, n_breaks)] = np.frombuffer(b"NnRY-", np.uint8)[rng.integers(0, 5, n_breaks)]
    return seq


def whole(seq, offs):
    v, p, h = O.minimizers(seq, offs, 31, 11, 42, True, brute=False)
    mn, fp, mp, sz, hs = O.super_kmers(seq, offs, 31, 15, 42, True)
    _, sy = O.syncmers(seq, offs, 31, 11, 0, 20, True)
    _, sd = O.syncmers(seq, offs, 31, 11, 0, 20, True, drop_last=True)
    return dict(v=v, p=p, h=h, mn=mn, fp=fp, mp=mp, sz=sz, hs=hs, sy=sy, sd=sd)


def by_ranks(ctx, seq, offs, world):
    parts = []
    for r in range(world):
        a = shard.minimizers_of_shard(ctx, seq, offs, 31, 11, seed=42, canonical=True, rank=r, world_size=world)
        b = shard.super_kmers_of_shard(ctx, seq, offs, 31, 15, seed=42, canonical=True, rank=r, world_size=world)
        c = shard.syncmers_of_shard(ctx, seq, offs, 31, 11, 0, 20, canonical=True, rank=r, world_size=world)
        d = shard.syncmers_of_shard(ctx, seq, offs, 31, 11, 0, 20, canonical=True, drop_last=True, rank=r, world_size=world)
        parts.append((a, b, c, d))
    cat = lambda i, k: np.concatenate([p[i][k] for p in parts])
    return dict(v=cat(0, "values"), p=cat(0, "positions"), h=cat(0, "hashes"), mn=cat(1, "minimizers"), fp=cat(1, "first_pos"), mp=cat(1, "mm_pos"),
                sz=cat(1, "sizes"), hs=cat(1, "hashes"), sy=cat(2, "positions"), sd=cat(3, "positions"))


def assert_same(got, exp, what):
    for k in exp:
        assert np.array_equal(got[k], exp[k]), (what, k, len(got[k]), len(exp[k]))


def test_base_ranges_partition_the_bases():
    for total in (0, 1, 7, 1000, 30_000_001):
        for world in (1, 2, 3, 8):
            r = [shard.base_range(total, world, g) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_piece_keeps_sequence_starts_and_halo():
    offs = np.array([0, 10, 500, 501, 2000, 2000, 3000], np.uint64)
    p = shard.contig_piece(offs, 400, 1900, 31, 11)
    assert (p["piece_lo"], p["piece_hi"], p["first"], p["n"]) == (399, 1900 + 40, 1, 1500)
    assert p["offsets"].tolist() == [0, 500 - 399, 501 - 399, 1940 - 399]
    p = shard.contig_piece(offs, 0, 3000, 31, 11, guard=17)
    assert (p["piece_lo"], p["piece_hi"], p["first"], p["n"]) == (0, 3000, 0, 3000)
    p = shard.contig_piece(offs, 1000, 1000, 31, 11)  # a rank that owns nothing
    assert p["n"] == 0


@pytest.mark.parametrize("layout", ["one_contig", "contig_among_reads", "short_reads"])
def test_ranks_in_process_equal_the_whole(layout):
    """every world size, cuts inside sequences, next to breaks and sequence ends, shards shorter than the halo"""
    ctx = OracleContext()
    if layout == "one_contig":
        seq = contig_with_breaks(5, 60_000, 40)
        offs = np.array([0, len(seq)], np.uint64)
    elif layout == "contig_among_reads":
        seq = contig_with_breaks(6, 50_000, 25)
        offs = np.array([0, 150, 300, 337, 40_000, 40_020, 40_051, 49_000, 50_000], np.uint64)
    else:
        seq = contig_with_breaks(7, 150 * 200, 10)
        offs = O.fixed_offsets(len(seq), 150)
    exp = whole(seq, offs)
    for world in (1, 2, 3, 5, 8, 64):
        assert_same(by_ranks(ctx, seq, offs, world), exp, (layout, world))
    assert_same(by_ranks(ctx, seq[:700], np.minimum(offs, 700)[: np.searchsorted(offs, 700) + 1] if layout != "one_contig" else np.array([0, 700], np.uint64), 40),
                whole(seq[:700], np.minimum(offs, 700)[: np.searchsorted(offs, 700) + 1] if layout != "one_contig" else np.array([0, 700], np.uint64)), (layout, "tiny shards"))


def test_scan_calls_of_a_long_piece_overlap(monkeypatch):
    """pieces beyond one scan call (1.5 G positions in production, 5,000 here): super-k-mers are not cut between the calls"""
    monkeypatch.setattr(shard, "_RANGE", 5_000)
    ctx = OracleContext()
    seq = contig_with_breaks(9, 42_000, 30)
    offs = np.array([0, 20_011, len(seq)], np.uint64)
    exp = whole(seq, offs)
    for world in (1, 2, 3):
        assert_same(by_ranks(ctx, seq, offs, world), exp, world)


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch.distributed as dist
    import oracle_lib as O
    from biolib_amd import shard
    from test_contig_split import contig_with_breaks
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = {n}
    seq = contig_with_breaks(11, n, n // 1500)   # every rank derives the same contig; it uploads only its piece
    offs = np.array([0, n], np.uint64)
    if {gpu}:
        import biolib_amd
        ctx = biolib_amd.Context(0)
    else:
        from shard_oracle_ctx import OracleContext
        ctx = OracleContext()
    a = shard.minimizers_of_shard(ctx, seq, offs, 31, 11, seed=42, canonical=True, rank=rank, world_size=world)
    b = shard.super_kmers_of_shard(ctx, seq, offs, 31, 15, seed=42, canonical=True, rank=rank, world_size=world)
    c = shard.syncmers_of_shard(ctx, seq, offs, 31, 11, 0, 20, canonical=True, rank=rank, world_size=world)
    mine = dict(v=a["values"], p=a["positions"], h=a["hashes"], mn=b["minimizers"], fp=b["first_pos"], mp=b["mm_pos"], sz=b["sizes"], hs=b["hashes"], sy=c["positions"])
    every = [None] * world
    dist.all_gather_object(every, mine)
    # the only collective of the path: the optional count reduction
    tot = shard.reduce_digests(dict(count=int(a["count"]), syncmer_count=int(c["count"]), xor_pos=O.xor_reduce(a["positions"])))
    if rank == 0:
        got = {{k: np.concatenate([e[k] for e in every]) for k in mine}}
        np.savez({out!r}, **got)
        print("RESULT " + json.dumps(tot))
    dist.destroy_process_group()
""")


def run_two_ranks(tmp_path, n, gpu, port):
    script = tmp_path / "worker.py"
    out_npz = str(tmp_path / "gathered.npz")
    script.write_text(WORKER.format(root=ROOT, n=n, gpu=gpu, out=out_npz))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("RESULT ")][-1]
    return dict(np.load(out_npz)), json.loads(line[len("RESULT "):])


def test_one_contig_on_two_ranks_over_gloo(tmp_path):
    """world 2 over gloo, oracle-backed batches: ONE contig with breaks, cut in the middle — the gathered records are the
    records of the whole, the count reduction gives the whole's counts"""
    n = 300_000
    got, tot = run_two_ranks(tmp_path, n, False, 29521)
    seq = contig_with_breaks(11, n, n // 1500)
    exp = whole(seq, np.array([0, n], np.uint64))
    exp.pop("sd")
    assert_same(got, exp, "gloo world 2")
    assert tot["count"] == len(exp["v"]) and tot["syncmer_count"] == len(exp["sy"]) and tot["xor_pos"] == O.xor_reduce(exp["p"])


@pytest.mark.gpu
def test_one_30mbp_contig_on_two_ranks_of_one_gpu(tmp_path):
    """two ranks (both on this GPU, gloo for the gather) split ONE 30-Mbp contig with breaks: the same ordered minimizer,
    super-k-mer and syncmer records as the single-GPU scan of the whole and as the oracle"""
    import biolib_amd

    n = 30_000_000
    got, tot = run_two_ranks(tmp_path, n, True, 29523)
    seq = contig_with_breaks(11, n, n // 1500)
    offs = np.array([0, n], np.uint64)
    exp = whole(seq, offs)
    exp.pop("sd")
    assert_same(got, exp, "two ranks on one GPU vs the oracle")
    ctx = biolib_amd.Context(0)
    b = ctx.upload(seq)
    a = b.minimizers(31, 11, seed=42, canonical=True)
    s = b.super_kmers(31, 15, seed=42, canonical=True)
    c = b.syncmers(31, 11, 0, 20, canonical=True)
    one = dict(v=a["values"], p=a["positions"], h=a["hashes"], mn=s["minimizers"], fp=s["first_pos"], mp=s["mm_pos"], sz=s["sizes"], hs=s["hashes"], sy=c["positions"])
    assert_same(got, one, "two ranks on one GPU vs the single-GPU scan")
    assert tot["count"] == a["count"] and tot["syncmer_count"] == c["count"] and tot["xor_pos"] == O.xor_reduce(a["positions"])
    b.close()
    ctx.close()


@pytest.mark.gpu
def test_ranks_in_process_on_real_batches():
    """the in-process form on the GPU: cuts inside sequences, beside breaks and sequence ends, several world sizes, drop_last"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    seq = contig_with_breaks(6, 500_000, 250)
    offs = np.array([0, 150, 300, 337, 400_000, 400_020, 400_051, 490_000, 500_000], np.uint64)
    exp = whole(seq, offs)
    for world in (1, 2, 3, 7):
        assert_same(by_ranks(ctx, seq, offs, world), exp, world)
    ctx.close()
