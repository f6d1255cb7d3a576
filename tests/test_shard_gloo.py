"""CPU-only, world_size 2 over gloo: the multi-GPU path of the scan — read sharding without a
data-path collective plus the final digest reduction (biolib_amd/shard.py).  Each rank scans its
shard with the CPU oracle (this is a test: no GPU here) and the reduced digest must equal the
whole-batch digest."""
import os
import subprocess
import sys
import textwrap

import numpy as np

import oracle_lib as O
from biolib_amd.shard import shard_ranges, shard_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_reads_partition():
    for n in (0, 1, 7, 8, 1000, 333_333_333):
        for w in (1, 2, 3, 8):
            parts = [shard_reads(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (a, c), (b, _) in zip(parts[:-1], parts[1:]):
                assert a + c == b
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_shard_ranges_ragged():
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 5000, 200)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for w in (1, 2, 5, 8):
        parts = [shard_ranges(offs, w, r) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == len(lens)
        for (_, e), (b, _) in zip(parts[:-1], parts[1:]):
            assert e == b


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch.distributed as dist
    import oracle_lib as O
    from biolib_amd.shard import shard_reads, reduce_digests
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L, n_reads = 150, 4001
    first, cnt = shard_reads(n_reads, world, rank)
    seq = O.synth(42, n_reads * L)[first * L:(first + cnt) * L]      # this rank's shard of the global input
    d = O.minimizer_digest(seq, O.fixed_offsets(cnt * L, L), 31, 11, 42, True)
    # positions are shard-relative on each rank: make them global before folding
    v, p, h = O.minimizers(seq, O.fixed_offsets(cnt * L, L), 31, 11, 42, True, brute=False)
    local = dict(count=d["count"], xor_value=d["xor_value"], xor_hash=d["xor_hash"], xor_pos=O.xor_reduce(p + np.uint64(first * L)))
    sy, _ = O.syncmers(seq, O.fixed_offsets(cnt * L, L), 31, 11, 0, 20, True, positions=False)
    local["syncmer_count"] = sy
    total = reduce_digests(local)
    if rank == 0:
        print("RESULT " + json.dumps(total))
    dist.destroy_process_group()
""")


def test_two_rank_scan_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    import json
    line = [x for x in out.stdout.splitlines() if x.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    L, n_reads = 150, 4001
    seq = O.synth(42, n_reads * L)
    offs = O.fixed_offsets(n_reads * L, L)
    d = O.minimizer_digest(seq, offs, 31, 11, 42, True)
    sy, _ = O.syncmers(seq, offs, 31, 11, 0, 20, True, positions=False)
    assert got["count"] == d["count"] and got["xor_value"] == d["xor_value"] and got["xor_hash"] == d["xor_hash"] and got["xor_pos"] == d["xor_pos"]
    assert got["syncmer_count"] == sy


def test_hash64_np_matches_oracle():
    L = O.oracle()
    v = np.random.default_rng(1).integers(0, 2**63, 500).astype(np.uint64)
    for seed in (0, 42, 2**32 + 5):
        assert np.array_equal(O.hash64_np(v, seed), np.array([L.blo_hash64_u64(int(x), seed) for x in v], np.uint64))


COUNT_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O
    from biolib_amd.shard import shard_reads, exchange_and_count
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L, n_reads = 100, 3001
    first, cnt = shard_reads(n_reads, world, rank)
    seq = O.synth(9, n_reads * L)[first * L:(first + cnt) * L]
    seq = np.concatenate([seq, seq[: 40 * L]])                      # repeats inside the shard -> multiplicities > 1
    vals, valid = O.units(seq, O.fixed_offsets(len(seq), L), 13, True)
    keys = torch.from_numpy(vals[valid != 0].view(np.int64).copy())

    def partition(t, parts):                                         # numpy stand-in of Context.partition
        k = t.numpy().view(np.uint64)
        owner = O.hash64_np(k, 0) % np.uint64(parts)
        order = np.argsort(owner, kind="stable")
        return torch.from_numpy(k[order].view(np.int64).copy()), np.bincount(owner.astype(np.int64), minlength=parts).tolist()

    def count(t):                                                    # numpy stand-in of Context.sort_count
        u, c = np.unique(t.numpy().view(np.uint64), return_counts=True)
        return u, c

    u, c = exchange_and_count(keys, partition, count)
    assert np.all(O.hash64_np(u, 0) % np.uint64(world) == rank)       # this rank only holds keys it owns
    np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), u=u, c=c)
    dist.destroy_process_group()
""")


def test_two_rank_exchange_and_count(tmp_path):
    script = tmp_path / "count_worker.py"
    script.write_text(COUNT_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    from biolib_amd.shard import shard_reads
    L, n_reads = 100, 3001
    whole = []
    for rank in range(2):
        first, cnt = shard_reads(n_reads, 2, rank)
        seq = O.synth(9, n_reads * L)[first * L:(first + cnt) * L]
        seq = np.concatenate([seq, seq[: 40 * L]])
        vals, valid = O.units(seq, O.fixed_offsets(len(seq), L), 13, True)
        whole.append(vals[valid != 0])
    u, c = np.unique(np.concatenate(whole), return_counts=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    gu = np.concatenate([p["u"] for p in parts])
    gc = np.concatenate([p["c"] for p in parts])
    order = np.argsort(gu)
    assert np.array_equal(gu[order], u) and np.array_equal(gc[order], c) and c.max() > 1


RECORD_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from biolib_amd.shard import exchange
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(100 + rank)
    n = 5000 + 777 * rank
    recs = rng.integers(0, 2**62, (n, 2)).astype(np.int64)
    recs[:, 1] = (recs[:, 1] & ~0xff) | rank                         # tag the source rank in the size byte
    owner = (recs[:, 0].view(np.uint64) % np.uint64(world)).astype(np.int64)
    order = np.argsort(owner, kind="stable")
    inbox = exchange(torch.from_numpy(recs[order].copy()), np.bincount(owner, minlength=world).tolist()).numpy()
    assert inbox.shape[1] == 2 and np.all(inbox[:, 0].view(np.uint64) % np.uint64(world) == rank)
    np.save(os.path.join({out!r}, f"inbox{{rank}}.npy"), inbox)
    np.save(os.path.join({out!r}, f"sent{{rank}}.npy"), recs)
    dist.destroy_process_group()
""")


def test_two_rank_record_exchange(tmp_path):
    script = tmp_path / "record_worker.py"
    script.write_text(RECORD_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29521", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    sent = np.concatenate([np.load(tmp_path / f"sent{r}.npy") for r in range(2)])
    got = np.concatenate([np.load(tmp_path / f"inbox{r}.npy") for r in range(2)])
    key = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]
    assert np.array_equal(key(sent), key(got))                       # every record arrived exactly once
    for r in range(2):                                               # grouped by source rank inside an inbox
        src = np.load(tmp_path / f"inbox{r}.npy")[:, 1] & 0xff
        assert np.all(np.diff(src) >= 0)
