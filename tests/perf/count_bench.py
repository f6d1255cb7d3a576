#!/usr/bin/env python3
"""Stage timings of the partitioned k-mer counter on ONE GPU: scan -> pack -> bucket split by owner ->
(all-to-all skipped: world 1) -> count (minimizer buckets in LDS hash tables, bl_count_super_kmers), with the round-1 chain
(expand -> global radix sort -> run-length count) timed beside it.  Checked against the oracle on a sample."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import biolib_amd as B
import oracle_lib as O

gbp = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "--old" else 1.5
OLD = "--old" in sys.argv  # also time the round-1 chain (expand -> global sort -> run-length count); the check below uses it either way
k, m, L = 31, 15, 150
ctx = B.Context(0)
n = int(gbp * 1e9) // L * L
b = ctx.synth(42, n, L)
out = {"bases": n, "k": k, "m": m}

def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[name + "_ms"] = round(dt * 1e3, 2); out[name + "_Gbp_s"] = round(n / dt / 1e9, 1)
    return r

out_keys = ctx.empty_u64(int(n * 0.82))  # caller-owned outputs: ~0.8 k-mers per base
out_cnts = torch.empty(int(n * 0.82), dtype=torch.int32, device="cuda")
for rep in range(3):  # the last pass = warm numbers (the first also runs the round-1 chain once, for the check)
    recs, hashes = timed("scan_pack", lambda: b.super_kmer_records(k, m, seed=42, canonical=True))
    bucketed, counts = timed("bucket_split_8", lambda: ctx.partition_records(hashes, recs, 8))
    u, c = timed("count_buckets", lambda: ctx.count_super_kmers(bucketed, k, m, seed=42, canonical=True, out=(out_keys, out_cnts)))
    if OLD or rep == 0:
        kmers = timed("old_expand", lambda: ctx.expand_super_kmers(bucketed, k, canonical=True))
        u0, c0 = timed("old_sort_count", lambda: ctx.sort_count(kmers))
        n_kmers, n_u0, sum_c0 = int(kmers.numel()), int(u0.numel()), int(c0.sum())
        del kmers, u0, c0
out["super_kmers"] = int(recs.shape[0]); out["kmers"] = n_kmers; out["distinct"] = int(u.numel())
assert u.numel() == n_u0 and int(c.sum()) == sum_c0 == n_kmers
if not OLD:
    for key in [x for x in out if x.startswith("old_")]:
        del out[key]
out["record_bytes_per_base"] = round(16 * recs.shape[0] / n, 3)
out["whole_chain_Gbp_s"] = round(n / sum(out[s + "_ms"] for s in ("scan_pack", "bucket_split_8", "count_buckets")) * 1e-6, 1)
if OLD:
    out["old_chain_Gbp_s"] = round(n / sum(out[s + "_ms"] for s in ("scan_pack", "bucket_split_8", "old_expand", "old_sort_count")) * 1e-6, 1)
# parity on a sample: the multiset of canonical k-mers of the first 20,000 reads
s = 20_000 * L
sb = ctx.upload(b.download(0, s), O.fixed_offsets(s, L))
r2, _ = sb.super_kmer_records(k, m, seed=42, canonical=True)
u2, c2 = ctx.count_super_kmers(r2, k, m, seed=42, canonical=True)
order = np.argsort(u2.cpu().numpy().view(np.uint64))
vals, ok = O.units(b.download(0, s), O.fixed_offsets(s, L), k, True)
eu, ec = np.unique(vals[ok != 0], return_counts=True)
out["parity_sample"] = bool(np.array_equal(u2.cpu().numpy().view(np.uint64)[order], eu) and np.array_equal(c2.cpu().numpy().astype(np.int64)[order], ec))
print(json.dumps(out))
