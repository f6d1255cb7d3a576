#!/usr/bin/env python3
"""The k-mer counter's chain on reads of sequencing COVERAGE — every read `copies` times over, shuffled ("coverage"), and reads drawn
at random offsets of a genome `copies` times its length ("sampled_coverage") — beside the same amount of reads without repeats.
One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import biolib_amd as B
import oracle_lib as O

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 30
k, m, L = 31, 15, 150
ctx = B.Context(0)
n_reads = int(gbp * 1e9) // L
out = {"bases": n_reads * L, "k": k, "m": m, "copies": copies}
for name, distinct in (("no_repeats", n_reads), ("coverage", n_reads // copies), ("sampled_coverage", n_reads // copies)):
    if name == "sampled_coverage":  # reads drawn at random offsets of a genome (what a sequencer does): overlapping, not identical, reads
        genome = O.synth(9, distinct * L)
        pos = np.random.default_rng(6).integers(0, genome.size - L, n_reads)
        seq = np.ascontiguousarray(genome[pos[:, None] + np.arange(L)[None, :]]).reshape(-1)
    else:
        base = O.synth(7, distinct * L).reshape(distinct, L)
        order = np.random.default_rng(5).permutation(n_reads) % distinct
        seq = np.ascontiguousarray(base[order]).reshape(-1)
    b = ctx.upload(seq, O.fixed_offsets(seq.size, L))
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        recs, hashes = b.super_kmer_records(k, m, seed=42, canonical=True)
        u, c = ctx.count_super_kmers(recs, k, m, seed=42, canonical=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out[name] = {"Gbp_s": round(seq.size / best / 1e9, 1), "ms": round(best * 1e3, 1), "distinct_kmers": int(u.numel()), "kmers": int(c.sum())}
    b.close()
    del recs, hashes, u, c
print(json.dumps(out))
