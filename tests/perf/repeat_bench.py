#!/usr/bin/env python3
"""Sensitivity of the minimizer scan to hash ties: the same 300 Mbp of 150-bp reads, random and with tandem repeats planted
in a given fraction of the reads (identical k-mers inside a window send their wave through the exact 64-bit branch)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import biolib_amd as B
import oracle_lib as O

ctx = B.Context(0, torch_stream=False)
L, n_reads = 150, 2_000_000
n = L * n_reads
rng = np.random.default_rng(1)
base = O.synth(3, n)
offs = O.fixed_offsets(n, L)
out = {}
for frac in (0.0, 0.02, 0.10, 0.50):
    seq = base.copy()
    reads = rng.choice(n_reads, int(frac * n_reads), replace=False)
    for r in reads[: 200_000]:  # plant a tandem repeat (period 2..12, 40..90 bases) inside the read
        per, ln = int(rng.integers(2, 13)), int(rng.integers(40, 91))
        st = r * L + int(rng.integers(0, L - ln))
        seq[st:st + ln] = np.resize(seq[st:st + per], ln)
    if len(reads) > 200_000:  # beyond 200k reads reuse the pattern of the planted ones (keeps the host loop short)
        src = reads[:200_000]
        for i, r in enumerate(reads[200_000:]):
            s = src[i % 200_000]
            seq[r * L:(r + 1) * L] = seq[s * L:(s + 1) * L]
    b = ctx.upload(seq, offs)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        res = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
        dt = time.perf_counter() - t0
    d = O.minimizer_digest(seq, offs, 31, 11, 42, True, threads=16)
    out[f"repeat_reads_{frac}"] = {"Gbp_s": round(n / dt / 1e9, 1), "records": int(res.count),
                                   "bit_identical": (res.count, res.xor_hash, res.xor_pos) == (d["count"], d["xor_hash"], d["xor_pos"])}
    b.close()
print(json.dumps(out))
