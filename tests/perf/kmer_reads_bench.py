#!/usr/bin/env python3
"""Dense k-mer scan (k=31 canonical, digest only) on one long sequence and on 150-bp reads: Gbp/s of each (one lane, synchronous)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import biolib_amd as B
ctx = B.Context(0, torch_stream=False)
n = 2_000_000_000
out = {}
for name, L in (("one_sequence", 0), ("reads_150", 150), ("reads_10000", 10000)):
    b = ctx.synth(42, n, L)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        r = b.kmers_raw(31, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
        dt = time.perf_counter() - t0
    out[name] = {"Gbps": round(n / dt / 1e9, 1), "count": int(r.count), "xor_hash": int(r.xor_hash)}
    b.close()
print(json.dumps(out))
