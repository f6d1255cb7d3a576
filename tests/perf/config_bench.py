#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one GPU (they are parity-test cases, not the bench
line): C2 canonical 31-mer + hash64 digest, C4 super-k-mers k=31 m=15 on 10-kbp reads, C5 syncmers k=31 s=11 on
10-kbp reads.  Each result is checked against the CPU oracle on a sample."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import biolib_amd as B
import oracle_lib as O

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = B.Context(0, torch_stream=False, lanes=lanes)
CH = int(float(sys.argv[3]) * 1e9) if len(sys.argv) > 3 else 1_500_000_000  # bases per scan range
out = {"lanes": lanes, "range_Gbp": CH / 1e9}
if len(sys.argv) > 4:  # LDS footprint the record pass is padded to beside the next hashing pass (bl_ctx_set_option)
    ctx.set_option("emit_lds_bytes", int(sys.argv[4]))
    out["emit_lds_bytes"] = int(sys.argv[4])

def timed(fn, n_bases, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return n_bases * reps / (time.perf_counter() - t0) / 1e9

# C2: k-mers + hash digest over `gbp` Gbp, one sequence
n = int(gbp * 1e9)
b = ctx.synth(42, n)
def c2():
    for a in range(0, n, CH): b.kmers_raw(31, 0, B.FLAG_CANONICAL, first=a, n=min(CH, n - a))
out["C2_kmer_hash_digest_Gbps"] = round(timed(c2, n), 1)
s = 50_000_000
d = O.kmer_digest(b.download(0, s + 30), np.array([0, s + 30], np.uint64), 31, True, 0, threads=16)
r = b.kmers_raw(31, 0, B.FLAG_CANONICAL | B.FLAG_SYNC, first=0, n=s)
g = r.as_dict(); out["C2_parity_sample"] = (g["count"], g["xor_hash"]) == (d["count"], d["xor_hash"])
b.close()

# C4 / C5: 10-kbp reads
L = 10_000
n = int(gbp * 1e9) // L * L
b = ctx.synth(42, n, L)
chunk = CH // L * L
cap = int(chunk * 2.3 / 18) + 65536
bufs = [(ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u8(cap), ctx.empty_u8(cap), ctx.empty_u64(cap)) for _ in range(2)]
def c4():
    for i, a in enumerate(range(0, n, chunk)):
        mn, fp, mp, sz, hs = bufs[i & 1]
        b.super_kmers_raw(31, 15, 42, B.FLAG_CANONICAL, first=a, n=min(chunk, n - a), minimizers=mn, first_pos=fp, mm_pos=mp, sizes=sz, hashes=hs, capacity=cap)
out["C4_super_kmers_Gbps"] = round(timed(c4, n), 1)
cap5 = int(chunk * 2.6 / 21) + 65536
pbuf = [ctx.empty_u64(cap5) for _ in range(2)]
def c5():
    for i, a in enumerate(range(0, n, chunk)):
        b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL, first=a, n=min(chunk, n - a), positions=pbuf[i & 1], capacity=cap5)
out["C5_syncmers_Gbps"] = round(timed(c5, n), 1)
s = 2000 * L
seq = b.download(0, s); offs = O.fixed_offsets(s, L)
mn, fp, mp, sz, hs = O.super_kmers(seq, offs, 31, 15, 42, True)
g = b.super_kmers(31, 15, seed=42, canonical=True, first=0, n=s)
out["C4_parity_sample"] = bool(g["count"] == len(mn) and np.array_equal(g["first_pos"], fp) and np.array_equal(g["sizes"], sz) and np.array_equal(g["minimizers"], mn))
cnt, pos = O.syncmers(seq, offs, 31, 11, 0, 20, True, threads=16)
g = b.syncmers(31, 11, 0, 20, canonical=True, first=0, n=s)
out["C5_parity_sample"] = bool(g["count"] == cnt)
print(json.dumps(out))
