#!/usr/bin/env python3
"""What the sequence-start bit vector costs the position-tiled kernels: C4 and C5 over 12 Gbp as 10-kbp reads (start bits
loaded) and as one sequence (none).  The difference is the most that arithmetic start flags for long fixed-length reads could gain."""
import os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import biolib_amd as B
ctx = B.Context(0, torch_stream=False, lanes=2)
CH = 1_500_000_000
n = 12_000_000_000
out = {}
for name, L in (("reads_10kbp", 10_000), ("one_sequence", 0)):
    nn = n // 10_000 * 10_000
    b = ctx.synth(42, nn, L) if L else ctx.synth(42, nn)
    chunk = CH // 10_000 * 10_000
    cap = int(chunk * 2.3 / 18) + 65536
    bufs = [(ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u8(cap), ctx.empty_u8(cap), ctx.empty_u64(cap)) for _ in range(2)]
    def c4():
        for i, a in enumerate(range(0, nn, chunk)):
            mn, fp, mp, sz, hs = bufs[i & 1]
            b.super_kmers_raw(31, 15, 42, B.FLAG_CANONICAL, first=a, n=min(chunk, nn - a), minimizers=mn, first_pos=fp, mm_pos=mp, sizes=sz, hashes=hs, capacity=cap)
    cap5 = int(chunk * 2.6 / 21) + 65536
    pbuf = [ctx.empty_u64(cap5) for _ in range(2)]
    def c5():
        for i, a in enumerate(range(0, nn, chunk)):
            b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL, first=a, n=min(chunk, nn - a), positions=pbuf[i & 1], capacity=cap5)
    for nm, fn in (("C4", c4), ("C5", c5)):
        fn(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(3): fn()
        ctx.sync(); out[f"{nm}_{name}"] = round(nn * 3 / (time.perf_counter() - t0) / 1e9, 1)
    b.close(); del bufs, pbuf
print(json.dumps(out))
