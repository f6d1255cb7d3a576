#!/usr/bin/env python3
"""Digests of the headline minimizer scan and of the closed-syncmer scan over many synthetic batches, one JSON line each.
Run twice on the GPU box — as is, and with BL_NO_APPROX=1 BL_NO_CLOSED=1 (pass 1 on the hashes themselves, syncmers in the
argmin form) — and compare the two outputs: the approximate pass 1 (murmur64_top, bl_scan_core.hpp) must not change one record.
    python tests/perf/approx_soak.py N_BATCHES [GBP_PER_BATCH] > a.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import biolib_amd as B

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(float(sys.argv[2]) * 1e9) if len(sys.argv) > 2 else 1_500_000_000
ctx = B.Context(0, torch_stream=False)
cap = n // 6
v, p, h = ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u64(cap)
for i in range(n_batches):
    b = ctx.synth(1000 + i, n, 150)
    r = b.minimizers_raw(31, 11, 42 + i, B.FLAG_CANONICAL | B.FLAG_SYNC, values=v, positions=p, hashes=h, capacity=cap)
    d = r.as_dict(); d.update(scan="minimizers", batch=i)
    print(json.dumps(d), flush=True)
    b.close()
    b = ctx.synth(5000 + i, n, 10000)
    r = b.syncmers_raw(31, 11, 0, 20, i, B.FLAG_CANONICAL | B.FLAG_SYNC, positions=p, capacity=cap)
    d = r.as_dict(); d.update(scan="closed_syncmers", batch=i)
    print(json.dumps(d), flush=True)
    b.close()
