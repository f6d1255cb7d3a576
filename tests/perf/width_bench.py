#!/usr/bin/env python3
"""Throughput of the minimizer and syncmer scans over window sizes (one lane, synchronous calls): the tuned template
widths against the sparse-table kernels that serve every other width.  Numbers quoted in DESIGN.md §5."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import biolib_amd as B

ctx = B.Context(0, torch_stream=False)
n = 1_500_000_000
out = {"minimizers": {}, "syncmers": {}}
b = ctx.synth(42, n, 150)
for (unit, w) in ((31, 11), (15, 10), (19, 19), (25, 5), (15, 17), (21, 21), (31, 2), (21, 3), (21, 4), (25, 7), (25, 8), (15, 9), (15, 13),
                  (15, 16), (15, 20), (15, 25), (15, 32), (15, 33), (15, 48), (31, 64)):
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        b.minimizers_raw(unit, w, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
        dt = time.perf_counter() - t0
    out["minimizers"][f"unit{unit}_w{w}"] = round(n / dt / 1e9, 1)
b.close()
b = ctx.synth(42, n, 10000)
for (k, s) in ((31, 11), (31, 15), (21, 11), (31, 8), (25, 12), (20, 16)):
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        b.syncmers_raw(k, s, 0, k - s, 0, B.FLAG_CANONICAL | B.FLAG_SYNC)
        dt = time.perf_counter() - t0
    out["syncmers"][f"k{k}_s{s}_w{k - s + 1}"] = round(n / dt / 1e9, 1)
print(json.dumps(out))
