#!/usr/bin/env python3
"""Scan rates off the BASELINE shapes: read lengths x window widths x syncmer shapes, ragged reads included,
one lane, one JSON object.  Every case scans the same number of bases (default 3 Gbp in ranges of <= 1.5 Gbp, records materialised)
and reports Gbp/s from the wall time of three passes; the first 3 Mbp of every case are checked against the CPU oracle.
    python tests/perf/shape_sweep.py [GBP] > gpurun_out/r4/shape_sweep.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import biolib_amd as B
import oracle_lib as O

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
only = sys.argv[2] if len(sys.argv) > 2 else ""
ctx = B.Context(0, torch_stream=False, lanes=1)
CH = 1_500_000_000
N = int(gbp * 1e9)
cap = CH // 4 + 65536
v, p, h = ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u64(cap)


def batch_for(L):
    """L > 0: reads of that length; 0: one sequence"""
    if L > 0:
        n = N // L * L
        return ctx.synth(42, n, L), n, O.fixed_offsets, L
    if L == 0:
        return ctx.synth(42, N), N, None, 0
    raise ValueError(L)


def timed(fn, n, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return round(n * reps / (time.perf_counter() - t0) / 1e9, 1)


def ranges(n, L):
    step = CH // L * L if L > 0 else CH
    return [(a, min(step, n - a)) for a in range(0, n, step)]


out = {"gbp_per_case": gbp, "lanes": 1, "minimizers": {}, "syncmers": {}, "super_kmers": {}, "parity_failures": []}
layouts = [150, 151, 100, 101, 125, 250, 300, 10000, 0]
mm_shapes = [(31, 11), (15, 10), (19, 19), (25, 5), (15, 17), (21, 21), (31, 2), (25, 8), (15, 13), (15, 16), (15, 22), (15, 25), (15, 32), (15, 33), (15, 48), (31, 64)]
sy_shapes = [(31, 11, 0, 20), (31, 15, 0, 16), (21, 11, 0, 10), (31, 8, 0, 23), (25, 12, 0, 13), (20, 16, 0, 4), (15, 5, 0, 10), (31, 11, 3, 9), (21, 8, 2, 5)]
for L in layouts:
    tag = f"reads{L}" if L > 0 else "one_sequence"
    if only and only != tag:
        continue
    b, n, offs_fn, _ = batch_for(L)
    sample_n = (3_000_000 // L * L) if L > 0 else 3_000_000
    seq = b.download(0, sample_n)
    offs = O.fixed_offsets(sample_n, L) if L > 0 else np.array([0, sample_n], np.uint64)
    for (unit, w) in (mm_shapes if L in (150, 10000, 0) else mm_shapes[:1]):
        def run():
            for a, m in ranges(n, L):
                b.minimizers_raw(unit, w, 42, B.FLAG_CANONICAL, first=a, n=m, values=v, positions=p, hashes=h, capacity=cap)
        out["minimizers"][f"{tag}_unit{unit}_w{w}"] = timed(run, n)
        if L > 0:  # parity on the sample (a range that ends where a read ends)
            ev, ep, eh = O.minimizers(seq, offs, unit, w, 42, True, brute=False)
            g = b.minimizers(unit, w, seed=42, canonical=True, first=0, n=sample_n)
            if not (g["count"] == len(ev) and np.array_equal(g["positions"], ep) and np.array_equal(g["hashes"], eh)):
                out["parity_failures"].append(f"mm {tag} {unit} {w}")
    if L in (10000, 150, 0):
        for (k, s, a0, e0) in sy_shapes:
            def run():
                for a, m in ranges(n, L):
                    b.syncmers_raw(k, s, a0, e0, 0, B.FLAG_CANONICAL, first=a, n=m, positions=p, capacity=cap)
            out["syncmers"][f"{tag}_k{k}_s{s}_off{a0}_{e0}"] = timed(run, n)
            if L > 0:
                cnt, pos = O.syncmers(seq, offs, k, s, a0, e0, True, threads=8)
                g = b.syncmers(k, s, a0, e0, canonical=True, first=0, n=sample_n)
                if not (g["count"] == cnt and np.array_equal(g["positions"], pos)):
                    out["parity_failures"].append(f"sync {tag} {k} {s} {a0} {e0}")
    b.close()
    print(f"# {tag} done", file=sys.stderr, flush=True)

# ragged reads (trimmed: lengths 100..151), the BASELINE shape: the position-tiled kernels
if not only or only == "ragged":
    rng = np.random.default_rng(5)
    lens = rng.integers(100, 152, N // 126)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(offs[-1])
    host = O.synth(42, min(n, 600_000_000))  # (host generator: bounded; the batch is its first part repeated on upload)
    reps_needed = 1
    n_up = int(offs[np.searchsorted(offs, len(host), side="right") - 1])
    b = ctx.upload(host[:n_up], offs[: np.searchsorted(offs, n_up, side="right")])
    def run():
        for a0 in range(0, 1):
            b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL, values=v, positions=p, hashes=h, capacity=cap)
    out["minimizers"]["ragged100to151_unit31_w11"] = timed(run, n_up, reps=5)
    k = int(np.searchsorted(offs, 3_000_000, side="right") - 1)
    so = offs[: k + 1]
    ev, ep, eh = O.minimizers(host[: int(so[-1])], so, 31, 11, 42, True, brute=False)
    g = b.minimizers(31, 11, seed=42, canonical=True, first=0, n=int(so[-1]))
    if not (g["count"] == len(ev) and np.array_equal(g["positions"], ep)):
        out["parity_failures"].append("mm ragged 31 11")
    b.close()
# a scan cannot outrun the memory it writes: 24 bytes per record, 2 / (w + 1) records per window (+ 1 byte read, ~1 of scratch per base);
# 4.5 TB/s is what the record pass reaches on its own: cases whose output alone needs that are output-bound, not slow
def output_bound(key, rate):
    w = int(key.rsplit("_w", 1)[1])
    bytes_per_base = 24 * 2.0 / (w + 1) + 2.0
    return {"Gbps": rate, "record_bytes_per_base": round(bytes_per_base, 2), "hbm_GBps_needed": round(rate * bytes_per_base, 0), "ceiling_Gbps_at_4500GBps": round(4500 / bytes_per_base, 0)}
out["minimizers_output_bound"] = {k: output_bound(k, x) for k, x in out["minimizers"].items() if x < 300 and int(k.rsplit("_w", 1)[1]) <= 8}
out["below_300_minimizers_w_le_32"] = sorted(k for k, x in out["minimizers"].items() if x < 300 and 8 < int(k.rsplit("_w", 1)[1]) <= 32)
out["below_250_syncmers"] = sorted(k for k, x in out["syncmers"].items() if x < 250)
print(json.dumps(out))
