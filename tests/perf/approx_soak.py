#!/usr/bin/env python3
"""The approximate pass 1 (murmur64_top, bl_scan_core.hpp) and the closed-syncmer form must not change one record: digests (count,
XOR of values, hashes, positions) of the headline minimizer scan and of the closed-syncmer scan over many synthetic batches, each
batch scanned twice on one context — as shipped, and with bl_ctx_set_exact_windows (pass 1 on the hashes themselves, syncmers in the
argmin form).  tests/test_gpu_edges.py runs soak() inside the -m gpu suite; as a script it prints one JSON line per batch and scan:
    python tests/perf/approx_soak.py N_BATCHES [GBP_PER_BATCH] > gpurun_out/approx_soak.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def soak(ctx, n_batches, n, emit=None):
    """returns (lines, mismatches, tiles decided again by the shipped form)"""
    import biolib_amd as B

    cap = n // 6
    v, p, h = ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u64(cap)
    lines, bad, redone = [], 0, 0
    for i in range(n_batches):
        for scan, L in (("minimizers", 150), ("closed_syncmers", 10000)):
            b = ctx.synth((1000 if L == 150 else 5000) + i, n // L * L, L)
            got = []
            for exact in (False, True):
                ctx.set_exact_windows(exact)
                if scan == "minimizers":
                    r = b.minimizers_raw(31, 11, 42 + i, B.FLAG_CANONICAL | B.FLAG_SYNC, values=v, positions=p, hashes=h, capacity=cap)
                else:
                    r = b.syncmers_raw(31, 11, 0, 20, i, B.FLAG_CANONICAL | B.FLAG_SYNC, positions=p, capacity=cap)
                assert r.status == 0
                got.append(r.as_dict())
                if not exact:
                    redone += int(r.redone)
            ctx.set_exact_windows(False)
            same = got[0] == got[1]
            bad += 0 if same else 1
            d = dict(got[0], scan=scan, batch=i, bases=n // L * L, same_as_exact=same)
            lines.append(d)
            if emit:
                emit(d)
            b.close()
    return lines, bad, redone


if __name__ == "__main__":
    import biolib_amd as B

    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(float(sys.argv[2]) * 1e9) if len(sys.argv) > 2 else 1_500_000_000
    ctx = B.Context(0, torch_stream=False)
    _, bad, redone = soak(ctx, n_batches, n, emit=lambda d: print(json.dumps(d), flush=True))
    print(json.dumps({"batches": n_batches, "mismatches": bad, "tiles_decided_again": redone}), flush=True)
    sys.exit(1 if bad else 0)
