#!/usr/bin/env python3
"""Reads/s of the reference driver's per-read loop (tests/test_kmer_view.cpp:30-42: one kmer_view per read, every k-mer used)
over a FASTQ of 150-bp reads: the drop-in headers with a read_pool (one GPU scan per batch of reads), the drop-in view by view
(one GPU round trip per read), and the REFERENCE itself on a host core (oracle/_ref).  Writes one JSON line."""
import ctypes as C, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
k, canon, L = 31, 1, 150
exe = os.path.join(ROOT, "tests", "cpp", "_build", "bench_view_loop")
seq = O.synth(42, n_reads * L).reshape(n_reads, L)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "reads.fq")
    with open(path, "wb") as f:
        qual = b"I" * L
        for i in range(n_reads):
            f.write(b"@r%d\n" % i + seq[i].tobytes() + b"\n+\n" + qual + b"\n")
    out = subprocess.run([exe, path, str(k), str(canon), "3000"], capture_output=True, text=True, check=True)
    res = json.loads(out.stdout.strip().splitlines()[-1])
    R = O.ref()
    R.ref_read_loop_kmers_xor.restype = C.c_uint64
    R.ref_read_loop_kmers_xor.argtypes = [C.c_char_p, C.c_uint8, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    nr, nk = C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter()
    x = R.ref_read_loop_kmers_xor(path.encode(), k, canon, C.byref(nr), C.byref(nk))
    dt = time.perf_counter() - t0
    # the minimizer driver's loop (tests/test_minimizer_view.cpp:37-43: k = 15, m = 10, seed 42) through the pool; the reference's own
    # minimizer_view yields nothing, so there is no reference figure beside it
    mm = json.loads(subprocess.run([exe, path, "15", "0", "2000", "10", "42"], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    res["minimizer_loop"] = {key: mm[key] for key in ("reads", "minimizers", "batch_scans", "pooled_seconds", "pooled_reads_per_s", "per_view_reads", "per_view_reads_per_s")}
    res["minimizer_loop"]["pooled_equals_per_view_on_head"] = mm["head_pooled"] == mm["head_per_view"]
    sk = json.loads(subprocess.run([exe, path, "31", "0", "2000", "13", "0", "super"], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    res["super_kmer_loop"] = {key: sk[key] for key in ("reads", "batch_scans", "pooled_seconds", "pooled_reads_per_s", "per_view_reads", "per_view_reads_per_s")}
    res["super_kmer_loop"]["super_kmers"] = sk["minimizers"]
    res["super_kmer_loop"]["pooled_equals_per_view_on_head"] = sk["head_pooled"] == sk["head_per_view"]
    res.update(reference_reads_per_s=round(nr.value / dt), reference_seconds=round(dt, 3), reference_cores=1,
               same_kmers_as_reference=bool(x == res["xor_values"] and nk.value == res["kmers"] and nr.value == res["reads"]),
               workload=f"{n_reads} reads x {L} bp FASTQ, kmer_view k={k} canonical, `it != cend()` idiom")
print(json.dumps(res))
