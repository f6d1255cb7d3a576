#!/usr/bin/env python3
"""Throughput of the FASTA/FASTQ ingest (host parse -> device batch) and of the scan on the ingested batches.
Writes a synthetic FASTQ of 150-bp reads (plain and gzip), then times  reader -> batches -> minimizer scan."""
import gzip, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import biolib_amd as B
import oracle_lib as O

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
seq = O.synth(42, n_reads * 150).reshape(n_reads, 150)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "reads.fq")
    with open(path, "wb") as f:
        qual = b"I" * 150
        for i in range(n_reads):
            f.write(b"@r%d\n" % i + seq[i].tobytes() + b"\n+\n" + qual + b"\n")
    gz = path + ".gz"
    with open(path, "rb") as fi, gzip.open(gz, "wb", compresslevel=1) as fo:
        fo.write(fi.read())
    ctx = B.Context(0)
    for p in (path, gz):
        t0 = time.perf_counter(); nb = 0; cnt = 0
        for batch, names, offs in B.Reader(p).batches(ctx, 256_000_000):
            nb += batch.n_bases
            cnt += batch.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC).count
        dt = time.perf_counter() - t0
        print(f"{os.path.basename(p)}: {nb/1e6:.0f} Mbp, file {os.path.getsize(p)/1e6:.0f} MB, {nb/dt/1e6:.1f} Mbp/s end to end (host parse+upload+scan), {cnt} minimizers")
    # device-side parser: the raw text goes to the GPU and is parsed there (bl_batch_from_text)
    raw = np.fromfile(path, dtype=np.uint8)
    for rep in range(3):
        t0 = time.perf_counter()
        batch = ctx.from_text(raw)
        t1 = time.perf_counter()
        cnt = batch.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC).count
        t2 = time.perf_counter()
        print(f"device parser rep {rep}: {batch.n_bases/1e6:.0f} Mbp from {raw.size/1e6:.0f} MB of FASTQ text: parse+upload {batch.n_bases/(t1-t0)/1e9:.2f} Gbp/s "
              f"({raw.size/(t1-t0)/1e9:.2f} GB/s of text), scan {batch.n_bases/(t2-t1)/1e9:.1f} Gbp/s, {cnt} minimizers")
        batch.close()
