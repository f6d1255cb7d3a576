#!/usr/bin/env python3
"""Where the time of the ingest paths goes: passes over a synthetic FASTQ as BGZF (inflated on the device), plain text and plain
gzip (decoded in parts by the host's cores) with BL_INGEST_TRACE=1 (the reader prints its own phase times), with and without the
scan behind each batch."""
import os, struct, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BL_INGEST_TRACE"] = "1"
import numpy as np
import biolib_amd as B
import oracle_lib as O


def bgzf(data, level=1, block=65280):
    out = bytearray()
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = z.compress(chunk) + z.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
span = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # the file holds the reads this many times over (BGZF members and FASTQ records concatenate)
seq = O.synth(42, n_reads * 150).reshape(n_reads, 150)
text = b"".join(b"@r%d\n" % i + seq[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(n_reads))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "r.bgzf.gz")
    packed = bgzf(text)
    eof_marker = 28
    open(path, "wb").write(packed[:-eof_marker] * repeat + packed[-eof_marker:])
    plain = os.path.join(d, "r.fq")
    with open(plain, "wb") as f:
        for _ in range(repeat):
            f.write(text)
    gz = os.path.join(d, "r.fq.gz")  # ONE deflate stream per member (zlib level 1): the host's many-threaded decoder, [pgzip] lines
    one = zlib.compressobj(1, zlib.DEFLATED, 31)
    one = one.compress(text) + one.flush()
    open(gz, "wb").write(one * repeat)
    ctx = B.Context(0)
    for what, scan in (("bgzf", False), ("bgzf", True), ("bgzf", True), ("plain", True), ("plain", True), ("gzip", True), ("gzip", True)):
        path = plain if what == "plain" else gz if what == "gzip" else os.path.join(d, "r.bgzf.gz")
        t0 = time.perf_counter(); nb = 0
        t_next = t_scan = t_close = 0.0
        r = B.Reader(path)
        it = iter(r.device_batches(ctx, span))
        while True:
            a = time.perf_counter()
            batch = next(it, None)
            b_ = time.perf_counter(); t_next += b_ - a
            if batch is None:
                break
            nb += batch.n_bases
            if scan:
                batch.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC)
            c_ = time.perf_counter(); t_scan += c_ - b_
            batch.close()
            t_close += time.perf_counter() - c_
        r.close()
        dt = time.perf_counter() - t0
        print(f"{what} scan={scan}: {nb / dt / 1e9:.2f} Gbp/s ({dt * 1e3:.0f} ms; next batch {t_next * 1e3:.0f}, scan {t_scan * 1e3:.0f}, close {t_close * 1e3:.0f} ms)", flush=True)
