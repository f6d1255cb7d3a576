#!/usr/bin/env python3
"""Throughput of FASTA/FASTQ ingest into device batches, end to end with a minimizer scan behind it, for a synthetic FASTQ of
150-bp reads as plain text, gzip and BGZF:
  host records    bl_reader_next_batch: host record parser -> upload
  device parser   bl_reader_next_batch_device: decompressed text spans -> H2D -> parse on the GPU (BGZF: parallel inflate)
Writes one JSON line."""
import gzip, json, os, struct, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import biolib_amd as B
import oracle_lib as O


def bgzf(data, block=65280):
    out = bytearray()
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        z = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = z.compress(chunk) + z.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # the files hold the reads this many times over (records, gzip and BGZF members
                                                        # all concatenate): a few GB, so that the readers' start-up does not dominate
LONG = len(sys.argv) > 4 and sys.argv[4] == "long"     # long reads instead: log-normal lengths (median 8 kbp, a tail beyond 100 kbp),
                                                        # qualities over the whole printable range, one base in 2,000 an N
if LONG:
    rng = np.random.default_rng(7)
    total = n_reads * 150
    lens = []
    while sum(lens) < total:
        lens += [int(x) for x in np.clip(rng.lognormal(np.log(8000), 0.9, 4096), 200, 400_000)]
    cut = int(np.searchsorted(np.cumsum(lens), total))  # the first read that reaches the total: shortened to end there
    lens = lens[:cut + 1]
    lens[-1] -= sum(lens) - total
    flat = O.synth(42, total)
    flat[rng.integers(0, total, total // 2000)] = ord("N")
    offs = np.concatenate([[0], np.cumsum(lens)])
    text = b"".join(b"@read%d runid=0123456789abcdef ch=%d\n" % (i, i % 512) + flat[offs[i]:offs[i + 1]].tobytes() + b"\n+\n" +
                    rng.integers(33, 91, lens[i], dtype=np.uint8).tobytes() + b"\n" for i in range(len(lens)))
    n_bases_file = total
    res = {"workload": f"{len(lens) * repeat} long reads ({total * repeat / 1e6:.0f} Mbp: log-normal lengths, median {int(np.median(lens))} bp, longest {max(lens)} bp; "
                       f"{len(lens)} reads, {repeat} times over), random qualities, 0.05 % N",
           "inflate_threads": threads or "one per core (max 16)", "Gbp_per_s": {}}
else:
    seq = O.synth(42, n_reads * 150).reshape(n_reads, 150)
    text = b"".join(b"@r%d\n" % i + seq[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(n_reads))
    n_bases_file = n_reads * 150
    res = {"workload": f"{n_reads * repeat} reads x 150 bp FASTQ ({n_reads * repeat * 150 / 1e6:.0f} Mbp: {n_reads} reads, {repeat} times over)",
           "inflate_threads": threads or "one per core (max 16)", "Gbp_per_s": {}}
with tempfile.TemporaryDirectory() as d:
    files = {"plain": os.path.join(d, "r.fq"), "gzip": os.path.join(d, "r.fq.gz"), "bgzf": os.path.join(d, "r.bgzf.gz")}
    with open(files["plain"], "wb") as f:
        for _ in range(repeat):
            f.write(text)
    open(files["gzip"], "wb").write(gzip.compress(text, 1) * repeat)
    packed = bgzf(text)
    open(files["bgzf"], "wb").write(packed[:-28] * repeat + packed[-28:])  # (one end-of-file marker, at the end)
    res["file_MB"] = {k: round(os.path.getsize(v) / 1e6) for k, v in files.items()}
    ctx = B.Context(0)
    want = None
    for form, p in files.items():
        for mode in ("host_records", "device_parser"):
            t0 = time.perf_counter(); nb = 0; cnt = 0
            r = B.Reader(p, threads=threads)
            it = (b for b, _, _ in r.batches(ctx, 64_000_000, names=False)) if mode == "host_records" else r.device_batches(ctx, 64 << 20)
            for batch in it:
                nb += batch.n_bases
                cnt += batch.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC).count
                batch.close()
            dt = time.perf_counter() - t0
            assert nb == n_bases_file * repeat
            want = cnt if want is None else want
            assert cnt == want
            res["Gbp_per_s"][f"{form}/{mode}"] = round(nb / dt / 1e9, 3)
    res["minimizers"] = want
print(json.dumps(res))
