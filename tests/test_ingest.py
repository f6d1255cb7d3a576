"""FASTA/FASTQ ingest: the library's reader against what the reference's own reader
(tests/kseq.h) returned for the fixture files (tests/golden/ingest/expected.json, generator
make_ingest_golden.py), and — on the GPU — file -> batches -> scans against the oracle on the parsed sequences."""
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ING = os.path.join(HERE, "golden", "ingest")
EXPECTED = json.load(open(os.path.join(ING, "expected.json")))


@pytest.mark.parametrize("fn", sorted(EXPECTED))
def test_reader_matches_reference_reader(fn):
    import biolib_amd

    exp = EXPECTED[fn]
    r = biolib_amd.Reader(os.path.join(ING, fn))
    if "error" in exp:
        with pytest.raises(biolib_amd.BiolibError):
            list(r.records())
        return
    got = list(r.records())
    assert [n for n, _ in got] == exp["names"]
    assert [s.decode("latin1") for _, s in got] == exp["seqs"]


def test_reader_missing_file():
    import biolib_amd

    with pytest.raises(biolib_amd.BiolibError):
        biolib_amd.Reader(os.path.join(ING, "does_not_exist.fa"))


@pytest.mark.gpu
@pytest.mark.parametrize("fn", ["mixed.fa.gz", "reads.fq", "wrapped.fq.gz", "many.fa"])
def test_file_to_scan_vs_oracle(fn):
    import biolib_amd

    ctx = biolib_amd.Context(0)
    exp = EXPECTED[fn]
    seq_all = np.frombuffer("".join(exp["seqs"]).encode("latin1"), np.uint8)
    offs_all = np.concatenate([[0], np.cumsum([len(s) for s in exp["seqs"]])]).astype(np.uint64)
    for max_bases in (0, 2000):  # whole file at once / several batches
        names, pos_all, val_all, total = [], [], [], 0
        sync = 0
        for batch, nm, offs in biolib_amd.Reader(os.path.join(ING, fn)).batches(ctx, max_bases):
            assert batch.n_bases == int(offs[-1]) and (max_bases == 0 or batch.n_bases <= max_bases or len(nm) == 1)
            got = batch.minimizers(15, 9, seed=3, canonical=True)
            pos_all.append(got["positions"] + np.uint64(total))
            val_all.append(got["values"])
            sync += batch.syncmers(21, 8, 0, 13, canonical=True, positions=False)["count"]
            total += batch.n_bases
            names += nm
        assert names == exp["names"] and total == len(seq_all)
        v, p, h = O.minimizers(seq_all, offs_all, 15, 9, 3, True, brute=False)
        assert np.array_equal(np.concatenate(pos_all) if pos_all else np.zeros(0, np.uint64), p)
        assert np.array_equal(np.concatenate(val_all) if val_all else np.zeros(0, np.uint64), v)
        assert sync == O.syncmers(seq_all, offs_all, 21, 8, 0, 13, True, positions=False)[0]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fn", ["mixed.fa", "reads.fq", "many.fa", "wrapped.fq", "bad_quality.fq"])
def test_device_side_parser_vs_reference_reader(fn):
    """bl_batch_from_text: same sequences as the reference reader for the regular layouts, an error for the others"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    text = open(os.path.join(ING, fn), "rb").read()
    exp = EXPECTED[fn]
    regular = fn in ("mixed.fa", "reads.fq")  # many.fa has junk before the first header, wrapped.fq is multi-line FASTQ
    if not regular:
        with pytest.raises(biolib_amd.BiolibError):
            ctx.from_text(text)
        if fn == "many.fa":  # without the junk line it is a regular 2-line FASTA
            b = ctx.from_text(text[text.index(b">"):])
            assert bytes(b.download()).decode("latin1") == "".join(exp["seqs"]) and b.n_seqs == len(exp["seqs"])
        ctx.close()
        return
    b = ctx.from_text(text)
    seqs = exp["seqs"]
    assert b.n_seqs == len(seqs) and b.n_bases == sum(len(s) for s in seqs)
    assert bytes(b.download()).decode("latin1") == "".join(seqs)
    seq_all = np.frombuffer("".join(seqs).encode("latin1"), np.uint8)
    offs_all = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    v, p, h = O.minimizers(seq_all, offs_all, 15, 9, 3, True, brute=False)  # sequence boundaries = the parser's offsets
    got = b.minimizers(15, 9, seed=3, canonical=True)
    assert np.array_equal(got["positions"], p) and np.array_equal(got["values"], v)
    ctx.close()


@pytest.mark.gpu
def test_device_side_parser_large_fastq_and_wrapped_fasta():
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(4)
    n_reads = 200_000
    lens = rng.integers(30, 251, n_reads)
    seq = O.synth(5, int(lens.sum()))
    seq[rng.integers(0, len(seq), len(seq) // 1000)] = ord("N")         # breaks, as real reads have them
    low = rng.integers(0, len(seq), len(seq) // 20)
    seq[low] |= 0x20                                                     # soft-masked (lower-case) bases
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    parts = []
    for i in range(n_reads):
        s = seq[int(offs[i]):int(offs[i + 1])].tobytes()
        parts.append(b"@r%d some comment\n" % i + s + (b"\r\n" if i % 1000 == 7 else b"\n") + b"+\n" + b"I" * len(s) + b"\n")
    b = ctx.from_text(b"".join(parts))
    assert b.n_seqs == n_reads and b.n_bases == len(seq) and np.array_equal(b.download(), seq)
    d = O.minimizer_digest(seq, offs, 31, 11, 42, True, threads=8)
    g = b.minimizers_raw(31, 11, 42, biolib_amd.FLAG_CANONICAL | biolib_amd.FLAG_SYNC)
    assert (g.count, g.xor_hash, g.xor_pos) == (d["count"], d["xor_hash"], d["xor_pos"])
    # FASTA wrapped at 60 columns, contigs from 1 base to 3 Mbp
    clens = [1, 59, 60, 61, 100_000, 3_000_000, 777]
    cseq = O.synth(6, sum(clens))
    coffs = np.concatenate([[0], np.cumsum(clens)]).astype(np.uint64)
    fa = []
    for i, L in enumerate(clens):
        s = cseq[int(coffs[i]):int(coffs[i + 1])].tobytes()
        fa.append(b">c%d\n" % i + b"\n".join(s[j:j + 60] for j in range(0, L, 60)) + b"\n")
    b2 = ctx.from_text(b"".join(fa))
    assert b2.n_seqs == len(clens) and np.array_equal(b2.download(), cseq)
    mn, fp, mp, sz, hs = O.super_kmers(cseq[:200_000], np.minimum(coffs, 200_000)[:6], 31, 15, 42, True)
    got = b2.super_kmers(31, 15, seed=42, canonical=True, first=0, n=int(coffs[5]))
    exp_mn, exp_fp, _, exp_sz, _ = O.super_kmers(cseq, coffs, 31, 15, 42, True)
    keep = exp_fp < coffs[5]
    assert np.array_equal(got["first_pos"], exp_fp[keep]) and np.array_equal(got["sizes"], exp_sz[keep])
    ctx.close()


# ---- fuzz: mutated inputs, the reference's reader as the judge -------------------------------------------------
def _ref_read(path):
    """(seqs list) the REFERENCE reader returns for the file, or None when it reports an error"""
    import ctypes as C

    R = O.ref()
    R.ref_kseq_read_all.restype = C.c_long
    R.ref_kseq_read_all.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    cap = 1 << 16
    bases = np.zeros(cap, np.uint8)
    offs = np.zeros(4096, np.uint64)
    names = np.zeros(cap, np.uint8)
    n = R.ref_kseq_read_all(str(path).encode(), O._ptr(bases), cap, O._ptr(offs), 4095, O._ptr(names), cap)
    if n < 0:
        return None
    return [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(n)]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_fuzz_host_reader_vs_reference_reader(tmp_path):
    """600 mutated FASTA/FASTQ texts: the library's host reader returns exactly what the reference's kseq returns,
    and fails exactly where it fails"""
    import biolib_amd
    from ingest_fuzz import cases

    for i, text in enumerate(cases(11, 600)):
        path = tmp_path / "f.txt"
        path.write_bytes(text)
        exp = _ref_read(path)
        try:
            got = [s for _, s in biolib_amd.Reader(path).records()]
        except biolib_amd.BiolibError:
            got = None
        assert got == exp, (i, text)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_padded_gzip_files_read_like_the_reference_reader(tmp_path):
    """block / tape padding behind the last gzip member (zeros, arbitrary bytes): the reference's reader (kseq over gzread)
    returns the records and no error, and so does the library's — plain gzip through both decoders, and BGZF"""
    import gzip

    import biolib_amd

    rng = np.random.default_rng(5)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)) for _ in range(400)]
    text = b"".join(b"@r%d\n" % i + s + b"\n+\n" + b"I" * 150 + b"\n" for i, s in enumerate(reads))
    files = {"one.fq.gz": gzip.compress(text), "two.fq.gz": gzip.compress(text[: len(text) // 2 // 304 * 304]) + gzip.compress(text[len(text) // 2 // 304 * 304:]),
             "b.fq.gz": _bgzf(text, 9000)}
    for name, packed in files.items():
        for j, tail in enumerate((b"", b"\0" * 512, b"garbage!", b"\x1f")):
            path = tmp_path / ("%d_%s" % (j, name))
            path.write_bytes(packed + tail)
            exp = _ref_read(path)
            assert exp == reads, (name, j)
            for env in ({"BL_PGZIP": "1", "BL_PGZIP_PART": "4096"}, {"BL_PGZIP": "0"}):
                os.environ.update(env)
                try:
                    got = [s for _, s in biolib_amd.Reader(path, threads=4).records()]
                finally:
                    for k in env:
                        os.environ.pop(k, None)
                assert got == exp, (name, j, env)


@pytest.mark.gpu
@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_fuzz_device_parser_vs_reference_reader(tmp_path):
    """the device-side parser either refuses a text or returns exactly the reference reader's sequences — never a
    silent mis-parse, never a fault"""
    import biolib_amd
    from ingest_fuzz import cases

    ctx = biolib_amd.Context(0)
    accepted = 0
    rounds = int(os.environ.get("BL_FUZZ_ROUNDS", "1"))  # more seeds for an exploratory run
    texts = [t for r in range(rounds) for t in cases(12 + r, 400)]
    for i, text in enumerate(texts):
        path = tmp_path / "f.txt"
        path.write_bytes(text)
        exp = _ref_read(path)
        try:
            b = ctx.from_text(text)
        except biolib_amd.BiolibError:
            continue
        accepted += 1
        assert exp is not None, (i, text)
        assert b.n_seqs == len(exp) and bytes(b.download()) == b"".join(exp), (i, text)
        # sequence boundaries: every base is its own unit, windows of 2 never cross a boundary
        seq_all = np.frombuffer(b"".join(exp), np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(s) for s in exp])]).astype(np.uint64)
        if len(seq_all):
            v, p, h = O.minimizers(seq_all, offs, 1, 2, 0, False, brute=False)
            got = b.minimizers(1, 2)
            assert np.array_equal(got["positions"], p), (i, text)
        b.close()
    assert accepted >= 80  # the unmutated texts and the harmless mutations
    ctx.close()


# ---- compressed inputs: stream gzip (one inflate thread running ahead), concatenated members, BGZF (parallel inflate) ----------
def _bgzf(data, block=60000, eof=True):
    """BGZF (SAM spec §4.1): gzip members of <= 64 KiB with a 'BC' extra field that holds the member's size - 1, then the EOF marker"""
    import struct
    import zlib

    out = bytearray()
    for a in list(range(0, len(data), block)) + ([None] if eof else []):
        chunk = b"" if a is None else data[a:a + block]
        z = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = z.compress(chunk) + z.flush()
        bsize = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def _fastq(n_reads, seed=1):
    rng = np.random.default_rng(seed)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(rng.integers(1, 300)), p=[.24, .24, .24, .24, .04]).tobytes()) for _ in range(n_reads)]
    text = b"".join(b"@r%d extra\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(seqs))
    return seqs, text


@pytest.mark.parametrize("form", ["plain", "gzip", "multi_member", "bgzf", "bgzf_1thread"])
def test_compressed_inputs_host_records(tmp_path, form):
    import gzip

    import biolib_amd

    seqs, text = _fastq(20_000)
    path = tmp_path / ("r.fq" if form == "plain" else "r.fq.gz")
    if form == "plain":
        path.write_bytes(text)
    elif form == "gzip":
        path.write_bytes(gzip.compress(text, 1))
    elif form == "multi_member":  # cat a.gz b.gz c.gz
        cuts = [0, len(text) // 3, len(text) // 2, len(text)]
        path.write_bytes(b"".join(gzip.compress(text[a:b], 1) for a, b in zip(cuts, cuts[1:])))
    else:
        path.write_bytes(_bgzf(text))
    r = biolib_amd.Reader(path, threads=1 if form == "bgzf_1thread" else 4)
    assert r.kind == {"plain": "plain", "gzip": "gzip", "multi_member": "gzip"}.get(form, "bgzf")
    got = list(r.records())
    assert [s for _, s in got] == seqs and got[7][0] == "r7"


@pytest.mark.parametrize("form", ["gzip", "bgzf"])
def test_damaged_compressed_inputs_fail_loudly(tmp_path, form):
    import gzip

    import biolib_amd

    _, text = _fastq(5_000, seed=2)
    blob = gzip.compress(text, 1) if form == "gzip" else _bgzf(text)
    for damage in ("truncated", "flipped"):
        bad = bytearray(blob)
        if damage == "truncated":
            bad = bad[: len(bad) // 2]
        else:
            bad[len(bad) // 2] ^= 0x55
        path = tmp_path / f"{damage}.gz"
        path.write_bytes(bytes(bad))
        with pytest.raises(biolib_amd.BiolibError):
            list(biolib_amd.Reader(path).records())


@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_text_spans_cut_at_record_boundaries(tmp_path, kind):
    """the spans handed to the device parser: their concatenation is the decompressed text and each one ends where a record ends"""
    import biolib_amd

    seqs, fq = _fastq(30_000, seed=3)
    if kind == "fastq":
        text = fq
    else:
        text = b"".join(b">c%d\n" % i + b"\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + b"\n" for i, s in enumerate(seqs))
    path = tmp_path / "t.gz"
    path.write_bytes(_bgzf(text))
    for limit in (1 << 16, 1 << 20, 0):
        spans = list(biolib_amd.Reader(path).text_spans(limit))
        assert b"".join(spans) == text
        for s in spans:
            assert s[:1] == (b"@" if kind == "fastq" else b">") and s[-1:] == b"\n"
            if kind == "fastq":
                assert s.count(b"\n") % 4 == 0
        if limit:
            assert max(len(s) for s in spans) <= limit and len(spans) >= len(text) // limit


@pytest.mark.parametrize("kind", ["fastq", "fastq_crlf", "fasta"])
def test_text_spans_cut_awkward_records(tmp_path, kind):
    """the cut is found from the last bytes of a span: quality lines that open with '@' or '+', '>' inside FASTA headers and
    sequence lines next to the limit, one record far longer than the limit, CRLF line ends; every span must open a record and
    hold whole records only (the host record reader, pinned against the reference's, parses each span to the same sequences)"""
    import biolib_amd

    rng = np.random.default_rng(5)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(n)).tobytes()) for n in rng.integers(1, 120, 4000)]
    seqs[1500] = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 50_000).tobytes())  # longer than the limits below
    nl = b"\r\n" if kind == "fastq_crlf" else b"\n"
    if kind.startswith("fastq"):
        first = [b"@", b"+", b"I", b"@", b">"]
        text = b"".join(b"@r%d x@y +z" % i + nl + s + nl + b"+" + nl + (first[i % 5] + b"@" * (len(s) - 1))[:len(s)] + nl for i, s in enumerate(seqs))
    else:
        text = b"".join(b">c%d a>b >c" % i + nl + nl.join(s[j:j + 37] for j in range(0, len(s), 37)) + nl for i, s in enumerate(seqs))
    path = tmp_path / "t.txt"
    path.write_bytes(text)
    for limit in (700, 4096, 1 << 16):
        spans = list(biolib_amd.Reader(path).text_spans(limit))
        assert b"".join(spans) == text
        got = []
        for i, sp in enumerate(spans):
            assert sp[:1] == (b"@" if kind.startswith("fastq") else b">")
            assert len(sp) <= limit or len(sp) < 110_000  # only the span of the long record may exceed the limit
            q = tmp_path / "span.txt"
            q.write_bytes(sp)
            got += [s for _, s in biolib_amd.Reader(q).records()]
        assert got == seqs
        assert sum(len(sp) > limit for sp in spans) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("inflate", ["device", "host"])
def test_bgzf_to_device_parser_to_scan(tmp_path, inflate, monkeypatch):
    """.gz -> inflate (on the device, one wave per BGZF member; or by the host's pool) -> device-side parser -> scan, against
    the oracle on the same reads"""
    import biolib_amd

    if inflate == "host":
        monkeypatch.setenv("BL_HOST_INFLATE", "1")

    rng = np.random.default_rng(8)
    n_reads, L = 200_000, 150
    seq = O.synth(21, n_reads * L)
    seq[rng.integers(0, len(seq), len(seq) // 2000)] = ord("N")
    text = b"".join(b"@r%d\n" % i + seq[i * L:(i + 1) * L].tobytes() + b"\n+\n" + b"I" * L + b"\n" for i in range(n_reads))
    path = tmp_path / "reads.fq.gz"
    path.write_bytes(_bgzf(text))
    ctx = biolib_amd.Context(0)
    cnt = xh = xp = 0
    total = 0
    n_batches = 0
    for b in biolib_amd.Reader(path, threads=8).device_batches(ctx, 8 << 20):
        n_batches += 1
        g = b.minimizers_raw(31, 11, 42, biolib_amd.FLAG_CANONICAL | biolib_amd.FLAG_SYNC)
        assert b.n_bases % L == 0
        pos = b.minimizers(31, 11, seed=42, canonical=True)["positions"] + np.uint64(total)
        cnt += int(g.count); xh ^= int(g.xor_hash); xp ^= O.xor_reduce(pos)
        total += b.n_bases
        b.close()
    d = O.minimizer_digest(seq, O.fixed_offsets(len(seq), L), 31, 11, 42, True, threads=8)
    assert total == len(seq) and (cnt, xh, xp) == (d["count"], d["xor_hash"], d["xor_pos"])
    assert n_batches >= len(text) // (8 << 20)
    ctx.close()


@pytest.mark.gpu
def test_bgzf_device_path_awkward_files(tmp_path):
    """the compressed path (device inflate, cut decided from the end of the text on the device) on files that stress the cut:
    wrapped FASTA with records longer than a span, FASTQ whose quality lines open with '@', CRLF, tiny spans, a file without the
    end-of-file marker — same bases and sequence starts as the host record reader; damaged and truncated files are errors"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(12)

    def check(text, limit, eof=True, block=60000, tail=b""):
        path = tmp_path / "t.gz"
        path.write_bytes(_bgzf(text, block=block, eof=eof) + tail)
        want = [s for _, s in biolib_amd.Reader(path).records()]
        r = biolib_amd.Reader(path)
        assert r.kind == "bgzf"
        got, n_seqs = [], 0
        for b in r.device_batches(ctx, limit):
            got.append(bytes(b.download()))
            n_seqs += b.n_seqs
            b.close()
        assert b"".join(got) == b"".join(want) and n_seqs == len(want)

    # FASTA, 70-column lines, records from 1 base to 700 kbp; spans of 256 KiB: several records are longer than a span
    lens = [1, 69, 70, 71, 5000, 700_000, 3, 300_000, 12]
    fa = b"".join(b">c%d x>y\n" % i + b"\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + b"\n"
                  for i, s in enumerate(O.synth(30 + i, n).tobytes() for i, n in enumerate(lens)))
    check(fa, 256 << 10)
    check(fa, 0)
    # FASTQ with quality lines that open with '@' and '+', CRLF line ends, spans of 64 KiB
    seqs = [O.synth(100 + i, int(n)).tobytes() for i, n in enumerate(rng.integers(1, 400, 3000))]
    first = [b"@", b"+", b"I", b"@", b">"]
    for nl in (b"\n", b"\r\n"):
        fq = b"".join(b"@r%d a@b" % i + nl + s + nl + b"+" + nl + (first[i % 5] + b"@" * (len(s) - 1))[:len(s)] + nl for i, s in enumerate(seqs))
        check(fq, 64 << 10)
        check(fq, 64 << 10, eof=False, block=9000)
        for tail in (b"\0" * 512, b"garbage!", b"\x1f"):  # padding behind the last member: the end of the stream, as for gzread
            check(fq, 64 << 10, tail=tail)
            check(fq, 64 << 10, eof=False, block=9000, tail=tail)
    # long reads (nanopore-like): records from 1 kbp to 600 kbp with noisy quality strings, spans of 256 KiB — most records are
    # longer than the window of text that is looked at first for the cut, some are longer than a span
    lens = [int(x) for x in rng.integers(1000, 60_000, 60)] + [600_000, 5, 300_000]
    long_seqs = [O.synth(900 + i, n).tobytes() for i, n in enumerate(lens)]
    fq = b"".join(b"@read%d ch=%d\n" % (i, i % 512) + s + b"\n+\n" + rng.integers(33, 74, len(s), dtype=np.uint8).tobytes() + b"\n" for i, s in enumerate(long_seqs))
    check(fq, 256 << 10)
    check(fq, 0)
    # damaged / truncated
    text = b"".join(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(seqs)) * 4
    good = bytearray(_bgzf(text))
    for how in ("flip", "truncate", "header"):
        data = bytearray(good)
        if how == "flip":
            data[len(data) // 2] ^= 0x10
        elif how == "truncate":
            data = data[:len(data) // 2]
        else:
            second = struct.unpack_from("<H", data, 16)[0] + 1
            data[second + 12] = ord("X")
        path = tmp_path / "bad.gz"
        path.write_bytes(bytes(data))
        with pytest.raises(biolib_amd.BiolibError):
            for b in biolib_amd.Reader(path).device_batches(ctx, 1 << 20):
                b.close()
    ctx.close()


def test_shard_ranges_are_member_boundaries(tmp_path):
    """CPU: the parts of a BGZF file (bl_reader_open_shard) begin at member boundaries, follow each other without a gap and
    cover the file; a deflate payload that happens to hold the gzip magic is not taken for a member; an empty first member
    does not hide the format; plain gzip is refused"""
    import gzip
    import struct

    import biolib_amd

    rng = np.random.default_rng(3)
    # incompressible text (stored blocks: the payload IS the text) seeded with fake BGZF headers
    fake = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x40\x00"
    body = bytearray(rng.integers(0, 256, 600_000, dtype=np.uint8).tobytes())
    for at in range(1000, len(body) - 100, 9973):
        body[at:at + len(fake)] = fake
    text = b">x\n" + bytes(body)
    data = _bgzf(text, block=30000)
    path = tmp_path / "f.gz"
    path.write_bytes(data)
    offs, at = [], 0
    while at < len(data):
        offs.append(at)
        at += struct.unpack_from("<H", data, at + 16)[0] + 1
    for world in (1, 2, 3, 7, 64):
        prev_end = 0
        for rank in range(world):
            r = biolib_amd.Reader(path, shard=(rank, world))
            a, b = r.shard_range
            r.close()
            assert a == prev_end and (a in offs or a == len(data)), (world, rank, a)
            assert (b == 2**64 - 1) == (rank == world - 1)
            if b != 2**64 - 1:
                assert b >= a and (b in offs or b == len(data)) and b >= len(data) // world * (rank + 1)
                prev_end = b
    # an empty member in front of the text
    empty = _bgzf(b"", eof=False)[:0] + _bgzf(b"")  # (the end-of-file marker alone is an empty member)
    (tmp_path / "e.gz").write_bytes(empty + _bgzf(b"@r\nACGT\n+\nIIII\n"))
    biolib_amd.Reader(tmp_path / "e.gz", shard=(1, 2)).close()
    (tmp_path / "g.gz").write_bytes(gzip.compress(b"@r\nACGT\n+\nIIII\n"))
    with pytest.raises(biolib_amd.BiolibError):
        biolib_amd.Reader(tmp_path / "g.gz", shard=(0, 2))


def _parts_text(kind, rng):
    """(sequences, file text) of the files that are read in parts"""
    if kind == "fastq":
        seqs = [O.synth(500 + i, int(n)).tobytes() for i, n in enumerate(rng.integers(1, 400, 6000))]
        first = [b"@", b"+", b"I", b"@", b">"]
        text = b"".join(b"@r%d a@b" % i + b"\n" + s + b"\n+\n" + (first[i % 5] + b"@" * (len(s) - 1))[:len(s)] + b"\n" for i, s in enumerate(seqs))
    elif kind == "fastq_long":  # nanopore-like: most members lie inside ONE record, quality strings hold every printable character
        lens = [int(x) for x in rng.integers(1000, 150_000, 40)] + [400_000, 7]
        seqs = [O.synth(800 + i, n).tobytes() for i, n in enumerate(lens)]
        text = b"".join(b"@read%d\n" % i + s + b"\n+\n" + rng.integers(33, 127, len(s), dtype=np.uint8).tobytes() + b"\n" for i, s in enumerate(seqs))
    else:
        lens = [1, 69, 70, 71, 5000, 400_000, 3, 150_000, 12] + [int(x) for x in rng.integers(1, 3000, 300)]
        seqs = [O.synth(700 + i, n).tobytes() for i, n in enumerate(lens)]
        text = b"".join(b">c%d x>y\n" % i + b"\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + b"\n" for i, s in enumerate(seqs))
    return seqs, text


@pytest.mark.parametrize("kind", ["fastq", "fasta", "fastq_long"])
def test_plain_file_read_in_parts(tmp_path, kind):
    """CPU: a plain FASTA / FASTQ file read by `world` readers that take a byte range each (bl_reader_open_shard): the ranges
    follow each other, meet at record starts, and the parts' records and text, in rank order, are the file's"""
    import biolib_amd

    seqs, text = _parts_text(kind, np.random.default_rng(31))
    path = tmp_path / "plain.txt"
    path.write_bytes(text)
    line_starts = {0} | {i + 1 for i in range(len(text)) if text[i:i + 1] == b"\n"} if len(text) < 3_000_000 else None
    for world in (1, 2, 3, 5, 8, 40, 5000):
        recs, spans, prev_end = [], [], 0
        for rank in range(world):
            r = biolib_amd.Reader(path, shard=(rank, world))
            a, b = r.shard_range
            assert a == prev_end and b >= a and (rank + 1 < world or b == len(text)), (world, rank, a, b)
            assert a == len(text) or text[a:a + 1] == (b"@" if kind != "fasta" else b">")
            if line_starts is not None:
                assert a in line_starts or a == len(text)
            prev_end = b
            if world <= 40:
                recs += [s for _, s in r.records()]
                r.close()
                r = biolib_amd.Reader(path, shard=(rank, world))
                spans += list(r.text_spans(1 << 16))
            r.close()
        if world <= 40:
            assert recs == seqs and b"".join(spans) == text, (kind, world)
    (tmp_path / "empty.fa").write_bytes(b"")
    for rank in range(3):
        r = biolib_amd.Reader(tmp_path / "empty.fa", shard=(rank, 3))
        assert r.shard_range == (0, 0) and list(r.records()) == []
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fastq", "fasta", "fastq_long"])
def test_bgzf_file_read_in_parts(tmp_path, kind):
    """one BGZF file read by `world` readers that take a part each (bl_reader_open_shard): the parts' sequences, in rank order,
    are the file's sequences — whatever the number of parts (more parts than members included), the span size, and wherever the
    member boundaries fall inside the records; the readers never see each other"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(31)
    seqs, text = _parts_text(kind, rng)
    whole = b"".join(seqs)
    for block, eof in ((60000, True), (7001, False)):
        path = tmp_path / f"parts_{block}.gz"
        path.write_bytes(_bgzf(text, block=block, eof=eof))
        n_members = (len(text) + block - 1) // block
        for world, limit in ((1, 0), (2, 0), (3, 256 << 10), (5, 64 << 10), (8, 0), (n_members + 3 if block == 60000 else 40, 0)):
            got, n_seqs, sizes = [], 0, []
            for rank in range(world):
                r = biolib_amd.Reader(path, shard=(rank, world))
                part = 0
                for b in r.device_batches(ctx, limit):
                    got.append(bytes(b.download()))
                    n_seqs += b.n_seqs
                    part += b.n_bases
                    b.close()
                r.close()
                sizes.append(part)
            assert b"".join(got) == whole, (kind, block, world, sizes)
            assert n_seqs == len(seqs) and sum(sizes) == len(whole)
            if world in (2, 3) and kind == "fastq":  # (short reads: the parts are cut close to the byte targets)
                assert min(sizes) > 0.5 * len(whole) / world  # the parts are about equal
    # a reader of one part of a BGZF file refuses the host calls
    r = biolib_amd.Reader(path, shard=(0, 2))
    with pytest.raises(biolib_amd.BiolibError):
        list(r.records())
    r.close()
    # the same file as plain text, read in parts into device batches
    plain = tmp_path / "plain.txt"
    plain.write_bytes(text)
    for world, limit in ((1, 0), (3, 256 << 10), (8, 0), (40, 64 << 10)):
        got, n_seqs = [], 0
        for rank in range(world):
            r = biolib_amd.Reader(plain, shard=(rank, world))
            for b in r.device_batches(ctx, limit):
                got.append(bytes(b.download()))
                n_seqs += b.n_seqs
                b.close()
            r.close()
        assert b"".join(got) == whole and n_seqs == len(seqs), (kind, world)
    ctx.close()
