"""One gzip stream inflated by many threads (biolib_amd/csrc/bl_pgzip.hpp + ParallelGzip in bl_ingest.cpp): the pieces against
zlib under the sanitizers (tests/emu/pgzip_check.cpp), and the reader with small parts (many seams) against the reader with
the feature off and against the text that was compressed.  CPU only."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "emu", "_build", "pgzip_check")


def fastq(n_reads, L, seed, noisy=True):
    rng = np.random.default_rng(seed)
    seq = O.synth(seed, n_reads * L).reshape(n_reads, L)
    if noisy:
        q = rng.integers(33, 74, (n_reads, L), dtype=np.uint8)
    else:
        q = np.repeat(rng.choice(np.array([70, 70, 70, 58, 44, 35], np.uint8), (n_reads, L // 5)), 5, axis=1)
    return b"".join(b"@SRR1234567.%d %d/1\n" % (i, i) + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(n_reads))


def fasta(n_contigs, length, seed, width=80):
    out = []
    for i in range(n_contigs):
        s = O.synth(seed + i, length).tobytes()
        out.append(b">contig%d\n" % i + b"\n".join(s[j:j + width] for j in range(0, length, width)) + b"\n")
    return b"".join(out)


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, name=None, comment=None, extra=None, hcrc=False, mem_level=8):
    """one gzip member with the optional header fields of RFC 1952 written out by hand"""
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03"
    if extra is not None:
        head += struct.pack("<H", len(extra)) + extra
    if name is not None:
        head += name + b"\0"
    if comment is not None:
        head += comment + b"\0"
    if hcrc:
        head += struct.pack("<H", zlib.crc32(head) & 0xffff)
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    body = c.compress(data) + c.flush()
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def cases():
    fq = fastq(12_000, 150, 5)
    smooth = fastq(20_000, 150, 6, noisy=False)
    fa = fasta(3, 900_000, 7)
    poly = b">low\n" + (b"A" * 70 + b"\n") * 40_000 + b">low2\n" + (b"ACACACACAC" * 7 + b"\n") * 40_000
    yield "level1", member(fq, 1), fq
    yield "level6", member(fq, 6), fq
    yield "level9", member(smooth, 9), smooth
    yield "small_blocks", member(smooth, 6, mem_level=1), smooth              # many short blocks
    yield "fixed", member(fq[:600_000], 6, zlib.Z_FIXED), fq[:600_000]        # nothing the finder looks for
    yield "huffman_only", member(fq, 6, zlib.Z_HUFFMAN_ONLY), fq
    yield "rle", member(smooth, 6, zlib.Z_RLE), smooth
    yield "stored", member(fq[:500_000], 0), fq[:500_000]
    yield "fasta", member(fa, 6), fa
    yield "runs", member(poly, 9) + member(fa, 1), poly + fa                  # matches that overlap themselves; 1000-fold text
    yield "header_fields", member(fq, 6, name=b"reads.fq", comment=b"made for a test", extra=b"XY\x02\0ab", hcrc=True), fq
    third = len(smooth) // 3 // 2400 * 2400
    yield ("members", member(smooth[:third], 6, name=b"a") + member(b"", 6) + member(smooth[third:2 * third], 1) + member(smooth[2 * third:], 9, name=b"c"),
           smooth)
    # what pigz writes: the stream flushed every so often (an empty stored block each time), here every 70,000 bytes of text
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = b"".join(c.compress(smooth[a:a + 70_000]) + c.flush(zlib.Z_SYNC_FLUSH if (a // 70_000) % 2 else zlib.Z_FULL_FLUSH) for a in range(0, len(smooth), 70_000)) + c.flush()
    yield "flushes", b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + body + struct.pack("<II", zlib.crc32(smooth) & 0xffffffff, len(smooth) & 0xffffffff), smooth
    yield "mixed", member(fq[:300_000], 0) + member(fq[300_000:900_000], 6, zlib.Z_FIXED) + member(fq[900_000:], 6), fq


CASES = list(cases())


@pytest.fixture(scope="module")
def checker():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    return CHECK


@pytest.mark.parametrize("name,packed,text", CASES, ids=[c[0] for c in CASES])
def test_pieces_against_zlib(checker, tmp_path, name, packed, text):
    assert zlib.decompressobj(31).decompress(packed[:64]) is not None
    path = tmp_path / (name + ".gz")
    path.write_bytes(packed)
    for part in (4096, 30_000, 65_536):
        out = subprocess.run([checker, str(path), str(part)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.startswith("ok "), (part, out.stdout, out.stderr[-2000:])
        fields = dict(kv.split("=") for kv in out.stdout.split()[1:])
        assert fields["true"] == fields["found"] or int(fields["found"]) - int(fields["true"]) <= 2  # false finds are rare
        if name in ("level6", "level9", "small_blocks", "fasta", "rle"):
            assert int(fields["text"]) > 0.9 * len(text)  # nearly all of it was decoded by parts


def test_crc32_by_carryless_multiplication_equals_zlib(checker):
    out = subprocess.run([checker, "--crc"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok crc"), out.stdout + out.stderr[-2000:]


def test_pieces_survive_damage(checker, tmp_path):
    rng = np.random.default_rng(11)
    _, packed, _ = CASES[1]
    for trial in range(24):
        b = bytearray(packed)
        kind = trial % 3
        if kind == 0:
            for at in rng.integers(20, len(b) - 8, 3):
                b[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            del b[int(rng.integers(len(b) // 2, len(b))):]
        else:
            at = int(rng.integers(20, len(b) - 5000))
            b[at:at + 4000] = rng.integers(0, 256, 4000, dtype=np.uint8).tobytes()
        path = tmp_path / "damaged.gz"
        path.write_bytes(bytes(b))
        out = subprocess.run([checker, str(path), "16384"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (trial, out.stdout, out.stderr[-3000:])


def read_text(path, parallel, part=8192):
    import biolib_amd as B

    os.environ["BL_PGZIP"] = "1" if parallel else "0"
    os.environ["BL_PGZIP_PART"] = str(part)
    try:
        r = B.Reader(str(path), threads=4)
        assert r.kind == "gzip"
        try:
            return b"".join(r.text_spans(1 << 20)), None
        except Exception as e:  # noqa: BLE001 - the reader reports damage as an error
            return None, str(e)
        finally:
            r.close()
    finally:
        os.environ.pop("BL_PGZIP", None)
        os.environ.pop("BL_PGZIP_PART", None)


@pytest.mark.parametrize("name,packed,text", CASES, ids=[c[0] for c in CASES])
def test_reader_many_threads_same_text(tmp_path, name, packed, text):
    path = tmp_path / (name + ".fq.gz")
    path.write_bytes(packed)
    for part in (8192, 100_000):
        got, err = read_text(path, True, part)
        assert err is None and got == text, (part, err)
    got, err = read_text(path, False)
    assert err is None and got == text


def test_reader_many_threads_reports_damage(tmp_path):
    rng = np.random.default_rng(12)
    _, packed, text = CASES[1]
    path = tmp_path / "d.fq.gz"
    seen = 0
    for trial in range(12):
        b = bytearray(packed)
        if trial % 3 == 0:
            b[int(rng.integers(len(b) // 4, len(b) - 8))] ^= 0x10
        elif trial % 3 == 1:
            del b[int(rng.integers(len(b) // 2, len(b) - 1)):]
        else:
            b += b"\x1f\x8b trailing bytes that open like a gzip member and are none"
        path.write_bytes(bytes(b))
        got_p, err_p = read_text(path, True)
        got_z, err_z = read_text(path, False)
        assert (err_p is None) == (err_z is None), (trial, err_p, err_z)
        if err_p is None:
            assert got_p == got_z
        else:
            seen += 1
    assert seen >= 8


TAILS = [b"\0" * 512, b"garbage!", b"\x1f", b"\x1f\x00\x8b", b"\0", b"\n" * 70000]


@pytest.mark.parametrize("case", [1, 2], ids=[CASES[1][0], CASES[2][0]])
def test_trailing_garbage_behind_the_last_member_is_the_end_of_the_stream(tmp_path, case):
    """zero padding / any bytes that do not open with the gzip magic, behind a complete member: read cleanly, as gzread does
    (the reference's drivers read gzFile through kseq); both decoders, parts small enough that the tail has parts of its own"""
    _, packed, text = CASES[case]
    for i, tail in enumerate(TAILS):
        path = tmp_path / ("t%d.fq.gz" % i)
        path.write_bytes(packed + tail)
        for parallel, part in ((True, 8192), (True, 100_000), (False, 8192)):
            got, err = read_text(path, parallel, part)
            assert err is None and got == text, (i, parallel, part, err)
    # a truncation INSIDE a member stays an error
    path = tmp_path / "cut.fq.gz"
    path.write_bytes(packed[: len(packed) - 9])
    for parallel in (True, False):
        got, err = read_text(path, parallel)
        assert err is not None, parallel


def test_reader_records_through_many_threads(tmp_path):
    """the record calls sit on the same byte source: names and sequences of a gzip'ed FASTQ, parts of 8 KiB"""
    import biolib_amd as B

    _, packed, text = CASES[2]
    path = tmp_path / "r.fq.gz"
    path.write_bytes(packed)
    os.environ["BL_PGZIP_PART"] = "8192"
    try:
        r = B.Reader(str(path), threads=4)
        recs = list(r.records())
        r.close()
    finally:
        os.environ.pop("BL_PGZIP_PART", None)
    lines = text.split(b"\n")
    assert len(recs) == len(lines) // 4
    assert all(recs[i][1] == lines[4 * i + 1] and recs[i][0] == lines[4 * i][1:].split(b" ")[0].decode() for i in range(0, len(recs), 97))


def test_reader_closed_early(tmp_path):
    """the consumer leaves after the first span: the decoder's threads notice and wind down"""
    import biolib_amd as B

    _, packed, text = CASES[2]
    path = tmp_path / "e.fq.gz"
    path.write_bytes(packed)
    os.environ["BL_PGZIP_PART"] = "8192"
    try:
        for _ in range(5):
            r = B.Reader(str(path), threads=4)
            first = next(iter(r.text_spans(1 << 16)))
            assert text.startswith(first)
            r.close()
    finally:
        os.environ.pop("BL_PGZIP_PART", None)
