"""BGZF on the device (biolib_amd/csrc/bl_inflate_core.hpp, bl_inflate.hip).  CPU: the decoder compiled for the host against
zlib (sound and damaged streams, under the sanitizers: tests/emu/emu_inflate.cpp) and the member walk.  GPU: the kernels
against zlib's text, CRC-32 included, and damaged members."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bgzf(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, block=65280, eof=True):
    out = bytearray()
    for a in list(range(0, len(data), block)) + ([None] if eof else []):
        chunk = b"" if a is None else data[a:a + block]
        z = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        body = z.compress(chunk) + z.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def fastq_text(n_reads, seed=1, L=150):
    rng = np.random.default_rng(seed)
    seq = O.synth(seed, n_reads * L).reshape(n_reads, L)
    qual = (rng.integers(0, 8, (n_reads, L)) == 0) * rng.integers(0, 40, (n_reads, L)) + 35
    return b"".join(b"@r%d/1\n" % i + seq[i].tobytes() + b"\n+\n" + qual[i].astype(np.uint8).tobytes() + b"\n" for i in range(n_reads))


def test_decoder_on_host_vs_zlib():
    """the wave decoder, lane loops instead of lanes: 600 sound streams of every block type, 6000 damaged ones with zlib as the
    judge, address + undefined-behaviour sanitizers on"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    out = subprocess.run([os.path.join(ROOT, "tests", "emu", "_build", "emu_inflate"), "600", "6000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "emu_inflate: OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def _gnu_gzip_payload(chunk, level):
    """raw deflate data as GNU gzip writes it (its own encoder, not zlib's: other block splits, other code lengths)"""
    out = subprocess.run(["gzip", "-n", "-c", f"-{level}"], input=chunk, capture_output=True, check=True).stdout
    assert out[:3] == b"\x1f\x8b\x08" and out[3] == 0  # 10-byte header, no optional fields
    return out[10:-8]


def bgzf_from_gnu_gzip(data, level, block=65280):
    out = bytearray()
    for a in range(0, len(data), block):
        chunk = data[a:a + block]
        body = _gnu_gzip_payload(chunk, level)
        if 12 + 6 + len(body) + 8 > 65536:  # would not fit a BGZF member: store it
            z = zlib.compressobj(0, zlib.DEFLATED, -15)
            body = z.compress(chunk) + z.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def test_decoder_on_host_vs_gnu_gzip(tmp_path):
    """streams from a second encoder (GNU gzip, levels 1 / 6 / 9) through the host build of the decoder"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    exe = os.path.join(ROOT, "tests", "emu", "_build", "emu_inflate")
    rng = np.random.default_rng(4)
    texts = [fastq_text(400, seed=2)[:65000], rng.integers(0, 4, 60000, dtype=np.uint8).tobytes(), bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) * 20,
             b"A" * 65536, b"", b"x", open(os.path.join(ROOT, "SURVEY.md"), "rb").read()[:60000]]
    for i, text in enumerate(texts):
        for level in (1, 6, 9):
            (tmp_path / "t.bin").write_bytes(text)
            (tmp_path / "p.bin").write_bytes(_gnu_gzip_payload(text, level))
            out = subprocess.run([exe, "--check", str(tmp_path / "p.bin"), str(tmp_path / "t.bin")], capture_output=True, text=True)
            assert out.returncode == 0, (i, level, out.stdout)


def test_member_walk():
    import ctypes as C

    from biolib_amd import capi

    lib = capi.lib()
    text = fastq_text(3000)
    data = bgzf(text, block=20000)
    members = np.zeros(4 * 100, np.uint64)
    n, used, tb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert lib.bl_bgzf_walk(data, len(data), 1000, 77, members.ctypes.data, 100, C.byref(n), C.byref(used), C.byref(tb)) == 0
    assert used.value == len(data) and tb.value == len(text) and n.value == (len(text) + 19999) // 20000 + 1
    m = members[:4 * n.value].reshape(-1, 4)
    src_off, dst_off = m[:, 0].astype(np.int64) - 1000, m[:, 1].astype(np.int64) - 77
    src_len, isize = (m[:, 2] & 0xFFFFFFFF).astype(np.int64), (m[:, 2] >> 32).astype(np.int64)
    crc = (m[:, 3] & 0xFFFFFFFF).astype(np.int64)
    at = 0
    for i in range(n.value):
        chunk = text[dst_off[i]:dst_off[i] + isize[i]]
        assert dst_off[i] == at and zlib.decompress(data[src_off[i]:src_off[i] + src_len[i]], -15) == chunk and crc[i] == zlib.crc32(chunk)
        at += isize[i]
    # a buffer that ends inside a member: the walk stops in front of it
    assert lib.bl_bgzf_walk(data, len(data) - 5, 0, 0, members.ctypes.data, 100, C.byref(n), C.byref(used), C.byref(tb)) == 0
    assert n.value == (len(text) + 19999) // 20000 and used.value < len(data) - 5
    # capacity reached
    assert lib.bl_bgzf_walk(data, len(data), 0, 0, members.ctypes.data, 2, C.byref(n), C.byref(used), C.byref(tb)) == 0 and n.value == 2 and tb.value == 40000
    # not BGZF
    bad = bytearray(data)
    bad[12] = ord("X")
    assert lib.bl_bgzf_walk(bytes(bad), len(bad), 0, 0, members.ctypes.data, 100, C.byref(n), C.byref(used), C.byref(tb)) == capi.BL_ERR_INVALID


@pytest.mark.gpu
def test_device_inflate_vs_zlib():
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(2)
    fq = fastq_text(20_000, seed=3)
    runs = b"".join(bytes([65 + int(rng.integers(0, 4))]) * int(rng.integers(1, 4000)) for _ in range(800))
    noise = rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    skew = rng.choice(np.arange(256, dtype=np.uint8), 500_000, p=np.r_[[0.6, 0.25, 0.1], np.full(253, 0.05 / 253)]).tobytes()
    for level in (1, 6, 9):  # a second encoder: GNU gzip
        got, status = ctx.bgzf_inflate(bgzf_from_gnu_gzip(fq[:3_000_000] + skew[:200_000], level))
        assert not status.any() and got.tobytes() == fq[:3_000_000] + skew[:200_000], level
    cases = [(fq, 1, zlib.Z_DEFAULT_STRATEGY, 65280), (fq, 6, zlib.Z_DEFAULT_STRATEGY, 65280), (fq, 9, zlib.Z_DEFAULT_STRATEGY, 65536), (fq, 6, zlib.Z_FIXED, 4096),
             (fq[:300_000], 0, zlib.Z_DEFAULT_STRATEGY, 65280), (runs, 6, zlib.Z_DEFAULT_STRATEGY, 65280), (runs, 6, zlib.Z_RLE, 30000),
             (noise, 6, zlib.Z_DEFAULT_STRATEGY, 65280), (skew, 9, zlib.Z_DEFAULT_STRATEGY, 65280), (skew, 6, zlib.Z_HUFFMAN_ONLY, 65280),
             (b"", 6, zlib.Z_DEFAULT_STRATEGY, 65280), (b"A", 6, zlib.Z_DEFAULT_STRATEGY, 65280), (fq[:70_001], 6, zlib.Z_DEFAULT_STRATEGY, 17)]
    for text, level, strategy, block in cases:
        if block == 17:
            text = text[:5000]
        got, status = ctx.bgzf_inflate(bgzf(text, level, strategy, block))
        assert not status.any(), (level, strategy, block, status[status != 0][:5])
        assert got.tobytes() == text, (level, strategy, block)
    ctx.close()


@pytest.mark.gpu
def test_device_inflate_many_code_sets():
    """3,000 members, each with a text of its own kind and size and therefore code sets of its own (alphabets from 1 to 256
    symbols, flat and steep distributions, runs, far and near repeats; zlib levels 1-9, fixed / Huffman-only / RLE strategies):
    the device builds every table with ballots over one symbol per lane, the host build of the decoder by walking the symbols"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(77)
    texts, data = [], bytearray()
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    for i in range(3000):
        n = int(rng.integers(1, 9000))
        kind = i % 6
        if kind == 0:
            k = int(rng.integers(1, 257))
            t = rng.integers(0, k, n, dtype=np.uint8)
        elif kind == 1:  # steep: a few common symbols and a long tail (long codes)
            k = int(rng.integers(2, 257))
            p = 1.0 / np.arange(1, k + 1) ** float(rng.uniform(0.5, 3.0))
            t = rng.choice(np.arange(k, dtype=np.uint8), n, p=p / p.sum())
        elif kind == 2:
            t = np.repeat(rng.integers(0, 256, n // 7 + 1, dtype=np.uint8), rng.integers(1, 15, n // 7 + 1))[:n]
        elif kind == 3:
            base = rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)
            t = np.tile(base, n // len(base) + 1)[:n].copy()
            t[rng.integers(0, n, n // 50 + 1)] ^= 1
        elif kind == 4:
            t = np.frombuffer(fastq_text(40, seed=i)[:n], np.uint8)
        else:
            t = rng.integers(0, 4, n, dtype=np.uint8) + 65
        text = t.tobytes()
        texts.append(text)
        data += bgzf(text, int(rng.integers(1, 10)), strategies[int(rng.integers(0, 5))], block=65536, eof=False)
    got, status = ctx.bgzf_inflate(bytes(data))
    assert len(status) == len(texts) and not status.any(), np.nonzero(status)[0][:10]
    assert got.tobytes() == b"".join(texts)
    ctx.close()


@pytest.mark.gpu
def test_device_inflate_damaged_members():
    """bit flips, garbage and wrong trailers inside some members: exactly those members report an error (their CRC-32 or size no
    longer fits, or their deflate data is no stream), the others are inflated as if nothing had happened, and the call returns"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(7)
    text = fastq_text(40_000, seed=5)
    block = 30_000
    data = bytearray(bgzf(text, 6, block=block))
    # member boundaries from the headers
    offs, at = [], 0
    while at < len(data):
        offs.append(at)
        at += struct.unpack_from("<H", data, at + 16)[0] + 1
    offs.append(len(data))
    n = len(offs) - 1
    hit = sorted(set(int(x) for x in rng.integers(0, n - 1, 60)))
    for j, i in enumerate(hit):
        a, b = offs[i] + 18, offs[i + 1] - 8  # deflate data
        how = j % 4
        if how == 0:
            data[int(rng.integers(a, b))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            p = int(rng.integers(a, b - 16))
            data[p:p + 16] = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
        elif how == 2:
            data[b] ^= 0x40  # CRC-32 in the trailer
        else:
            data[a:b] = rng.integers(0, 256, b - a, dtype=np.uint8).tobytes()
    got, status = ctx.bgzf_inflate(bytes(data))
    assert len(status) == n and len(got) == len(text)
    bad = set(np.nonzero(status)[0].tolist())
    assert bad == set(hit)
    for i in range(n - 1):
        if i not in bad:
            assert got[i * block:(i + 1) * block].tobytes() == text[i * block:(i + 1) * block]
    ctx.close()


@pytest.mark.gpu
def test_device_inflate_large_and_fast():
    """~190 MB of FASTQ text: same bytes as zlib; the rate is printed (the host pool of 16 threads reaches about 6 GB/s)"""
    import time

    import biolib_amd

    ctx = biolib_amd.Context(0)
    text = fastq_text(600_000, seed=9)
    data = bgzf(text, 1)
    got, status = ctx.bgzf_inflate(data)  # includes allocation and copies
    assert not status.any() and got.tobytes() == text
    import ctypes as C

    lib = ctx._lib
    cap = len(data) // 26 + 1
    members = np.zeros(cap * 4, np.uint64)
    n, used, tb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert lib.bl_bgzf_walk(data, len(data), 0, 0, members.ctypes.data, cap, C.byref(n), C.byref(used), C.byref(tb)) == 0
    ptrs = []
    for size in (len(data) + 8, 32 * n.value, tb.value + 16, 4 * n.value):
        p = C.c_void_p()
        assert lib.bl_device_alloc(ctx._h, size, C.byref(p)) == 0
        ptrs.append(p)
    assert lib.bl_copy_to_device(ctx._h, ptrs[0], data, len(data)) == 0 and lib.bl_copy_to_device(ctx._h, ptrs[1], members.ctypes.data, 32 * n.value) == 0
    best = 1e9
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        assert lib.bl_bgzf_inflate(ctx._h, ptrs[0], len(data), ptrs[1], n.value, ptrs[2], tb.value, ptrs[3]) == 0
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"\ndevice inflate: {len(text) / best / 1e9:.1f} GB/s of text ({n.value} members, {len(data) / 1e6:.0f} MB packed, {best * 1e3:.1f} ms)")
    for p in ptrs:
        lib.bl_device_free(ctx._h, p)
    assert len(text) / best > 5e9
    ctx.close()
