#!/usr/bin/env python3
"""Device BGZF inflate rate (bl_bgzf_inflate: one wave per member + CRC-32) on FASTQ text of three kinds, zlib level 6 (what
bgzip writes by default) and level 1.  One JSON line: GB/s of text, members, kernel ms."""
import ctypes as C, json, os, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import biolib_amd as B
import oracle_lib as O


def bgzf(data, level, block=65280):
    out = bytearray()
    for a in range(0, len(data), block):
        chunk = data[a:a + block]
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = z.compress(chunk) + z.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def fastq(n_reads, quality, L=150):
    rng = np.random.default_rng(1)
    seq = O.synth(7, n_reads * L).reshape(n_reads, L)
    if quality == "binned":      # NovaSeq-like: four values, long runs of the best one
        q = np.where(rng.random((n_reads, L)) < 0.93, ord("F"), rng.choice(np.frombuffer(b":,#", np.uint8), (n_reads, L)))
    elif quality == "phred40":   # older instruments: ~40 values, falling towards the end of the read
        q = np.clip(40 - (np.arange(L) / 6)[None, :] - rng.exponential(4, (n_reads, L)), 2, 40).astype(np.int64) + 33
    else:                         # constant
        q = np.full((n_reads, L), ord("I"))
    q = q.astype(np.uint8)
    return b"".join(b"@A00123:45:HXXXXXXXX:1:1101:%d:%d 1:N:0:ACGTACGT\n" % (1000 + i % 30000, 1000 + i // 7) + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n"
                    for i in range(n_reads))


n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 370_000
QUICK = len(sys.argv) > 2 and sys.argv[2] == "quick"  # one case, one launch: for counter passes under rocprofv3
ctx = B.Context(0)
lib = ctx._lib
res = {"reads": n_reads, "cases": {}}
for quality in ("binned",) if QUICK else ("binned", "phred40", "constant"):
    text = fastq(n_reads, quality)
    for level in (6,) if QUICK else (6, 1):
        data = bgzf(text, level)
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
        for _ in range(1 if QUICK else 4):
            ctx.sync()
            t0 = time.perf_counter()
            assert lib.bl_bgzf_inflate(ctx._h, ptrs[0], len(data), ptrs[1], n.value, ptrs[2], tb.value, ptrs[3]) == 0
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        if quality == "binned" and level == 6 and not QUICK:  # how many members run side by side: time against the number of members
            sweep = {}
            for nm in (64, 128, 256, 512, 768, 1024, 1280, 1536, 2048):
                if nm > n.value:
                    break
                t = 1e9
                for _ in range(3):
                    ctx.sync()
                    t0 = time.perf_counter()
                    assert lib.bl_bgzf_inflate(ctx._h, ptrs[0], len(data), ptrs[1], nm, ptrs[2], tb.value, ptrs[3]) == 0
                    ctx.sync()
                    t = min(t, time.perf_counter() - t0)
                sweep[nm] = round(t * 1e3, 2)
            res["ms_by_members"] = sweep
        out = np.zeros(tb.value, np.uint8)
        st = np.zeros(n.value, np.uint32)
        lib.bl_copy_to_host(ctx._h, out.ctypes.data, ptrs[2], tb.value)
        lib.bl_copy_to_host(ctx._h, st.ctypes.data, ptrs[3], 4 * n.value)
        assert not st.any() and out.tobytes() == text
        for p in ptrs:
            lib.bl_device_free(ctx._h, p)
        res["cases"][f"{quality}/level{level}"] = {"text_GBps": round(len(text) / best / 1e9, 2), "ms": round(best * 1e3, 2), "members": n.value,
                                                  "ratio": round(len(text) / len(data), 2), "text_MB": round(len(text) / 1e6)}
print(json.dumps(res))
