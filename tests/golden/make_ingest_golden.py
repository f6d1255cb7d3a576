#!/usr/bin/env python3
"""Generate the ingest fixtures: small FASTA/FASTQ files (plain and gzip, awkward formatting included) and, in
expected.json, what the REFERENCE's own reader (tests/kseq.h through oracle/ref_kseq_shim.cpp) returns for each.
Build container only:  make -C oracle ref && python tests/golden/make_ingest_golden.py"""
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

R = O.ref()
R.ref_kseq_read_all.restype = C.c_long
R.ref_kseq_read_all.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
OUT = os.path.join(HERE, "ingest")
rng = np.random.default_rng(7)


def dna(n, breaks=0):
    s = bytearray(O.synth(int(rng.integers(1, 1 << 30)), n).tobytes())
    for p in rng.integers(0, max(n, 1), breaks):
        s[p] = ord("NnRYK"[int(rng.integers(5))])
    return s.decode()


def wrap(s, width):
    return "\n".join(s[i:i + width] for i in range(0, len(s), width)) if width else s


files = {}
# plain multi-line FASTA with comments, an empty line, Windows line ends in one record, lower case, Ns
seqs = [dna(401, 3), dna(60), dna(1), dna(1300, 10), dna(35).lower()]
fa = f">r1 first read\n{wrap(seqs[0], 70)}\n>r2\tcomment with tab\n{wrap(seqs[1], 60)}\n\n>r3\n{seqs[2]}\n"
fa += ">r4 crlf\r\n" + "\r\n".join(seqs[3][i:i + 80] for i in range(0, len(seqs[3]), 80)) + "\r\n>r5\n" + seqs[4]  # no final newline
files["mixed.fa"] = fa.encode()
# 4-line FASTQ incl. qualities starting with '@' and '+', and a record of length 0
q = [dna(150, 1), dna(150), dna(75), ""]
quals = ["@" + "I" * 149, "+" + "#" * 149, "5" * 75, ""]
files["reads.fq"] = "".join(f"@q{i} len={len(s)}\n{s}\n+\n{ql}\n" for i, (s, ql) in enumerate(zip(q, quals))).encode()
# multi-line FASTQ (sequence and quality wrapped) followed directly by a FASTA record
s2 = dna(333, 2)
files["wrapped.fq"] = (f"@w0\n{wrap(s2, 50)}\n+w0\n{wrap('F' * 333, 61)}\n>tail\n{dna(90)}\n").encode()
# junk before the first header, many short reads
many = [dna(int(rng.integers(20, 260)), int(rng.integers(0, 2))) for _ in range(300)]
files["many.fa"] = ("garbage line\n" + "".join(f">m{i}\n{s}\n" for i, s in enumerate(many))).encode()
# malformed: quality shorter than sequence
files["bad_quality.fq"] = b"@ok\nACGT\n+\nIIII\n@bad\nACGTACGT\n+\nIII\n"

os.makedirs(OUT, exist_ok=True)
expected = {}
for name, data in files.items():
    for gz in (False, True):
        fn = name + (".gz" if gz else "")
        path = os.path.join(OUT, fn)
        if gz:
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write(data)
        else:
            open(path, "wb").write(data)
        cap = len(data) + 16
        bases = np.zeros(cap, np.uint8)
        offs = np.zeros(4096, np.uint64)
        names = np.zeros(cap, np.uint8)
        n = R.ref_kseq_read_all(path.encode(), O._ptr(bases), cap, O._ptr(offs), 4095, O._ptr(names), cap)
        if n < 0:
            expected[fn] = {"error": int(n)}
        else:
            nm = bytes(names).split(b"\x00")[0].decode().split("\n")[:n]
            expected[fn] = {"names": nm, "seqs": [bytes(bases[int(offs[i]):int(offs[i + 1])]).decode("latin1") for i in range(n)]}
json.dump(expected, open(os.path.join(OUT, "expected.json"), "w"), indent=0)
print({k: (v.get("error") or len(v["names"])) for k, v in expected.items()})
