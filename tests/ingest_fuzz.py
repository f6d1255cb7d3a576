"""Shared generator for the ingest fuzz tests: small regular FASTA / FASTQ texts and random mutations of them."""
import numpy as np

ALPHABET = b"\n\n\n\r>@+ACGTNacgt x-"


def base_texts(rng):
    def dna(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes())

    fq = b"".join(b"@r%d c\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(dna(int(rng.integers(1, 60))) for _ in range(6)))
    fa2 = b"".join(b">s%d\n" % i + dna(int(rng.integers(1, 90))) + b"\n" for i in range(5))
    wrapped = b""
    for i in range(3):
        s = dna(int(rng.integers(1, 200)))
        wrapped += b">c%d desc\n" % i + b"\n".join(s[j:j + 40] for j in range(0, len(s), 40)) + b"\n"
    return [fq, fa2, wrapped, fq.replace(b"\n", b"\r\n"), wrapped[:-1]]


def mutate(rng, text):
    t = bytearray(text)
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 7))
        if not t:
            break
        p = int(rng.integers(0, len(t)))
        if kind == 0:
            del t[p]
        elif kind == 1:
            t.insert(p, ALPHABET[int(rng.integers(len(ALPHABET)))])
        elif kind == 2:
            t[p] = ALPHABET[int(rng.integers(len(ALPHABET)))]
        elif kind == 3:
            del t[p:]
        elif kind == 4:  # duplicate a line
            lines = bytes(t).split(b"\n")
            i = int(rng.integers(len(lines)))
            lines.insert(i, lines[i])
            t = bytearray(b"\n".join(lines))
        elif kind == 5:  # swap two lines
            lines = bytes(t).split(b"\n")
            i, j = int(rng.integers(len(lines))), int(rng.integers(len(lines)))
            lines[i], lines[j] = lines[j], lines[i]
            t = bytearray(b"\n".join(lines))
        else:
            t += b"\n" * int(rng.integers(1, 4))
    return bytes(t)


def cases(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        for base in base_texts(rng):
            out.append(base)
            for _ in range(6):
                out.append(mutate(rng, base))
    return out[:n]
