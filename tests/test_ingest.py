"""FASTA/FASTQ ingest: the library's reader against what the reference's own reader
(tests/kseq.h) returned for the fixture files (tests/golden/ingest/expected.json, generator
make_ingest_golden.py), and — on the GPU — file -> batches -> scans against the oracle on the parsed sequences."""
import json
import os

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
