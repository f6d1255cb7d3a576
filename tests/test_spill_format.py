"""Spill-format compatibility: files written from device k-mer arrays are byte-identical to
what the reference's external_memory_vector / io::basic_store wrote for the same keys (tests/golden/spill/,
generator make_spill_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SP = os.path.join(HERE, "golden", "spill")


def test_run_file_name_matches_reference_scheme():
    from biolib_amd import capi

    L = capi.lib()
    buf = C.create_string_buffer(256)
    assert L.bl_run_file_name(b"/tmp/x", b"first", 3, buf, 256) == 0 and buf.value == b"/tmp/x/tmp.run_first_3.bin"
    assert L.bl_run_file_name(b"d", b"", 0, buf, 256) == 0 and buf.value == b"d/tmp.run_0.bin"
    assert L.bl_run_file_name(b"d", b"n", 0, buf, 4) != 0


@pytest.mark.gpu
def test_run_and_vector_files_byte_identical(tmp_path):
    import torch

    import biolib_amd
    from biolib_amd import capi

    ctx = biolib_amd.Context(0)
    L = capi.lib()
    keys = np.load(os.path.join(SP, "keys.npy"))
    t = torch.from_numpy(np.sort(keys).view(np.int64)).cuda()
    run = str(tmp_path / "run.bin")
    vec = str(tmp_path / "vec.bin")
    capi.check(L.bl_write_run_u64(ctx._h, C.c_void_p(t.data_ptr()), len(keys), run.encode()))
    capi.check(L.bl_write_vector_u64(ctx._h, C.c_void_p(t.data_ptr()), len(keys), vec.encode()))
    assert open(run, "rb").read() == open(os.path.join(SP, "tmp.run_first_0.bin"), "rb").read()
    assert open(vec, "rb").read() == open(os.path.join(SP, "vector.bin"), "rb").read()
    # the library's own sort orders keys as unsigned 64-bit, like the reference's std::sort on uint64_t
    t2 = torch.from_numpy(keys.view(np.int64)).cuda()
    n_unique = ctx.sort_unique(t2)
    assert np.array_equal(t2[:n_unique].cpu().numpy().view(np.uint64), np.unique(keys))
    ctx.close()


def test_read_side_host(tmp_path):
    """bl_file_count_u64 / bl_read_file_u64_host on the files the REFERENCE wrote (tests/golden/spill): the run file is
    the sorted keys, the stored vector (generated from the same sorted keys) carries its count word; malformed sizes are refused"""
    import ctypes as C

    import biolib_amd
    from biolib_amd import capi

    L = capi.lib()
    keys = np.load(os.path.join(SP, "keys.npy"))
    for fn, with_count, exp in (("tmp.run_first_0.bin", 0, np.sort(keys)), ("vector.bin", 1, np.sort(keys))):
        n = C.c_uint64()
        capi.check(L.bl_file_count_u64(os.path.join(SP, fn).encode(), with_count, C.byref(n)))
        assert n.value == len(exp)
        out = np.zeros(n.value, np.uint64)
        capi.check(L.bl_read_file_u64_host(os.path.join(SP, fn).encode(), with_count, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        assert np.array_equal(out, exp)
        assert L.bl_read_file_u64_host(os.path.join(SP, fn).encode(), with_count, out.ctypes.data_as(C.c_void_p), n.value - 1, C.byref(n)) == capi.BL_ERR_CAPACITY
    bad = tmp_path / "odd.bin"
    bad.write_bytes(b"\x00" * 12)
    n = C.c_uint64()
    assert L.bl_file_count_u64(str(bad).encode(), 0, C.byref(n)) == capi.BL_ERR_INVALID
    assert L.bl_file_count_u64(str(bad).encode(), 1, C.byref(n)) == capi.BL_ERR_INVALID
    assert L.bl_file_count_u64(str(tmp_path / "missing.bin").encode(), 0, C.byref(n)) == capi.BL_ERR_INVALID


@pytest.mark.gpu
def test_merge_runs_on_device(tmp_path):
    """run files as the reference's external_memory_vector writes them (sorted chunks of a pushed sequence) -> one sorted
    device array, duplicates kept = what iterating that vector yields; 1, 2, 5 and 8 runs, with an empty run among them"""
    import biolib_amd

    ctx = biolib_amd.Context(0)
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 20, 300_000, dtype=np.uint64)  # plenty of duplicates across runs
    for n_runs in (1, 2, 5, 8):
        cuts = np.sort(rng.integers(0, len(keys), n_runs - 1)) if n_runs > 1 else np.zeros(0, np.int64)
        edges = np.concatenate([[0], cuts, [len(keys)]]).astype(np.int64)
        if n_runs == 5:
            edges[2] = edges[1]  # an empty run
        paths = []
        for i in range(n_runs):
            p = tmp_path / f"tmp.run_x_{n_runs}_{i}.bin"
            np.sort(keys[edges[i]:edges[i + 1]]).tofile(p)
            paths.append(p)
        got = ctx.merge_runs(paths).cpu().numpy().view(np.uint64)
        used = np.concatenate([keys[edges[i]:edges[i + 1]] for i in range(n_runs)])
        assert np.array_equal(got, np.sort(used))
    # and the golden run file the reference itself wrote
    ref_keys = np.load(os.path.join(SP, "keys.npy"))
    got = ctx.merge_runs([os.path.join(SP, "tmp.run_first_0.bin")]).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, np.sort(ref_keys))
    v = ctx.read_file_u64(os.path.join(SP, "vector.bin"), with_count=True).cpu().numpy().view(np.uint64)
    assert np.array_equal(v, np.sort(ref_keys))
    ctx.close()
