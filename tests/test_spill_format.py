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
