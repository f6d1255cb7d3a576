"""CPU-only: the C-ABI library builds, loads and exports every symbol include/biolib_amd.h declares
(no compute call is made here: there is no GPU in the build container)."""
import os
import re
import subprocess

import pytest

import biolib_amd
from biolib_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(capi.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "biolib_amd", "csrc")])


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "biolib_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)


def test_library_exports_every_symbol():
    L = capi.lib()
    for name in capi.SYMBOLS:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (bl_[a-z0-9_]+)", out))
    assert set(capi.SYMBOLS) <= exported
    assert L.bl_version() == 100


def test_host_hash_matches_reference_kats(golden_kats):
    for v, s, h in golden_kats["hash64_u64"]:
        assert biolib_amd.hash64(v, s) == h


def test_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(biolib_amd.BiolibError):
        biolib_amd.Context(0)
