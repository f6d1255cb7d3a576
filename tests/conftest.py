import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_kats():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_arrays():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "arrays.npz"))


@pytest.fixture(scope="session", autouse=True)
def _native_library_built():
    """The HIP library is a build artefact (git-ignored): build it when a fresh checkout runs the tests.
    (hipcc cross-compiles gfx950 without a GPU; this is a build step, not a fallback.)"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "biolib_amd", "lib", "libbiolib_amd.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "biolib_amd", "csrc")])
