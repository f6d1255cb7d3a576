"""GPU test: the C++ drop-in headers (include/compat/) driven the way the reference's own test
programs drive biolib's views, checked against reference-generated vectors and the CPU oracle
(tests/cpp/test_compat_views.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_compat_views():
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_compat_views")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "test_compat_views: OK" in out.stdout


def test_cpp_compat_headers_compile():
    """CPU-only: the drop-in headers and their test compile and link against the C ABI."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "biolib_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "_build", "test_compat_views"))
