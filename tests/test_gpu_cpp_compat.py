"""GPU test: the C++ drop-in headers (include/compat/) driven the way the reference's own test
programs drive biolib's views, checked against reference-generated vectors and the CPU oracle
(tests/cpp/test_compat_views.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def built(name):
    """the test program, rebuilt by make whenever a header under include/ (or its source) is newer than the binary: a stale
    binary must not pass for the headers of the tree"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], timeout=900)
    return os.path.join(ROOT, "tests", "cpp", "_build", name)


@pytest.mark.gpu
def test_cpp_compat_views():
    exe = built("test_compat_views")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "test_compat_views: OK" in out.stdout


@pytest.mark.gpu
def test_cpp_compat_jaccard(tmp_path):
    """a Jaccard workflow (what the reference's tests/test_jaccard.cpp does, written as our own caller) over external_memory_vector / ordered_unique_sampler /
    jaccard from include/compat/, on two fixture files"""
    exe = built("test_compat_jaccard")
    ing = os.path.join(ROOT, "tests", "golden", "ingest")
    out = subprocess.run([exe, os.path.join(ing, "many.fa"), os.path.join(ing, "many.fa.gz"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "test_compat_jaccard: OK" in out.stdout, out.stdout[-4000:] + out.stderr[-4000:]
    assert out.stdout.count(" = 1.000000\n") == 4  # the same sequences, plain and gzip: every set equals itself
    out = subprocess.run([exe, os.path.join(ing, "many.fa"), os.path.join(ing, "mixed.fa"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "test_compat_jaccard: OK" in out.stdout, out.stdout[-4000:] + out.stderr[-4000:]
    assert not [f for f in os.listdir(tmp_path) if f.startswith("tmp.run")]  # the vectors removed their run files


@pytest.mark.gpu
def test_cpp_multi_gpu_driver():
    """include/compat/multi_gpu.hpp + bl_count_allreduce (RCCL C API) with every visible device (one on the test box)"""
    exe = built("test_multi_gpu")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "test_multi_gpu: OK" in out.stdout, out.stdout[-4000:] + out.stderr[-4000:]


@pytest.mark.gpu
def test_cpp_view_loop_pooled_vs_per_view(tmp_path):
    """the reference driver's per-read loop through a read_pool: same k-mers as view-by-view, a handful of batch scans
    instead of one GPU round trip per read"""
    import json

    import numpy as np

    import oracle_lib as O

    exe = built("bench_view_loop")
    n_reads, L = 20_000, 150
    seq = O.synth(3, n_reads * L)
    seq[::1009] = ord("N")
    path = tmp_path / "reads.fq"
    with open(path, "wb") as f:
        for i in range(n_reads):
            f.write(b"@r%d\n" % i + seq[i * L:(i + 1) * L].tobytes() + b"\n+\n" + b"I" * L + b"\n")
    out = subprocess.run([exe, str(path), "21", "1", "500"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    dg = O.kmer_digest(seq, O.fixed_offsets(len(seq), L), 21, True, 0, drop_last=True)  # the idiom skips each read's last k-mer
    assert (d["reads"], d["kmers"], d["xor_values"]) == (n_reads, dg["count"], dg["xor_value"])
    d500 = O.kmer_digest(seq[:500 * L], O.fixed_offsets(500 * L, L), 21, True, 0, drop_last=True)
    assert (d["per_view_kmers"], d["per_view_xor"]) == (d500["count"], d500["xor_value"])
    assert d["batch_scans"] <= 2 and d["pooled_reads_per_s"] > 5 * d["per_view_reads_per_s"]


@pytest.mark.gpu
def test_cpp_minimizer_loop_pooled_vs_per_view(tmp_path):
    """the reference's minimizer driver loop (tests/test_minimizer_view.cpp:37-43, k = 15, m = 10, seed 42) over a read_pool:
    the minimizers of every read come from ONE scan per batch — same records as view by view, and as the oracle on the reads"""
    import json

    import numpy as np

    import oracle_lib as O

    exe = built("bench_view_loop")
    rng = np.random.default_rng(6)
    lens = rng.integers(12, 300, 20_000)  # some reads shorter than k: no window, no record
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seq = O.synth(4, int(offs[-1]))
    seq[::911] = ord("N")
    path = tmp_path / "reads.fa"
    with open(path, "wb") as f:
        for i in range(len(lens)):
            f.write(b">r%d\n" % i + seq[int(offs[i]):int(offs[i + 1])].tobytes() + b"\n")
    out = subprocess.run([exe, str(path), "15", "0", "400", "10", "42"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    v, p, h = O.minimizers(seq, offs, 10, 6, 42, False, brute=False)  # unit = m = 10, window = k - m + 1 = 6
    read_of = np.searchsorted(offs, p, side="right") - 1
    rel = p - offs[read_of]
    assert (d["reads"], d["minimizers"], d["xor_values"], d["sum_positions"]) == (len(lens), len(v), O.xor_reduce(v), int(rel.sum()))
    assert d["head_pooled"] == d["head_per_view"] and d["batch_scans"] <= 2
    assert d["pooled_reads_per_s"] > 5 * d["per_view_reads_per_s"]


@pytest.mark.gpu
def test_cpp_super_kmer_loop_pooled_vs_per_view(tmp_path):
    """the reference's super-k-mer driver loop (tests/test_super_kmer_view.cpp:30-36) over a read_pool: every read's super-k-mers come
    from ONE scan per batch — same records as view by view, and as the oracle on the reads"""
    import json

    import numpy as np

    import oracle_lib as O

    exe = built("bench_view_loop")
    rng = np.random.default_rng(16)
    lens = rng.integers(20, 300, 20_000)  # some reads shorter than k
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seq = O.synth(14, int(offs[-1]))
    seq[::733] = ord("N")
    path = tmp_path / "reads.fq"
    with open(path, "wb") as f:
        for i in range(len(lens)):
            r = seq[int(offs[i]):int(offs[i + 1])].tobytes()
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    for canon in (0, 1):
        out = subprocess.run([exe, str(path), "31", str(canon), "300", "13", "7", "super"], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        mn, fp, mp, sz, _ = O.super_kmers(seq, offs, 31, 13, 7, bool(canon))
        read_of = np.searchsorted(offs, fp, side="right") - 1
        rel = fp - offs[read_of]
        want = int(rel.sum()) + int(mp.astype(np.uint64).sum()) + 256 * int(sz.astype(np.uint64).sum())
        assert (d["reads"], d["minimizers"], d["xor_values"], d["sum_positions"]) == (len(lens), len(mn), O.xor_reduce(mn), want)
        assert d["head_pooled"] == d["head_per_view"] and d["batch_scans"] <= 2
        assert d["pooled_reads_per_s"] > 5 * d["per_view_reads_per_s"]


def test_cpp_compat_headers_compile():
    """CPU-only: the drop-in headers and their test compile and link against the C ABI."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "biolib_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "_build", "test_compat_views"))
