"""Host logic of bench.py that needs no GPU: the tie between the static roofline inputs (profiles/traffic.json,
profiles/valu_model.json) and the kernel sources they were measured on."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_profiles_are_tied_to_the_kernel_sources(tmp_path):
    src = os.path.join(ROOT, "biolib_amd", "csrc")
    dst = tmp_path / "biolib_amd" / "csrc"
    os.makedirs(dst)
    for name in bench.KERNEL_SOURCES:
        shutil.copy(os.path.join(src, name), dst / name)
    d0 = bench.kernel_sources_digest(str(tmp_path))
    assert d0 == bench.kernel_sources_digest() and len(d0) == 64
    # a profile stamped with today's sources is fresh; one without a stamp, or stamped on other sources, is stale
    assert not bench.profile_is_stale({"kernel_sources_sha256": d0}, d0)
    assert bench.profile_is_stale({"commit": "debca20"}, d0) and bench.profile_is_stale(None, d0)
    with open(dst / "bl_scan_core.hpp", "a") as f:
        f.write("// a kernel edit\n")
    d1 = bench.kernel_sources_digest(str(tmp_path))
    assert d1 != d0 and bench.profile_is_stale({"kernel_sources_sha256": d0}, d1)


def test_committed_profiles_say_which_sources_they_were_measured_on():
    """what bench.py will print for this tree: the committed traffic / VALU profiles either carry the digest of the sources as
    they are (fresh) or the line says stale — never silently quoted"""
    digest = bench.kernel_sources_digest()
    for path, prov_of in ((bench.TRAFFIC_PATH, lambda d: d.get("provenance")), (bench.MODEL_PATH, lambda d: d["provenance"].get("pmc"))):
        d = json.load(open(path))
        stale = bench.profile_is_stale(prov_of(d), digest)
        assert stale in (True, False)
        if not stale:
            assert prov_of(d)["kernel_sources_sha256"] == digest
