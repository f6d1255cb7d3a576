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


# ---- `python bench.py --gpus N` without a launcher: the process starts its own N ranks ----

_CHILD = r"""
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "fail" and r == 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(60)   # a rank stuck in a collective its peer never joins: the launcher must end it
print('{"rank": %d, "world": %d}' % (r, w), flush=True)
"""


def test_launch_ranks_relays_rank0_and_sets_the_rendezvous(capfd):
    rc = bench.launch_ranks(3, [sys.executable, "-c", _CHILD, "ok"])
    out, err = capfd.readouterr()
    assert rc == 0
    assert out.strip() == '{"rank": 0, "world": 3}'          # ONE line on stdout: rank 0's
    assert '{"rank": 1, "world": 3}' in err and '{"rank": 2, "world": 3}' in err


def test_launch_ranks_worst_exit_code_and_no_orphans(capfd):
    import time

    t0 = time.time()
    rc = bench.launch_ranks(2, [sys.executable, "-c", _CHILD, "fail"])
    assert rc == 7
    assert time.time() - t0 < 30  # rank 0 (sleeping) was terminated, not waited for
    capfd.readouterr()


def test_bench_refuses_a_line_with_fewer_ranks_than_asked_for():
    """no launcher, --gpus 2, fewer than 2 devices visible (none here): exit code 2 and no JSON line, instead of an n_gpus:1 line"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BL_BENCH_REHEARSE")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "{" not in r.stdout and "--gpus 2" in r.stderr
    # a launcher that started another number of ranks than --gpus says
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "{" not in r.stdout and "WORLD_SIZE=1" in r.stderr


def test_kernel_time_must_fit_the_step_it_ran_in():
    """VERDICT r03 weak #4 on its own canned line: C2 timed on two lanes gave avg_kernel_ms 4.4005 x 7 launches per step against
    ms_per_step 16.0 — not a kernel duration; the driver's C3 line (34 launches x 2.9389 ms in 101.6 ms) is one"""
    assert not bench.kernel_time_fits(4.4005, 14, 2, 16.0)
    assert bench.kernel_time_fits(2.585, 14, 2, 18.2)
    assert bench.kernel_time_fits(2.9389, 34 * 20, 20, 101.6)


def test_power_sampler_reports_the_busiest_card(tmp_path):
    """the bench line's `power` object: hwmon sensors of every card are sampled, the summary is of the one that draws the most"""
    import time

    for card, (power, freq) in {"card0": (250_000_000, 150_000_000), "card8": (1_270_000_000, 2_080_000_000)}.items():
        d = tmp_path / card / "device" / "hwmon" / "hwmon3"
        d.mkdir(parents=True)
        (d / "power1_input").write_text(str(power))
        (d / "power1_cap").write_text("1400000000")
        (d / "freq1_input").write_text(str(freq))
    s = bench.PowerSampler(str(tmp_path)).start()
    time.sleep(0.3)
    out = s.finish()
    assert out["card"] == "card8" and out["power_W"] == {"median": 1270, "p90": 1270, "max": 1270} and out["power_cap_W"] == 1400
    assert out["sensor_clock_MHz"]["median"] == 2080 and out["samples"] >= 3
    assert bench.PowerSampler(str(tmp_path / "nothing_here")).start().finish() is None
