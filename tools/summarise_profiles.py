#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (tools/collect_profiles.sh, run on the GPU box) into the files under profiles/ that bench.py and
DESIGN.md cite: rNN_kernel_stats.csv (+ _lanes1), rNN_pmc_summary.txt, rNN_pmc.json, rNN_ubench_valu.{txt,json},
traffic.json — each stamped with where and from which commit it was measured.  Usage: summarise_profiles.py [rNN]"""
import collections, csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
from bench import kernel_sources_digest  # what bench.py compares at run time: traffic_stale / valu_stale

# the collection must have been made on the kernel sources of this tree (a refused or failed gpurun call leaves the previous one in place)
_stamp = os.path.join(src, "sources.sha256")
if os.path.exists(_stamp):
    assert open(_stamp).read().strip() == kernel_sources_digest(), "gpurun_out/prof was collected on other kernel sources than this tree's: run tools/collect_profiles.sh again"
else:
    print("warning: gpurun_out/prof/sources.sha256 missing (collected by an older tools/collect_profiles.sh): cannot check what was measured", file=sys.stderr)
git = lambda *a: subprocess.run(["git", "-C", ROOT, *a], capture_output=True, text=True).stdout.strip()
box = open(os.path.join(src, "box.txt")).read().split("\n")
prov = {"commit": git("rev-parse", "--short", "HEAD") + ("+dirty" if git("status", "--porcelain", "--untracked-files=no") else ""),
        "box": box[0], "collected_utc": box[1] if len(box) > 1 else "", "command": open(os.path.join(src, "pmc_command.txt")).read().strip(),
        "tools": "tools/collect_profiles.sh + tools/summarise_profiles.py", "kernel_sources_sha256": kernel_sources_digest()}

# bases per launch of every configuration, from the JSON line the PMC command itself printed (its workloads are bench.py's defaults)
line = [x for x in open(os.path.join(src, "pmc_sq1.log")).read().splitlines() if x.startswith("{")][-1]
bj = json.loads(line)
per_launch = {"c3": bj["config"]["bases_per_launch"]}
for cfg, key in (("c2", "C2_kmer_hash_10Gbp"), ("c4", "C4_super_kmers_50Gbp_10kbp_reads"), ("c5", "C5_syncmers_50Gbp_shard_10kbp_reads")):
    oc = bj["other_configs"][key]
    per_launch[cfg] = oc["bases"] * oc.get("kernel_steps", oc["steps"]) / oc["launches_timed"]
BASES_OF = {"c3_count": per_launch["c3"], "c3_emit": per_launch["c3"], "c2_kmer": per_launch["c2"], "c4_count": per_launch["c4"], "c4_emit": per_launch["c4"],
            "c5_count": per_launch["c5"], "c5_emit": per_launch["c5"], "c3_redo": per_launch["c3"], "c5_redo": per_launch["c5"]}

shutil.copy(os.path.join(src, "stats_lanes2", "s_kernel_stats.csv"), os.path.join(dst, f"{RND}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "stats_lanes1", "s_kernel_stats.csv"), os.path.join(dst, f"{RND}_kernel_stats_lanes1.csv"))
shutil.copy(os.path.join(src, "ubench_valu.txt"), os.path.join(dst, f"{RND}_ubench_valu.txt"))
shutil.copy(os.path.join(src, "ubench_valu.json"), os.path.join(dst, f"{RND}_ubench_valu.json"))
for lanes in ("lanes2", "lanes1"):
    line = [x for x in open(os.path.join(src, f"stats_{lanes}.log")).read().splitlines() if x.startswith("{")]
    if line:
        open(os.path.join(dst, f"{RND}_bench_1gpu{'' if lanes == 'lanes2' else '_lanes1'}_under_rocprof.json"), "w").write(line[-1] + "\n")

# power / clock sensors of the card beside the headline scan (tools/power_probe.py), and the shape sweep
power = {}
for lanes in ("lanes2", "lanes1"):
    pj, lg = os.path.join(src, f"power_{lanes}.json"), os.path.join(src, f"power_{lanes}.log")
    if os.path.exists(pj):
        d = json.load(open(pj))
        # the card that ran the scan: the host's other cards (other tenants') show in hwmon too; ours draws the most on average
        card = max((k for k in d["sensors"] if k.endswith("power1_input")), key=lambda k: d["sensors"][k]["mean"]).split(":")[0]
        line = [x for x in open(lg).read().splitlines() if x.startswith("{\"metric")] if os.path.exists(lg) else []
        bl = json.loads(line[-1]) if line else {}
        power[lanes] = {"sensors": {k: v for k, v in d["sensors"].items() if k.startswith(card + ":")}, "seconds": d["seconds"],
                        "bench": {"value_Gbps": bl.get("value"), "shader_clock_GHz_in_timed_region": bl.get("roofline", {}).get("valu", {}).get("shader_clock_GHz"), "steps": bl.get("steps")},
                        "trace_every_10th": [(t, {k.split(":")[1]: v for k, v in s.items() if k.startswith(card + ":")}) for t, s in d["trace_every_10th"]]}
if power:
    json.dump({"provenance": prov, "note": "hwmon sensors of the card sampled every 50 ms beside `bench.py --steps 40` (C3, 50 Gbp): power in microwatts, clocks in Hz", **power},
              open(os.path.join(dst, f"{RND}_power_probe.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "shape_sweep.json")):
    sw = json.loads(open(os.path.join(src, "shape_sweep.json")).read().strip().splitlines()[-1])
    sw["provenance"] = prov
    json.dump(sw, open(os.path.join(dst, f"{RND}_shape_sweep.json"), "w"), indent=1)

# kernel key -> substring of the kernel name rocprofv3 reports
kernels = {
    "c3_count": "scan_count_frl_kernel<0, 11, 15, 31, 150, 1, true>", "c3_emit": "scan_emit_kernel<0, 31, 1, 1>",
    "c2_kmer": "kmer_kernel",
    "c4_count": "scan_count_kernel<1, 17, 15, 1, 0>", "c4_emit": "scan_emit_kernel<1, 0, -1, -1>",
    "c5_count": "scan_count_kernel<2, 21, 11, 1, 1>", "c5_emit": "scan_emit_kernel<2, 0, -1, -1>",
    "c3_redo": "scan_redo_frl_kernel<0, 11, 15, 31, 150, 1>", "c5_redo": "scan_redo_kernel<2, 21, 11, 1>",  # the tiles pass 1 could not decide, again
}
means = {k: {} for k in kernels}
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_sq3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        for k, name in kernels.items():
            if name in row["Kernel_Name"]:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in acc:
        for c, v in acc[k].items():
            means[k][c] = (sum(v) / len(v), len(v))
with open(os.path.join(dst, f"{RND}_pmc_summary.txt"), "w") as f:
    f.write(f"# {json.dumps(prov)}\n# mean per launch; bases per launch: {json.dumps({k: round(v) for k, v in per_launch.items()})}; FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE to be doubled on gfx950 (MI355X_MICROARCH.md, HBM)\n")
    for k in kernels:
        f.write(f"== {k}: {kernels[k]}\n")
        for c, (m, n) in sorted(means[k].items()):
            f.write(f"{c:28s} n={n:3d} mean={m:.6g}\n")
        if "SQ_INSTS_VALU" in means[k]:
            f.write(f"{'lane-instructions per base':28s} {means[k]['SQ_INSTS_VALU'][0] * 64 / BASES_OF[k]:.2f}\n")
pmc = {"provenance": prov, "kernels": {k: dict(kernel=kernels[k], bases_per_launch=BASES_OF[k], **{c: m for c, (m, n) in means[k].items()}) for k in kernels}}
json.dump(pmc, open(os.path.join(dst, f"{RND}_pmc.json"), "w"), indent=1)

hbm = lambda m: (2 * m["FETCH_SIZE"][0] + m["WRITE_SIZE"][0]) * 1024  # FETCH_SIZE doubled: gfx950 tallies a wide coalesced read stream at 1/2
traffic = {"provenance": prov,
           "method": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; --kernel-trace only), one lane so that every counter belongs to one kernel; "
                     "FETCH_SIZE doubled (gfx950 reports 1/2 of a wide coalesced read stream, MI355X_MICROARCH.md HBM section); mean over the dispatches; "
                     "bench.py scales bytes per base by its own bases per launch"}
for cfg, (kc, ke) in {"c3": ("c3_count", "c3_emit"), "c4": ("c4_count", "c4_emit"), "c5": ("c5_count", "c5_emit"), "c2": ("c2_kmer", None)}.items():
    if "FETCH_SIZE" not in means[kc]:
        continue
    traffic[cfg] = {"count_kernel": {"kernel": kernels[kc], "FETCH_SIZE_KiB": means[kc]["FETCH_SIZE"][0], "WRITE_SIZE_KiB": means[kc]["WRITE_SIZE"][0],
                                     "hbm_bytes_per_base": hbm(means[kc]) / BASES_OF[kc]}}
    if ke and "FETCH_SIZE" in means[ke]:
        traffic[cfg]["emit_kernel"] = {"kernel": kernels[ke], "FETCH_SIZE_KiB": means[ke]["FETCH_SIZE"][0], "WRITE_SIZE_KiB": means[ke]["WRITE_SIZE"][0],
                                       "hbm_bytes_per_base": hbm(means[ke]) / BASES_OF[ke]}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({k: round(means[k]["SQ_INSTS_VALU"][0] * 64 / BASES_OF[k], 2) for k in kernels if "SQ_INSTS_VALU" in means[k]}))
