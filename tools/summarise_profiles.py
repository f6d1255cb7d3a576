#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (tools/collect_profiles.sh) into the files under profiles/ that bench.py and DESIGN.md cite:
r01_kernel_stats.csv (+ _lanes1), r01_pmc_summary.txt, traffic.json."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "stats_lanes2", "s_kernel_stats.csv"), os.path.join(dst, "r01_kernel_stats.csv"))
shutil.copy(os.path.join(src, "stats_lanes1", "s_kernel_stats.csv"), os.path.join(dst, "r01_kernel_stats_lanes1.csv"))

BASES = 1_500_000_000  # --gbp 6 -> 4 launches of 1.5e9 bases
kernels = {"scan_count": "bl::scan_count_kernel<0, 11, 31, 1>", "scan_emit": "bl::scan_emit_kernel<0>"}
means = {k: {} for k in kernels}
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        for k, name in kernels.items():
            if name in row["Kernel_Name"]:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in acc:
        for c, v in acc[k].items():
            means[k][c] = (sum(v) / len(v), len(v))
with open(os.path.join(dst, "r01_pmc_summary.txt"), "w") as f:
    for k in kernels:
        f.write(f"== {k} (mean per launch of 1.5e9 bases over the dispatches of `bench.py --gbp 6 --steps 1 --warmup 0 --lanes 1`; "
                "FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE to be doubled on gfx950, MI355X_MICROARCH.md)\n")
        for c, (m, n) in sorted(means[k].items()):
            f.write(f"{c:28s} n={n:3d} mean={m:.6g}\n")
cnt, emi = means["scan_count"], means["scan_emit"]
hbm = lambda m: int((2 * m["FETCH_SIZE"][0] + m["WRITE_SIZE"][0]) * 1024)
traffic = {
    "kernel": kernels["scan_count"],
    "measured_bases_per_launch": BASES,
    "FETCH_SIZE_KiB": cnt["FETCH_SIZE"][0], "WRITE_SIZE_KiB": cnt["WRITE_SIZE"][0],
    "hbm_bytes_per_launch_measured": hbm(cnt), "hbm_bytes_per_base": hbm(cnt) / BASES,
    "valu_wave_instr_per_base": cnt["SQ_INSTS_VALU"][0] / BASES,
    "emit_valu_wave_instr_per_base": emi["SQ_INSTS_VALU"][0] / BASES,
    "method": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ counters; --kernel-trace only) over `python3 bench.py --gbp 6 --steps 1 "
              "--warmup 0 --lanes 1 --no-cpu-baseline` (4 launches of 1.5e9 bases, one lane so that every counter belongs to one kernel); FETCH_SIZE "
              "doubled: gfx950 reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM); mean over the dispatches; bench.py scales by its "
              "own bases per launch.  tools/collect_profiles.sh + tools/summarise_profiles.py",
    "breakdown": "reads: 1 B/base ASCII + 1/8 B/base sequence-start bits (+3 % halo); writes: 2-byte record-list entries, 8-byte tile counts, "
                 "0.26 B/base packed 2-bit codes handed to the emit pass",
    "valu_note": "SQ_INSTS_VALU: wave-level VALU instructions per launch / bases per launch; x64 = lane-instructions per base",
    "emit_kernel": {"kernel": kernels["scan_emit"], "FETCH_SIZE_KiB": emi["FETCH_SIZE"][0], "WRITE_SIZE_KiB": emi["WRITE_SIZE"][0],
                    "hbm_bytes_per_launch_measured": hbm(emi)},
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("hbm_bytes_per_base", "valu_wave_instr_per_base", "emit_valu_wave_instr_per_base")}))
