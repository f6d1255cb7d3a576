#!/usr/bin/env python3
"""The VALU ceiling of bench.py, made reproducible: profiles/valu_model.json.

For every kernel of the BASELINE configurations:
  1. a static census of the HOT path's ISA (the kernels compiled with -DBL_CENSUS_HOT, which removes the branches clean data
     never takes: hash-prefix ties, breaks, ragged batch ends) — tools/isa_census.py;
  2. the dynamic count of wave-level VALU instructions from the PMC passes (SQ_INSTS_VALU, SQ_WAVES: profiles/rNN_pmc.json);
     the census is scaled to it (loops make the dynamic count a few percent larger than the static one: `scale`);
  3. the issue cost in shader cycles of every opcode, measured by tools/ubench_valu.hip at 8 waves per SIMD with nothing
     else running (profiles/rNN_ubench_valu.json; opcodes the table does not hold are priced at the cheapest class, so
     the total is a LOWER bound on the cycles the instruction stream needs).
cycles_per_base = sum(count x cycles) x scale x waves / bases: the SIMD cycles a base costs at peak issue.  bench.py divides
(cycles_per_base x bases) by (1024 SIMDs x the shader clock measured during its timed region x the time taken): the fraction of
the issue ceiling reached, good to a few percent (per-opcode rates measured in isolation; a real mix can pair slightly better: 1.01 seen).
Usage: python tools/valu_model.py [rNN]   (in the build container: hipcc cross-compiles, no GPU needed)"""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_census

RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
ub = json.load(open(os.path.join(ROOT, "profiles", f"{RND}_ubench_valu.json")))
pmc = json.load(open(os.path.join(ROOT, "profiles", f"{RND}_pmc.json")))

# issue cycles per opcode: host-timed throughput at 8 waves per SIMD (the most favourable case)
cyc = {name: v["w8"]["wall_cycles"] for name, v in ub["ops"].items() if not v["compound"]}
pair = ub["ops"]["v_cmp+v_cndmask (pair)"]["w8"]["wall_cycles"]
cyc["v_cndmask_b32"] = pair - cyc["v_cmp_lt_u32"]  # v_cndmask behind the compare that feeds it
cheapest = min(cyc.values())
alias = {"v_mov_b32_dpp": "v_mov_b32_dpp wave_shl", "v_add_u32_dpp": "v_add_u32_dpp wave_shl", "v_subrev_u32": "v_sub_u32",
         "v_cmp_gt_u32": "v_cmp_lt_u32", "v_cmp_ge_u32": "v_cmp_lt_u32", "v_cmp_le_u32": "v_cmp_lt_u32", "v_cmp_eq_u32": "v_cmp_lt_u32", "v_cmp_ne_u32": "v_cmp_lt_u32",
         "v_cmp_gt_i32": "v_cmp_lt_u32", "v_cmp_lt_i32": "v_cmp_lt_u32", "v_cmp_gt_u64": "v_cmp_lt_u64", "v_cmp_ge_u64": "v_cmp_lt_u64", "v_cmp_le_u64": "v_cmp_lt_u64",
         "v_cmp_eq_u64": "v_cmp_lt_u64", "v_cmp_ne_u64": "v_cmp_lt_u64", "v_cmp_gt_i64": "v_cmp_lt_u64", "v_cmp_lt_i64": "v_cmp_lt_u64", "v_cmp_ge_i64": "v_cmp_lt_u64",
         "v_cmp_le_i64": "v_cmp_lt_u64", "v_max_u32": "v_min_u32", "v_min_i32": "v_min_u32", "v_max_i32": "v_min_u32", "v_max3_u32": "v_min3_u32"}


def price(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    for name in (op, base, alias.get(base, "")):
        if name in cyc:
            return cyc[name], True
    return cheapest, False


with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBL_CENSUS_HOT", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "biolib_amd", "csrc", "bl_kernels.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    mangled = {"c3_count": r"scan_count_frl_kernelILi0ELi11ELi15ELi31ELi150ELi1ELb1E", "c3_emit": r"scan_emit_kernelILi0ELi31ELi1ELi1E", "c2_kmer": r"kmer_kernel",
               "c4_count": r"scan_count_kernelILi1ELi17ELi15ELi1ELi0E", "c4_emit": r"scan_emit_kernelILi1ELi0ELin1ELin1E", "c5_count": r"scan_count_kernelILi2ELi21ELi11ELi1ELi1E",
               "c5_emit": r"scan_emit_kernelILi2ELi0ELin1ELin1E",
               # the tiles pass 1 could not decide, counted again (inside the region bench.py's kernel events bracket): a few launches' worth of waves
               "c3_redo": r"scan_redo_frl_kernelILi0ELi11ELi15ELi31ELi150ELi1E", "c5_redo": r"scan_redo_kernelILi2ELi21ELi11ELi1E"}
    out = {"provenance": {"pmc": pmc["provenance"], "ubench": f"profiles/{RND}_ubench_valu.json (same collection run)", "census": "hipcc -DBL_CENSUS_HOT -S of biolib_amd/csrc/bl_kernels.hip, tools/isa_census.py",
                          "tool": "tools/valu_model.py"},
           "issue_cycles": {"cheapest_class (v_add/v_sub/v_xor/v_and/v_or/v_mov/v_lshrrev)": round(cheapest, 3), "v_mul_lo_u32": round(cyc["v_mul_lo_u32"], 3),
                            "v_mad_u64_u32": round(cyc["v_mad_u64_u32"], 3), "v_alignbit_b32": round(cyc["v_alignbit_b32"], 3), "v_cndmask_b32": round(cyc["v_cndmask_b32"], 3),
                            "v_lshl_add_u64": round(cyc["v_lshl_add_u64"], 3), "note": "shader cycles a SIMD spends per wave64 instruction, 8 waves per SIMD, host-timed"},
           "kernels": {}}
    for key, pat in mangled.items():
        name, ops = isa_census.census(asm, pat)
        assert name, f"kernel {pat} not found"
        valu = {op: n for op, n in ops.items() if op.startswith("v_")}
        static = sum(valu.values())
        cycles = sum(n * price(op)[0] for op, n in valu.items())
        unknown = {op: n for op, n in valu.items() if not price(op)[1]}
        k = pmc["kernels"][key]
        entry = {"kernel": name, "static_valu": static, "static_cycles": round(cycles, 1), "unpriced_opcodes_at_cheapest": unknown,
                 "class_counts": {"cheapest": sum(n for op, n in valu.items() if price(op)[0] <= cheapest * 1.15), "other": sum(n for op, n in valu.items() if price(op)[0] > cheapest * 1.15)}}
        if "SQ_INSTS_VALU" in k and "SQ_WAVES" in k:
            dyn_per_wave = k["SQ_INSTS_VALU"] / k["SQ_WAVES"]
            entry.update(dynamic_valu_per_wave=round(dyn_per_wave, 1), scale=round(dyn_per_wave / static, 4),
                         valu_per_base=k["SQ_INSTS_VALU"] / k["bases_per_launch"],
                         cycles_per_base=cycles / static * k["SQ_INSTS_VALU"] / k["bases_per_launch"])
        out["kernels"][key] = entry
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_model.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {x: e.get(x) for x in ("static_valu", "dynamic_valu_per_wave", "scale", "cycles_per_base")}, "lane-instr/base", round(e.get("valu_per_base", 0) * 64, 1))
