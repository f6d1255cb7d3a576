#!/usr/bin/env python3
"""gpurun_out/prof_inflate (tools/prof_inflate.sh) -> profiles/rNN_inflate_pmc.txt: counters of the device inflate kernel per
member and per symbol.  Symbols per member of the benchmark text were counted once with the host build of the decoder
(1,750 literals + 5,687 matches per 65,280-byte member of the NovaSeq-like FASTQ at zlib level 6)."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_inflate")
acc = collections.defaultdict(list)
dur = []
for d in ("a", "b"):
    for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        if "inflate_members" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(os.path.join(src, d, "p_kernel_trace.csv"))):
        if "inflate_members" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
case = bench["cases"]["binned/level6"]
members, symbols = case["members"], 1750 + 5687
lines = [f"# device inflate kernel, one launch of {members} members ({case['text_MB']} MB of FASTQ text, zlib level 6); kernel {sum(dur) / len(dur):.3f} ms under the counters",
         f"# per member / per symbol ({symbols} symbols per member); SQ_WAVE_CYCLES and SQ_*_INST_ANY count in units of 4 cycles"]
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    lines.append(f"{k:22s} total {v:14.0f}   per member {v / members:12.0f}   per symbol {v / members / symbols:8.1f}")
open(os.path.join(ROOT, "profiles", f"{tag}_inflate_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
