#!/usr/bin/env python3
"""VALU / SALU / LDS / VMEM instructions between the BL_MARK comments of a kernel (a -DBL_MARKS build):
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DBL_MARKS [-DBL_CENSUS_HOT] biolib_amd/csrc/bl_kernels.hip -o k.s
   python tools/phase_census.py k.s 'scan_count_kernelILi2ELi21ELi11ELi1ELb1'
The scheduler moves independent instructions across the markers, so the split is approximate; the total is exact."""
import collections, re, sys
rx = re.compile(sys.argv[2])
inside, phase = False, "(before)"
acc = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        inside = bool(rx.search(m.group(1)))
        phase = "(before)"
        continue
    if not inside:
        continue
    t = line.strip()
    if "BL_MARK" in t:
        phase = "after " + t.split("BL_MARK")[1].strip()
        continue
    if t.startswith("s_endpgm"):
        inside = False
        continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if re.match(r"(global|buffer|flat|scratch)_", op) else None
    if cls:
        acc.setdefault(phase, collections.Counter())[cls] += 1
tot = collections.Counter()
for ph, c in acc.items():
    print(f"{ph:24s} valu {c['valu']:5d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}")
    tot.update(c)
print(f"{'total':24s} valu {tot['valu']:5d} salu {tot['salu']:4d} lds {tot['lds']:3d} vmem {tot['vmem']:3d}")
