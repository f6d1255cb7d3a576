#!/usr/bin/env python3
"""Static census of a kernel's ISA (gfx950): instructions by class and opcode.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only biolib_amd/csrc/bl_kernels.hip -o /tmp/k.s
  python tools/isa_census.py /tmp/k.s 'scan_count_frl_kernelILi0ELi11ELi15' [--json out.json]

The scan kernels are almost straight-line (everything is unrolled), so the static count is close to what one thread
executes per tile; tools/valu_model.py weights it with the measured issue cycles of tools/ubench_valu.hip."""
import collections
import json
import re
import sys


def census(path, pattern):
    rx = re.compile(pattern)
    ops = collections.Counter()
    name = None
    inside = False
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = bool(rx.search(m.group(1)))
            if inside:
                name = m.group(1)
            continue
        if not inside:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            ops["s_endpgm"] += 1
            inside = False
            continue
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        op = t.split()[0]
        if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
            ops[op] += 1
    return name, ops


def classes(ops):
    c = collections.Counter()
    for op, n in ops.items():
        if op.startswith("v_"):
            c["valu"] += n
        elif op.startswith("s_"):
            c["salu"] += n
        elif op.startswith("ds_"):
            c["lds"] += n
        else:
            c["vmem"] += n
    return c


if __name__ == "__main__":
    name, ops = census(sys.argv[1], sys.argv[2])
    cl = classes(ops)
    print(name)
    print(dict(cl))
    for op, n in ops.most_common():
        if op.startswith("v_") or op.startswith("ds_") or op.startswith("global_"):
            print(f"  {op:28s} {n}")
    if "--json" in sys.argv:
        json.dump({"kernel": name, "classes": dict(cl), "ops": dict(ops)}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
