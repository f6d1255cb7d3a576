#!/usr/bin/env python3
"""Per-kernel resources from hipcc -S output: python tools/kres.py file.s [name-regex]"""
import re, sys
txt = open(sys.argv[1]).read()
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for block in txt.split("  - .agpr_count:")[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
    name = f.get("name", "?")
    if rx and not rx.search(name):
        continue
    print(f"{name[:110]:110s} vgpr {f.get('vgpr_count')} sgpr {f.get('sgpr_count')} lds {f.get('group_segment_fixed_size')} scratch {f.get('private_segment_fixed_size')}")
