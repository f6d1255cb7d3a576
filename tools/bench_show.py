#!/usr/bin/env python3
"""Prints the figures of a bench.py JSON line that are compared from run to run:  python tools/bench_show.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("C3", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["avg_kernel_ms"], "frac", r["frac"], "clock", r.get("valu", {}).get("shader_clock_GHz"),
      {k: d[k] for k in ("median_value", "best_value", "h2d_inclusive_Gbps") if k in d})
for k, v in d.get("other_configs", {}).items():
    if isinstance(v, dict):
        print(k, v["value"], "kernel_ms", v["avg_kernel_ms"], "clock", v["roofline"].get("valu", {}).get("shader_clock_GHz"))
for k in ("next_rows", "cpu_baseline"):
    if k in d:
        print(k, json.dumps(d[k])[:900])
