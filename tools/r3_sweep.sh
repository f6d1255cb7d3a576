#!/bin/bash
# sweep of the pass-2 LDS padding (two-lane residency cap) on the C3 bench line: bash tools/r3_sweep.sh TAG v1 v2 ...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/sweep_$TAG; mkdir -p $OUT
for r in 1 2; do
for v in "$@"; do
  timeout -k 10 300 python bench.py --opt emit_lds_bytes=$v --no-cpu-baseline --no-other-configs --no-h2d --steps 6 > $OUT/l$v.$r.json 2> $OUT/l$v.$r.err || { tail -3 $OUT/l$v.$r.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/l$v.$r.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("emit_lds $v", "value", d["value"], "median", d["median_value"], "kernel_ms", r["avg_kernel_ms"], "clock", r["valu"].get("shader_clock_GHz"), flush=True)
PY
done
done
