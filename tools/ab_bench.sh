#!/bin/bash
# A/B on the GPU box: the C3 bench line with each library variant in turn, ROUNDS times over; prints value / clock per run.
#   bash tools/ab_bench.sh TAG ROUNDS "bench args" NAME...     (NAME = a file under biolib_amd/lib/ab/, or HEAD = the tree's own)
TAG=$1; ROUNDS=$2; ARGS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/ab_$TAG; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ $v = HEAD ]; then unset BIOLIB_AMD_LIB; else export BIOLIB_AMD_LIB=$ROOT/biolib_amd/lib/ab/$v.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-h2d $ARGS > $OUT/${v}_$r.json 2> $OUT/${v}_$r.err || { tail -3 $OUT/${v}_$r.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$OUT/${v}_$r.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("$v", "round $r", "value", d["value"], "kernel_ms", r["avg_kernel_ms"], "clock", r["valu"].get("shader_clock_GHz"), flush=True)
PY
  done
done
