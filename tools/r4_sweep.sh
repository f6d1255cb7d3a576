#!/bin/bash
# On the GPU box: the C3 bench line over context options and library variants, ROUNDS times, interleaved (boxes differ, clocks drift).
#   bash tools/r4_sweep.sh TAG ROUNDS "bench args" CASE...     CASE = LIB[:opt=value[,opt=value...]]   (LIB = HEAD or a name under biolib_amd/lib/ab/)
TAG=$1; ROUNDS=$2; ARGS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/sweep_$TAG; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  for c in "$@"; do
    lib=${c%%:*}; o=""; [ "$c" != "$lib" ] && o=${c#*:}
    if [ $lib = HEAD ]; then unset BIOLIB_AMD_LIB; else export BIOLIB_AMD_LIB=$ROOT/biolib_amd/lib/ab/$lib.so; fi
    OPTS=""; for kv in ${o//,/ }; do OPTS="$OPTS --opt $kv"; done
    f=$OUT/$(echo $c | tr ':=,' '___')_$r
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-h2d $ARGS $OPTS > $f.json 2> $f.err || { tail -3 $f.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$f.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("$c", "round $r", "value", d["value"], "median", d["median_value"], "kernel_ms", r["avg_kernel_ms"], "clock", r["valu"].get("shader_clock_GHz"), flush=True)
PY
  done
done
