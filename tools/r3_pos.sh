#!/bin/bash
# the position-tiled minimizer kernels (BL_NO_FRL=1 keeps the C3 workload on them): parity tests, then the bench line per library variant
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/pos_$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_contig_split.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for r in 1 2; do for v in "$@"; do
  if [ $v = HEAD ]; then unset BIOLIB_AMD_LIB; else export BIOLIB_AMD_LIB=$ROOT/biolib_amd/lib/ab/$v.so; fi
  timeout -k 10 300 python bench.py --opt position_tiled=1 --no-cpu-baseline --no-other-configs --no-h2d --steps 4 --gbp 25 > $OUT/${v}_$r.json 2> $OUT/${v}_$r.err || { tail -3 $OUT/${v}_$r.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/${v}_$r.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("$v", "value", d["value"], "kernel_ms", r["avg_kernel_ms"], flush=True)
PY
done; done
