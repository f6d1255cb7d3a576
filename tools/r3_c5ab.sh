#!/bin/bash
# C5 (closed syncmers) A/B on one box: the tree's library against biolib_amd/lib/ab/closed_exact.so
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/c5ab
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for v in HEAD closed_exact HEAD closed_exact; do
  if [ $v = HEAD ]; then unset BIOLIB_AMD_LIB; else export BIOLIB_AMD_LIB=$ROOT/biolib_amd/lib/ab/$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-next-rows --no-h2d --steps 3 > $OUT/bench_$v.json 2> $OUT/bench_$v.err || { tail -5 $OUT/bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
oc = d.get("other_configs", {})
print("$v", "C3", d["value"], {k: (v.get("value"), v.get("roofline", {}).get("avg_kernel_ms")) for k, v in oc.items() if isinstance(v, dict)})
PY
done
