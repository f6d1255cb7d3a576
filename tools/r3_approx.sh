#!/bin/bash
# A/B on one box: the headline scan with pass 1 on murmur64_top (default) and on the hashes themselves (BL_NO_APPROX=1)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/approx
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python -m pytest ${@:-tests/test_gpu_parity.py tests/test_gpu_edges.py} -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for v in approx exact approx exact; do
  if [ $v = exact ]; then OPT="--opt exact_windows=1"; else OPT=""; fi
  timeout -k 10 300 python bench.py $OPT --no-cpu-baseline --no-next-rows --no-h2d --steps 5 > $OUT/bench_$v.json 2> $OUT/bench_$v.err || { tail -5 $OUT/bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
r = d["roofline"]
oc = d.get("other_configs", {})
print("$v", "value", d["value"], "median", d.get("median_value"), "kernel ms", r.get("avg_kernel_ms"), "clock", r["valu"].get("shader_clock_GHz"), {k: v.get("value") for k, v in oc.items() if isinstance(v, dict)})
PY
done
