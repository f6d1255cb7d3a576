#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: GPU test suite, then the default bench line.
#   bash tools/gpu_check.sh TAG [bench args...]   -> gpurun_out/r2/{gputest,bench}_TAG.*
TAG=${1:-x}; shift
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/gputest_$TAG.log 2>&1
RC=$?
tail -3 gpurun_out/r2/gputest_$TAG.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 400 python bench.py "$@" > gpurun_out/r2/bench_$TAG.json 2> gpurun_out/r2/bench_$TAG.err || { tail -5 gpurun_out/r2/bench_$TAG.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/r2/bench_$TAG.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "valu", r.get("valu", {}).get("frac"),
      {k: v["value"] for k, v in d.get("other_configs", {}).items() if isinstance(v, dict)}, {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "cores", "single_thread_value", "gpu_result_bit_identical_on_sample")})
PY
