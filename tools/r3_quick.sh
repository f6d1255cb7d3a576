#!/bin/bash
# Quick iteration loop on the GPU box (through gpurun): parity tests of the scan, then the C3 bench line with two lanes and
# with one, then one PMC pass for the executed instruction count.   bash tools/r3_quick.sh TAG [pytest files...]
set -o pipefail
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/$TAG
rm -rf $OUT && mkdir -p $OUT
TESTS=${@:-tests/test_gpu_parity.py}
timeout -k 10 600 python -m pytest $TESTS -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 3 > $OUT/bench_l2.json 2> $OUT/bench_l2.err || { tail -5 $OUT/bench_l2.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 2 --gbp 12 --lanes 1 > $OUT/bench_l1.json 2> $OUT/bench_l1.err || { tail -5 $OUT/bench_l1.err; exit 1; }
python - <<PY
import json
for f in ("bench_l2", "bench_l1"):
    d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], "clock", r["valu"].get("shader_clock_GHz"))
PY
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --gbp 6 --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -o p -- $CMD > $OUT/pmc_sq1.log 2>&1 || exit 1
(for k in scan_count scan_emit; do echo "== $k"; python3 $ROOT/tools/pmc_summary.py $k $OUT/pmc_sq1; done) > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT/stats -name "*kernel_stats.csv" | xargs grep -h "scan_count\|scan_emit" | cut -c1-200
