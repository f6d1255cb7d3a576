#!/bin/bash
# pass 1 alone (one lane) with and without BL_NO_APPROX, and the executed VALU instructions of each
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/approx2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in approx exact approx exact; do
  if [ $v = exact ]; then OPT="--opt exact_windows=1"; else OPT=""; fi
  timeout -k 10 300 python3 $ROOT/bench.py $OPT --no-cpu-baseline --no-next-rows --no-h2d --no-other-configs --lanes 1 --gbp 12 --steps 3 > $OUT/l1_$v.json 2> $OUT/l1_$v.err || { tail -5 $OUT/l1_$v.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("$OUT/l1_$v.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$v", "value", d["value"], "kernel ms", r.get("avg_kernel_ms"), "clock", r["valu"].get("shader_clock_GHz"))
PY
done
for v in approx exact; do
  if [ $v = exact ]; then OPT="--opt exact_windows=1"; else OPT=""; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_$v -o p -- python3 $ROOT/bench.py $OPT --steps 1 --warmup 0 --lanes 1 --gbp 6 --no-cpu-baseline --no-next-rows --no-h2d --no-other-configs > $OUT/pmc_$v.log 2>&1 || { tail -5 $OUT/pmc_$v.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$v/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if row["Counter_Name"] == "SQ_INSTS_VALU":
        k = row["Kernel_Name"][:60]
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (s, n) in acc.items():
    if "scan_" in k: print("$v", k, "launches", n, "VALU wave-instr per launch", s / n)
PY
done
