#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: counters of the k-mer counter's table kernel (count_buckets_kernel) on
# the 1.5-Gbp chain of tests/perf/count_bench.py.  Separate PMC passes, --kernel-trace only (MI355X_MICROARCH.md).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_count; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/tests/perf/count_bench.py"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, os
acc = collections.defaultdict(list); dur = []
for d in "abc":
    f = "$OUT/%s/p_counter_collection.csv" % d
    if not os.path.exists(f): continue
    for r in csv.DictReader(open(f)):
        if "count_buckets_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open("$OUT/%s/p_kernel_trace.csv" % d)):
        if "count_buckets_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
lines = ["# count_buckets_kernel on 1.5 Gbp (4,194,304 buckets, 1.2e9 k-mers): mean per launch; kernel %.3f ms under the counters (%d launches)" % (sum(dur) / max(len(dur), 1), len(dur))]
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    lines.append("%-22s %16.0f   per bucket %10.1f   per k-mer %8.2f" % (k, v, v / 4194304, v / 1.2e9))
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
