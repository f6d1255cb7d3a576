#!/bin/bash
# executed VALU instructions of the scan kernels under tests/perf/config_bench.py, for each library variant:  bash tools/r3_pmc_ab.sh NAME...
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/pmc_ab; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = HEAD ]; then unset BIOLIB_AMD_LIB; else export BIOLIB_AMD_LIB=$ROOT/biolib_amd/lib/ab/$v.so; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/$v -o p -- python3 $ROOT/tests/perf/config_bench.py 3 1 > $OUT/$v.log 2>&1 || { tail -5 $OUT/$v.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$v/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if row["Counter_Name"] == "SQ_INSTS_VALU":
        k = row["Kernel_Name"][:58]
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (s, n) in sorted(acc.items()):
    if "scan_count" in k or "kmer_kernel" in k: print("$v", k, "launches", n, "wave-instr total", round(s))
PY
done
