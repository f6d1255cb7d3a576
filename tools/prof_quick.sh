#!/bin/bash
# Quick profile of one lane of the C3 scan on the GPU box: kernel stats + three SQ counter passes (separate runs).
# (add `--opt position_tiled=1` to CMD to profile the position-tiled kernels instead)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/prof1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --gbp 6 --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -o p -- $CMD > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -o p -- $CMD > $OUT/pmc_sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq3 -o p -- $CMD > $OUT/pmc_sq3.log 2>&1 || exit 1
(for k in scan_count scan_emit; do echo "== $k"; python3 $ROOT/tools/pmc_summary.py $k $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3; done) > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
find $OUT/stats -name "*kernel_stats.csv" | xargs grep -h "scan_count\|scan_emit" | head
