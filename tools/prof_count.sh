ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r2/cpmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/tests/perf/count_bench.py 1.5"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
echo ok
