#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: instruction and cycle counters of the device inflate kernel on one
# resident set of members (tests/perf/inflate_bench.py N quick = one launch of 1,0xx members of NovaSeq-like FASTQ, zlib level 6).
# Two PMC passes, --kernel-trace only (MI355X_MICROARCH.md).  tools/summarise_inflate.py turns them into profiles/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_inflate; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/tests/perf/inflate_bench.py 186000 quick"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1 || exit 1
grep '^{' $OUT/b.log > $OUT/bench.json
echo ok
