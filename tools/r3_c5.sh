#!/bin/bash
# C5 (syncmer) iteration loop on the GPU box: parity tests, then the other-configs timing for each library variant.
#   bash tools/r3_c5.sh TAG NAME...   (NAME under biolib_amd/lib/ab/, or HEAD)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/c5_$TAG; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }; fi
[ -f $OUT/tests.log ] && tail -2 $OUT/tests.log
for v in "$@"; do
  if [ $v = HEAD ]; then unset BIOLIB_AMD_LIB; else export BIOLIB_AMD_LIB=$ROOT/biolib_amd/lib/ab/$v.so; fi
  timeout -k 10 300 python tests/perf/config_bench.py 12 2 > $OUT/cfg_$v.json 2> $OUT/cfg_$v.err || { tail -5 $OUT/cfg_$v.err; exit 1; }
  echo "$v $(tail -1 $OUT/cfg_$v.json)"
done
