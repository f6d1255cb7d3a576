#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects what profiles/ holds for this round.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (two lanes) and of --lanes 1
#   2. PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE, SQ counters,
#      with the DEFAULT bench command's workloads (C3 at 50 Gbp, the others at their stated sizes) over ONE lane, so that every
#      counter belongs to exactly one kernel; one timed step (the counters do not need repetitions)
#   3. the VALU issue-cost table (tools/ubench_valu.hip)
# Before sending (in the build container, which has hipcc but no GPU; tools/_build/ travels with the snapshot):
#   mkdir -p tools/_build && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibiolib_amd/csrc tools/ubench_valu.hip -o tools/_build/ubench_valu
# Outputs under gpurun_out/prof/; tools/summarise_profiles.py (in the build container, where git is) turns them into profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
(hostname; date -u +%Y-%m-%dT%H:%M:%SZ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI3" ) > $OUT/box.txt 2>&1
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_sources_digest())" > $OUT/sources.sha256 || exit 1  # what was measured
cd /tmp && export TMPDIR=/tmp
$ROOT/tools/_build/ubench_valu --json $OUT/ubench_valu.json > $OUT/ubench_valu.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes2 -o s -- python3 $ROOT/bench.py --no-cpu-baseline --no-next-rows --no-h2d > $OUT/stats_lanes2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes1 -o s -- python3 $ROOT/bench.py --lanes 1 --no-cpu-baseline --no-next-rows --no-h2d > $OUT/stats_lanes1.log 2>&1 || exit 1
PMC_CMD="python3 $ROOT/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-next-rows --no-h2d --no-other-shapes"
echo "$PMC_CMD" > $OUT/pmc_command.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $PMC_CMD > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -o p -- $PMC_CMD > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -o p -- $PMC_CMD > $OUT/pmc_sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -o p -- $PMC_CMD > $OUT/pmc_sq3.log 2>&1 || exit 1
# 4. power / clock sensors beside the headline scan, two lanes and one (is the shader clock held by the card's power management?), and the
#    shape sweep off the BASELINE shapes
cd $ROOT
python3 tools/power_probe.py $OUT/power_lanes2.json -- python3 bench.py --steps 40 --no-cpu-baseline --no-other-configs --no-next-rows --no-h2d > $OUT/power_lanes2.log 2>&1 || exit 1
python3 tools/power_probe.py $OUT/power_lanes1.json -- python3 bench.py --lanes 1 --steps 40 --no-cpu-baseline --no-other-configs --no-next-rows --no-h2d > $OUT/power_lanes1.log 2>&1 || exit 1
python3 tests/perf/shape_sweep.py 3 > $OUT/shape_sweep.json 2> $OUT/shape_sweep.err || exit 1
find $OUT -name "*.csv" | head -40
