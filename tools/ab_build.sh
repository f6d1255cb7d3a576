#!/bin/bash
# Build the library of another revision beside the working tree's, for A/B runs on ONE GPU box (boxes differ by several
# percent in the clock they hold under power):   tools/ab_build.sh NAME REV   ->  biolib_amd/lib/ab/NAME.so
set -e
NAME=$1; REV=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/ab_wt_$NAME
rm -rf $WT; git -C $ROOT worktree prune; git -C $ROOT worktree add -f --detach $WT $REV > /dev/null 2>&1
make -s -j8 -C $WT/biolib_amd/csrc > /dev/null 2>&1
mkdir -p $ROOT/biolib_amd/lib/ab
cp $WT/biolib_amd/lib/libbiolib_amd.so $ROOT/biolib_amd/lib/ab/$NAME.so
git -C $ROOT worktree remove --force $WT
echo "built biolib_amd/lib/ab/$NAME.so from $(git -C $ROOT rev-parse --short $REV)"
