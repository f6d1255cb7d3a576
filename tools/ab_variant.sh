#!/bin/bash
# Build the working tree's library with extra -D switches into biolib_amd/lib/ab/NAME.so:  tools/ab_variant.sh NAME "-DX=1 ..."
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/ab_var_$NAME
rm -rf $WT; mkdir -p $WT/biolib_amd $WT/include
cp -r $ROOT/biolib_amd/csrc $WT/biolib_amd/; rm -rf $WT/biolib_amd/csrc/_obj; cp -r $ROOT/include/* $WT/include/
make -s -j8 -C $WT/biolib_amd/csrc EXTRA="$EXTRA" > /dev/null 2>&1
mkdir -p $ROOT/biolib_amd/lib/ab
cp $WT/biolib_amd/lib/libbiolib_amd.so $ROOT/biolib_amd/lib/ab/$NAME.so
rm -rf $WT
echo "built biolib_amd/lib/ab/$NAME.so with $EXTRA"
