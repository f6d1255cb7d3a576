#!/bin/bash
# The reader's CPU tests with the host side of ingest (bl_ingest.cpp: byte sources, the many-threaded gzip decoder, span
# cutting, sharded readers) built under AddressSanitizer.  The other objects of the library are linked as they are.
#   bash tools/asan_ingest.sh        (from the repo root, after `make -C biolib_amd/csrc`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/bl_asan
mkdir -p "$OUT"
HIPCC=/opt/rocm/bin/hipcc
cd "$ROOT/biolib_amd/csrc"
$HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -shared-libasan -Wno-option-ignored -c bl_ingest.cpp -o "$OUT/bl_ingest.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan $(ls _obj/*.o | grep -v bl_ingest) "$OUT/bl_ingest.o" -lz -lpthread -ldl -o "$OUT/libbiolib_amd_asan.so"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
# (tests that start sanitizer-built helper programs of their own are left out: two ASan runtimes do not mix)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 BIOLIB_AMD_LIB="$OUT/libbiolib_amd_asan.so" \
    python -m pytest tests/test_pgzip.py tests/test_ingest.py -x -q -m "not gpu" -p no:cacheprovider -k "not pieces and not crc32 and not fuzz and not ref"
