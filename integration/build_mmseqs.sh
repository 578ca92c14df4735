#!/bin/bash
# Builds two `mmseqs` binaries from the reference tree where it lies (never copied into this repository):
#
#   oracle/_ref/mmseqs_stock   the reference as shipped (CPU, AVX2).  The Rust block-aligner crate cannot be built here (no rustc):
#                              its C API is provided by oracle/ref_block_capi.cpp over the plain-C restatement of the block
#                              aligner (oracle/block_oracle.c) for the sequence-sequence calls, and by the generated stubs of
#                              oracle/gen_block_stub.py for the rest (profile queries then take the reference's own
#                              Smith-Waterman fallback).  MMGPU_BLOCK_STUB_ONLY=1: stubs for everything,
#                              as in rounds 1-2.
#   oracle/_ref/mmseqs_mmgpu   the same tree + integration/mmseqs_mmgpu.patch (10 files, every change under #ifdef HAVE_MMGPU),
#                              integration/*.cpp compiled in, linked against mmseqs2_amd/lib/libmmgpu.so.  ALWAYS with the
#                              do-nothing block-aligner stubs (round 4): nothing of oracle/*.c is on the product-side binary's link
#                              line - int16-range pairs get start / CIGAR from the device's block aligner (block_kernel.hip, blocks
#                              up to the crate's 4096 rows); a maintainer's build links the real Rust crate here instead
#
# Both are checkers / demonstrators of the drop-in (tests/test_mmseqs_dropin.py diffs their result DBs); they are git-ignored
# and travel to the GPU box with the snapshot.  Uses the reference's CMake files on a scratch copy.
#
#   REF=/root/reference  MMGPU_BUILD_DIR=/tmp/mmgpu_mmseqs_build  integration/build_mmseqs.sh [stock|stub|mmgpu|all]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(dirname "$HERE")"
REF="${REF:-/root/reference}"
# the scratch trees are keyed on what goes into them: a changed patch / block-aligner stand-in gets fresh trees, a half-patched
# tree of an interrupted run is never built from
KEY="$(cat "$HERE/mmseqs_mmgpu.patch" "$REPO/oracle/gen_block_stub.py" "$REPO/oracle/ref_block_capi.cpp" "$REPO/oracle/block_oracle.c" "$REPO/oracle/mm_oracle.h" | sha256sum | cut -c1-12)${MMGPU_BLOCK_STUB_ONLY:+_stub}"
WORK="${MMGPU_BUILD_DIR:-/tmp/mmgpu_mmseqs_build}_$KEY"
OUT="$REPO/oracle/_ref"
WHAT="${1:-all}"
JOBS="${JOBS:-$(nproc)}"
[ -d "$REF/src" ] || { echo "build_mmseqs: no reference tree at $REF" >&2; exit 2; }
mkdir -p "$WORK" "$OUT"

BLOCK_CAPI_FUNCS="$(sed -n 's/^BLOCK_CAPI_FUNCS := //p' "$REPO/oracle/Makefile")"
prepare_tree() {   # $1 = destination
    if [ ! -f "$1/.mmgpu_prepared" ]; then
        rm -rf "$1"
        cp -r "$REF" "$1"
        chmod -R u+w "$1"
        python3 - "$1" "${MMGPU_BLOCK_STUB_ONLY:-0}" <<'PY'
import sys
p = sys.argv[1] + '/CMakeLists.txt'
s = open(p).read()
start = s.index('set(ENV{CARGO_NET_OFFLINE} true)')
end = s.index('if (USE_SYSTEM_ZSTD)')
srcs = '${CMAKE_CURRENT_SOURCE_DIR}/block_stub.c' if sys.argv[2] == '1' else \
    '${CMAKE_CURRENT_SOURCE_DIR}/block_stub.c ${CMAKE_CURRENT_SOURCE_DIR}/ref_block_capi.cpp ${CMAKE_CURRENT_SOURCE_DIR}/block_oracle.c'
s = s[:start] + 'include_directories(lib/block-aligner/c)\nadd_library(block_aligner_c STATIC ' + srcs + ')\n' + s[end:]
open(p, 'w').write(s)
PY
        if [ -n "${MMGPU_BLOCK_STUB_ONLY:-}" ]; then
            python3 "$REPO/oracle/gen_block_stub.py" "$1" "$1/block_stub.c"
        else
            python3 "$REPO/oracle/gen_block_stub.py" "$1" "$1/block_stub.c" --skip "$BLOCK_CAPI_FUNCS"
            cp "$REPO/oracle/ref_block_capi.cpp" "$REPO/oracle/block_oracle.c" "$REPO/oracle/mm_oracle.h" "$1/"
        fi
        touch "$1/data/resources/K4000.crf"     # large blob absent from the checkout (.MISSING_LARGE_BLOBS)
        if [ -n "${2:-}" ]; then patch -d "$1" -p1 < "$2"; fi
        touch "$1/.mmgpu_prepared"
    fi
}

if [ "$WHAT" = stock ] || [ "$WHAT" = all ]; then
    prepare_tree "$WORK/ref_stock"
    cmake -S "$WORK/ref_stock" -B "$WORK/build_stock" -DHAVE_AVX2=1 -DCMAKE_BUILD_TYPE=Release -DHAVE_TESTS=0 -DHAVE_SHELLCHECK=0 > "$WORK/cmake_stock.log" 2>&1
    make -C "$WORK/build_stock" -j"$JOBS" mmseqs > "$WORK/make_stock.log" 2>&1 || { tail -30 "$WORK/make_stock.log"; exit 1; }
    cp "$WORK/build_stock/src/mmseqs" "$OUT/mmseqs_stock"
    echo "built $OUT/mmseqs_stock"
    # BASELINE.json configs[0]: the reference's example proteins as a sequence DB (createdb output, not the FASTA itself)
    # for the drop-in tests on the GPU box, where /root/reference does not exist
    rm -rf "$OUT/dropin_data" && mkdir -p "$OUT/dropin_data"
    "$OUT/mmseqs_stock" createdb "$REF/examples/QUERY.fasta" "$OUT/dropin_data/examples" -v 1
fi

# timing partner only (bench.py `modules`): the stock tree with do-nothing block-aligner stubs, as in rounds 1-2 - every int16-range
# hit then takes the reference's Smith-Waterman fallback, whose cost is close to the real (AVX2, Rust) crate's; the restated crate
# above is scalar C and slower
if [ "$WHAT" = stub ] || [ "$WHAT" = all ]; then
    if [ -z "${MMGPU_BLOCK_STUB_ONLY:-}" ]; then
        MMGPU_BLOCK_STUB_ONLY=1 prepare_tree "$WORK/ref_stock_stub"
        cmake -S "$WORK/ref_stock_stub" -B "$WORK/build_stock_stub" -DHAVE_AVX2=1 -DCMAKE_BUILD_TYPE=Release -DHAVE_TESTS=0 -DHAVE_SHELLCHECK=0 > "$WORK/cmake_stock_stub.log" 2>&1
        make -C "$WORK/build_stock_stub" -j"$JOBS" mmseqs > "$WORK/make_stock_stub.log" 2>&1 || { tail -30 "$WORK/make_stock_stub.log"; exit 1; }
        cp "$WORK/build_stock_stub/src/mmseqs" "$OUT/mmseqs_stock_stub"
        echo "built $OUT/mmseqs_stock_stub"
    fi
fi

if [ "$WHAT" = mmgpu ] || [ "$WHAT" = all ]; then
    LIB="${MMGPU_LIBRARY:-$REPO/mmseqs2_amd/lib/libmmgpu.so}"
    [ -f "$LIB" ] || { echo "build_mmseqs: $LIB missing (run make -C mmseqs2_amd/csrc first)" >&2; exit 2; }
    MMGPU_BLOCK_STUB_ONLY=1 prepare_tree "$WORK/ref_mmgpu_stubonly" "$HERE/mmseqs_mmgpu.patch"
    cmake -S "$WORK/ref_mmgpu_stubonly" -B "$WORK/build_mmgpu" -DHAVE_AVX2=1 -DCMAKE_BUILD_TYPE=Release -DHAVE_TESTS=0 -DHAVE_SHELLCHECK=0 \
        -DHAVE_MMGPU=1 -DMMGPU_DIR="$REPO" -DMMGPU_LIBRARY="$LIB" \
        -DCMAKE_EXE_LINKER_FLAGS="-Wl,-rpath,'\$ORIGIN/../../mmseqs2_amd/lib' -Wl,-rpath-link,/opt/rocm/lib" > "$WORK/cmake_mmgpu.log" 2>&1
    make -C "$WORK/build_mmgpu" -j"$JOBS" mmseqs > "$WORK/make_mmgpu.log" 2>&1 || { grep -B2 -A12 "error" "$WORK/make_mmgpu.log" | head -80; exit 1; }
    cp "$WORK/build_mmgpu/src/mmseqs" "$OUT/mmseqs_mmgpu"
    echo "built $OUT/mmseqs_mmgpu"
fi
