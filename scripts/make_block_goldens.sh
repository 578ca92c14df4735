#!/bin/bash
# ONE-COMMAND RECIPE FOR A RUST-EQUIPPED BOX (rustc >= 1.68, cargo; no network needed: the crate has no dependencies).
#
# Records tests/golden/block_vectors.npz: start position, identities and backtrace of int16-range hits exactly as a Rust-linked
# build of the reference computes them (SmithWaterman::alignStartPosBacktraceBlock, src/alignment/StripedSmithWaterman.cpp:943-1127
# -> lib/block-aligner 0.4.0 with the reference's own feature choice for -DHAVE_AVX2=1 builds, CMakeLists.txt:222-224: simd_avx2).
# tests/test_block_oracle.py::test_recorded_rust_vectors then pins oracle/block_oracle.c (and, through it, the device path of
# row a15) against that file; until it exists the test is skipped with "PARITY UNPINNED".
#
#   REF=/path/to/MMseqs2 scripts/make_block_goldens.sh
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(dirname "$HERE")"
REF="${REF:-/root/reference}"
command -v cargo >/dev/null || { echo "make_block_goldens: cargo not found - this recipe needs a Rust toolchain" >&2; exit 2; }
WORK="${WORK:-/tmp/mmgpu_block_goldens}"
mkdir -p "$WORK"
# 1. the crate as the reference builds it (corrosion_import_crate(MANIFEST_PATH lib/block-aligner/c/Cargo.toml FEATURES simd_avx2
#    CRATE_TYPES staticlib), CMakeLists.txt:246-250); the tree may be read-only, so build in a copy
rm -rf "$WORK/block-aligner" && cp -r "$REF/lib/block-aligner" "$WORK/block-aligner" && chmod -R u+w "$WORK/block-aligner"
( cd "$WORK/block-aligner/c" && CARGO_NET_OFFLINE=true RUSTFLAGS="-C target-feature=+avx2" cargo build --release --offline --features simd_avx2 )
LIB="$WORK/block-aligner/c/target/release/libblock_aligner_c.a"
[ -f "$LIB" ] || { echo "make_block_goldens: $LIB was not produced" >&2; exit 1; }
# 2. the reference classes around it: oracle/_ref/libmmref_rust.so = libmmref.so with the REAL crate instead of the stubs
make -C "$REPO/oracle" REF="$REF" BLOCK_ALIGNER_STATIC="$LIB" refrust
# 3. the vectors
python3 "$REPO/tests/golden/make_block_golden.py" "$REPO/oracle/_ref/libmmref_rust.so" "$REPO/tests/golden/block_vectors.npz"
echo "wrote tests/golden/block_vectors.npz - commit it; python -m pytest tests/test_block_oracle.py now pins the restatement"
