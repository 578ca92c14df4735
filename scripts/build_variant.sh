#!/bin/bash
# A/B builds of libmmgpu.so for GPU-box experiments: scripts/build_variant.sh NAME FILE.hip "-DFLAG=.. ..." [FILE2.hip "flags" ...]
# recompiles the named sources with the extra flags and links them with the stock objects of mmseqs2_amd/lib/obj into
# variants/NAME/libmmgpu.so (variants/ is git-ignored and travels to the GPU box); use with MMGPU_LIB=variants/NAME/libmmgpu.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/variants/$NAME
mkdir -p "$OUT/obj"
make -s -C "$ROOT/mmseqs2_amd/csrc" >/dev/null
cp "$ROOT"/mmseqs2_amd/lib/obj/*.o "$OUT/obj/"
while [ $# -gt 0 ]; do
  SRC=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c "$ROOT/mmseqs2_amd/csrc/$SRC" -o "$OUT/obj/${SRC%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT/libmmgpu.so" "$OUT"/obj/*.o
rm -rf "$OUT/obj"
echo "$OUT/libmmgpu.so"
