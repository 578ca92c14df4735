#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/_ref build helper) - not part of the product path.

The reference links a Rust crate (lib/block-aligner 0.4.0, lib/block-aligner/Cargo.toml:3)
for the traceback of int16-range Smith-Waterman hits
(src/alignment/StripedSmithWaterman.cpp:943-1127).  This image has no rustc/cargo, so
the crate cannot be built.  This script reads the crate's *C header* where it lies in
the reference tree and emits do-nothing definitions for every `block_*` prototype into
the build directory (never into git).  `block_res_aa_trace_xdrop` reports score -1e9, so
the reference's own code takes its documented fallback
(StripedSmithWaterman.cpp:873-882: "Block alignment failed, falling back to
Smith-Waterman") and produces start positions / CIGARs with its own
alignStartPosBacktrace + banded_sw.  Scores, end positions, E-values are unaffected by
the stub.

usage: gen_block_stub.py <reference_root> <out.c>
"""
import re
import sys


def main():
    ref, out = sys.argv[1], sys.argv[2]
    # --skip a,b,c : functions defined elsewhere (oracle/ref_block_capi.cpp over the restated block aligner)
    skip = set()
    if "--skip" in sys.argv:
        skip = set(sys.argv[sys.argv.index("--skip") + 1].split(","))
    hdr = ref + "/lib/block-aligner/c/block_aligner.h"
    h = open(hdr).read()
    h2 = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h2 = re.sub(r"//[^\n]*", "", h2)
    protos = re.findall(r"\n\s*([A-Za-z_][A-Za-z0-9_ \*]*?\b(\w+)\s*\(([^;{}]*?)\))\s*;", h2)
    lines = ['#include "%s"\n#include <string.h>\nstatic char mm_stub_scratch[1 << 20];\n' % hdr]
    n = 0
    for full, name, _args in protos:
        if not (name.startswith("block_") or name.startswith("_block") or name.startswith("aaprofile")):
            continue
        if name in skip:
            continue
        ret = full[: full.index(name)].strip()
        if ret == "void":
            body = "{}"
        elif "AlignResult" in ret:
            body = "{ struct AlignResult r; r.score = -1000000000; r.query_idx = 0; r.reference_idx = 0; return r; }"
        elif "OpLen" in ret:
            body = "{ struct OpLen o; o.op = Sentinel; o.len = 0; return o; }"
        elif "*" in ret or "BlockHandle" in ret:
            body = "{ return (%s)mm_stub_scratch; }" % ret
        else:
            body = "{ return 0; }"
        lines.append(" ".join(full.split()) + " " + body + "\n")
        n += 1
    open(out, "w").write("".join(lines))
    print("gen_block_stub: %d stubs -> %s" % (n, out))


if __name__ == "__main__":
    main()
