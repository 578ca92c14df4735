#!/usr/bin/env python3
"""Digest of the device sources a profile was taken with: sha256 over the names and contents of mmseqs2_amd/csrc/*.hip, *.h and
the Makefile, in name order.  rocprof_summary.py writes it into the header of every summary under profiles/; bench.py reads
`roofline.traffic` from a PMC summary only when the digest there equals the digest of the sources it runs with, and
tests/test_bench_contract.py fails when the committed round's PMC summaries were taken with other kernels.  (The GPU box gets
a snapshot without .git, so the digest is computed from the files; for a clean tree it changes exactly when
`git rev-parse HEAD:mmseqs2_amd/csrc` does.)"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_digest(root=ROOT):
    d = os.path.join(root, "mmseqs2_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(d, name), "rb").read())
            h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_digest())
