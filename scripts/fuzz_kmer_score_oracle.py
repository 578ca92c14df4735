"""Fuzz of the --diag-score 0 restatement (oracle/prefilter_oracle.c, kmer_score) against the real reference classes
(oracle/_ref/libmmref.so): random databases, --max-seqs, forced bin counts, --min-ungapped-score 0..40, composition bias on / off.
Runs in the build container only (needs /root/reference/data)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle
from tests import pf_common as pc
from mmseqs2_amd import workloads as wl
ref = pyoracle.RefPrefilter(6)
km8, um8, km16, pback = ref.matrices()
o = pc.pf_oracle()
swo = pyoracle.Oracle()
rng = np.random.default_rng(123)
thr = pyoracle.kmer_threshold(5.7, 6)
bad = tot = 0
for rep in range(4):
    (qres, qoff), (tres, toff) = pc.synthetic_case(8, int(rng.integers(300, 2500)), seed=200 + rep, planted=float(rng.uniform(0.2, 0.9)))
    ref.build_index(tres, toff, thr)
    o.build_index(tres, toff, thr)
    for trial in range(8):
        mh = int(rng.choice([1, 3, 10, 50, 300]))
        fb = int(rng.choice([0, 2, 4, 16, 64, 256, 2048]))
        mds = int(rng.choice([0, 1, 2, 5, 15, 40]))
        cb = bool(rng.integers(0, 2))
        bins = ref.make_matcher(max_hits=mh, force_bins=fb, min_diag_score=mds, diag_score=False, comp_bias=cb)
        for qi, q in enumerate(wl.split(qres, qoff)):
            bias = swo.comp_bias(km16, pback, q) if cb else None
            ident = None if qi % 3 else qi
            r = ref.match(q, ident)
            x = o.match(q, bias, bins, max_hits=mh, min_diag_score=mds, identity_id=ident, kmer_score=True)
            ok = np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]) and np.array_equal(r["diagonal"], x["diagonal"])
            tot += 1
            if not ok:
                bad += 1
                print("DIFF", rep, mh, fb, mds, cb, qi, len(r["id"]), len(x["id"]))
print("compared", tot, "differing", bad)
