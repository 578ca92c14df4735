"""Records tests/golden/kmer_score_pf.npz from the REAL reference (oracle/_ref/libmmref.so): the queries and targets of
prefilter_vectors.npz through QueryMatcher::matchQuery with diagonalScoring == false (--diag-score 0) for the settings of
tests/test_kmer_score.py (max_hits, forced bin count, --min-ungapped-score).  Run in the build container."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle                                        # noqa: E402
from tests import pf_common as pc                                  # noqa: E402
from tests.test_kmer_score import KS_SETTINGS                      # noqa: E402

g = pc.golden()
ref = pyoracle.RefPrefilter(int(g["k"]))
ref.build_index(g["tres"], g["toff"], int(g["kmer_thr"]))
qs = pc.golden_queries(g)
out = {}
n = 0
for si, (mh, bins, mds) in enumerate(KS_SETTINGS):
    got = ref.make_matcher(max_hits=mh, force_bins=bins, min_diag_score=mds, diag_score=False)
    assert got == bins
    for qi, qd in enumerate(qs):
        r = ref.match(qd["q"], qd["identity_id"])
        assert r["db_matches"] == int(g["db_matches"][qi])
        out["hits_%d_%d" % (si, qi)] = np.stack([r["id"].astype(np.int64), r["score"].astype(np.int64), r["diagonal"].astype(np.int64)])
        n += len(r["id"])
np.savez_compressed(os.path.join(HERE, "kmer_score_pf.npz"), **out)
print("wrote kmer_score_pf.npz:", n, "hits over", len(KS_SETTINGS), "settings x", len(qs), "queries")
