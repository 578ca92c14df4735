"""--diag-score 0 (Prefiltering's diagonalScoring == false): the prefilter score of a target is the number of its double
k-mer matches (CacheFriendlyOperations::findDuplicates with computeTotalScore, CacheFriendlyOperations.cpp:218-239;
QueryMatcher.cpp:215-232, getResult<KMER_SCORE>).  Golden lists recorded from the real reference
(tests/golden/make_kmer_score_golden.py) against the restatement, the reference itself where it is available, and the
device path."""
import os

import numpy as np
import pytest

from tests import pf_common as pc

KS_SETTINGS = [(300, 2, 15), (300, 16, 1), (10, 4, 1), (40, 128, 3), (25, 2, 2)]     # (max_hits, reference bins, --min-ungapped-score)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kmer_score_pf.npz")


def _gold():
    return dict(np.load(GOLD))


def _expected(G, si, qi):
    h = G["hits_%d_%d" % (si, qi)]
    return h[0], h[1], h[2]


def test_oracle_matches_golden_kmer_score_lists():
    g = pc.golden()
    G = _gold()
    o = pc.pf_oracle()
    o.build_index(g["tres"], g["toff"], int(g["kmer_thr"]))
    qs = pc.golden_queries(g)
    n_hits = n_cut = 0
    for si, (mh, bins, mds) in enumerate(KS_SETTINGS):
        for qi, qd in enumerate(qs):
            r = o.match(qd["q"], qd["comp_bias"], bins, max_hits=mh, min_diag_score=mds, identity_id=qd["identity_id"], kmer_score=True)
            ids, sc, dg = _expected(G, si, qi)
            assert r["stats"]["rc"] == 0 and r["stats"]["overflow"] == 0 and r["stats"]["big_list"] == 0
            assert np.array_equal(r["id"], ids) and np.array_equal(r["score"], sc) and np.array_equal(r["diagonal"], dg), (si, qi)
            n_hits += len(ids)
            n_cut += len(ids) == mh
    assert n_hits > 500 and n_cut > 5       # lists exist and some are cut at max_hits (tie order at the cut)


def test_oracle_kmer_score_vs_reference():
    from oracle import pyoracle
    if not (pyoracle.ref_available() and pyoracle.ref_matrix_available()):
        pytest.skip("needs oracle/_ref/libmmref.so and /root/reference/data")
    from mmseqs2_amd import workloads as wl
    ref = pyoracle.RefPrefilter(6)
    km8, um8, km16, pback = ref.matrices()
    o = pc.pf_oracle()
    (qres, qoff), (tres, toff) = pc.synthetic_case(12, 1500, seed=91, planted=0.6)
    thr = pyoracle.kmer_threshold(5.7, 6)
    ref.build_index(tres, toff, thr)
    o.build_index(tres, toff, thr)
    swo = pyoracle.Oracle()
    total = 0
    for mh, fb, mds in ((300, 0, 15), (20, 32, 1), (7, 2, 2), (300, 512, 4)):
        bins = ref.make_matcher(max_hits=mh, force_bins=fb, min_diag_score=mds, diag_score=False)
        for qi, q in enumerate(wl.split(qres, qoff)):
            cb = swo.comp_bias(km16, pback, q)
            ident = None if qi % 2 else qi
            r = ref.match(q, ident)
            x = o.match(q, cb, bins, max_hits=mh, min_diag_score=mds, identity_id=ident, kmer_score=True)
            assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]), (mh, fb, qi)
            assert np.array_equal(r["diagonal"], x["diagonal"]) and r["db_matches"] == x["stats"]["db_matches"]
            total += len(r["id"])
    assert total > 300
    g = pc.golden()
    o.build_index(g["tres"], g["toff"], int(g["kmer_thr"]))


@pytest.mark.gpu
def test_device_kmer_score_lists_match_golden(gpu):
    from tests import pf_gpu_check as chk
    g = pc.golden()
    G = _gold()
    thr = int(g["kmer_thr"])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)
    qs = pc.golden_queries(g)
    for si, (mh, bins, mds) in enumerate(KS_SETTINGS):
        hits, counts, status, stats = gpu.pf_batch(qs, thr, max_hits=mh, min_diag_score=mds, ref_bins=bins, kmer_score=True)
        for qi in range(len(qs)):
            ids, sc, dg = _expected(G, si, qi)
            assert int(status[qi]) == 0
            n = int(counts[qi])
            assert n == len(ids), (si, qi, n, len(ids))
            h = hits[qi][:n]
            assert np.array_equal(h["id"], ids) and np.array_equal(h["score"], sc) and np.array_equal(h["diagonal"], dg), (si, qi)
            assert int(stats[qi]["db_matches"]) == int(g["db_matches"][qi])
