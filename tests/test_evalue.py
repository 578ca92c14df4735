"""mmseqs2_amd/evalue.py (the E-value / start-score threshold a caller without the reference hands to the device) against
the real reference's EvalueComputation (oracle/_ref, ALP): E-values to 1e-6 relative, thresholds exactly."""
import numpy as np
import pytest

from mmseqs2_amd import evalue as ev
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not (po.ref_available() and po.ref_matrix_available()),
                                reason="needs oracle/_ref/libmmref.so and /root/reference/data")


@pytest.mark.parametrize("db_res", [150000, 3.0e7, 2.8e8, 2.3e9])
def test_evalue_and_thresholds_match_reference(db_res):
    ref = po.RefLib(db_residues=int(db_res))
    for qlen in [30, 59, 120, 233, 350, 777, 2000, 5000, 32000]:
        for s in [20, 35, 50, 80, 120, 254, 400, 2000]:
            a, b = ref.evalue(s, qlen), ev.evalue(s, qlen, db_res)
            assert abs(a - b) <= 2e-5 * max(abs(a), 1e-300), (qlen, s, a, b)
        for thr in [1e-3, 1e-5, 10.0, 1e-30]:
            lo, hi = 1, 32767      # the reference-side threshold by the same bisection on the reference's own function
            while lo < hi:
                mid = (lo + hi) // 2
                if ref.evalue(mid, qlen) > thr:
                    lo = mid + 1
                else:
                    hi = mid
            assert ev.min_score_for_evalue(thr, qlen, db_res) == lo, (qlen, thr)
    assert abs(ref.bitscore(100) - ev.bit_score(100)) < 1e-9


def test_vectorised_thresholds_equal_the_scalar_ones():
    L = np.array([30, 31, 59, 120, 233, 233, 350, 777, 2000, 5000, 32000])
    for db_res in (150000, 2.8e8):
        for thr in (1e-3, 10.0, 1e-30, 1e-300):
            a = ev.min_scores_for_evalue(thr, L, db_res)
            b = np.array([ev.min_score_for_evalue(thr, int(x), db_res) for x in L])
            assert np.array_equal(a, b), (db_res, thr)
