"""CPU tests of the nucleotide alignment oracle (oracle/nucl_oracle.c): against the
committed vectors recorded from the real reference (tests/golden/nucl_vectors.npz, made by make_golden.py) and,
where oracle/_ref exists, against the real BandedNucleotideAligner / ksw_extz2_sse directly."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import nucl_common as nc


def test_oracle_matches_golden_vectors():
    g = nc.golden()
    orc = po.NuclOracle()
    cases = nc.golden_cases(g)
    assert len(cases) > 300
    long_bt = 0
    for q, t, diag, rev, pq, pt, exp, bt in cases:
        res, s = orc.align(q, t, g["mat"].reshape(-1), g["reverse"], 5, 2, 40, diag, rev, pq, pt)
        assert res[:6] == exp[:6] and s == bt, (len(q), len(t), diag, rev, res, exp)
        long_bt += len(bt) > 1000
    assert long_bt >= 10          # the vectors hold real long alignments, not only seeds


def test_golden_vectors_are_consistent():
    """size-independent properties of the recorded alignments: the backtrace spans exactly the reported rectangle and
    the identity count is what the string says"""
    g = nc.golden()
    rl = g["reverse"]
    for q, t, diag, rev, pq, pt, exp, bt in nc.golden_cases(g):
        score, qs, qe, ts, te, ident = exp[:6]
        if not bt:
            continue
        assert bt.count("M") + bt.count("I") == qe - qs + 1
        assert bt.count("M") + bt.count("D") == te - ts + 1
        qa = np.array([rl[x] for x in q[::-1]], np.uint8) if rev else q
        qp, tp, ids = qs, ts, 0
        for c in bt:
            if c == "M":
                ids += int(qa[qp] == t[tp])
                qp += 1
                tp += 1
            elif c == "I":
                qp += 1
            else:
                tp += 1
        assert ids == ident


@pytest.mark.skipif(not (po.ref_available() and po.ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_oracle_matches_reference_fuzz():
    rng = np.random.default_rng(5)
    ref, orc = po.RefNucl(), po.NuclOracle()
    mat, rl = ref.matrix(), ref.reverse_lookup()
    letters = lambda a: "".join(po.NUCL_LETTERS[int(x)] for x in a)
    n_ksw = n_al = 0
    for it in range(40):
        L = int(rng.choice([3, 16, 31, 64, 65, 130, 500, 2000]))
        base = rng.integers(0, 4, size=L).astype(np.uint8)
        q = nc.mutate(rng, base, rng.choice([0.0, 0.05, 0.2]), rng.choice([0.0, 0.02, 0.06]))
        t = nc.mutate(rng, base, rng.choice([0.0, 0.05]), rng.choice([0.0, 0.02]))
        for flag in (po.KSW_SCORE_ONLY | po.KSW_EXTZ_ONLY, po.KSW_EXTZ_ONLY):
            a = ref.ksw_extz2(q, t, mat.reshape(-1), 5, 2, 64, 40, flag)
            b = orc.ksw_extz2(q, t, mat.reshape(-1), 5, 2, 64, 40, flag)
            assert a[0] == b[0] and np.array_equal(a[1], b[1])
            n_ksw += 1
        pq, pt = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        ref.set_query(letters(q), pq)
        for rev in (0, 1):
            tt = np.array([rl[x] for x in t[::-1]], np.uint8) if rev else t
            for diag in (0, int(rng.integers(0, 65536))):
                assert ref.align(letters(tt), diag, rev, pt) == orc.align(q, tt, mat.reshape(-1), rl, 5, 2, 40, diag, rev, pq, pt)
                n_al += 1
    ref.close()
    assert n_ksw == 80 and n_al == 160
