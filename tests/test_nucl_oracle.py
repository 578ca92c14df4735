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


@pytest.mark.skipif(not (po.ref_available() and po.ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_oracle_matches_reference_with_wrapped_scoring():
    """--wrapped-scoring (circular sequences): the query is handed over written twice (Alignment.cpp:332-337); the ungapped seed
    wraps around (computeUngappedWrappedAlignment, when the doubled query is at least twice the target) or is taken over the first
    half, the extensions are capped at the original length (BandedNucleotideAligner.cpp:98-113,171-174,189-191)."""
    rng = np.random.default_rng(17)
    ref, orc = po.RefNucl(), po.NuclOracle()
    mat, rl = ref.matrix(), ref.reverse_lookup()
    letters = lambda a: "".join(po.NUCL_LETTERS[int(x)] for x in a)
    n = wrapped_seed = 0
    for it in range(60):
        L = int(rng.choice([40, 64, 130, 500, 1500]))
        circle = rng.integers(0, 4, size=L).astype(np.uint8)
        rot = int(rng.integers(0, L))
        q1 = nc.mutate(rng, np.roll(circle, -rot), rng.choice([0.0, 0.04]), rng.choice([0.0, 0.02]))      # the circle read from another origin
        q = np.concatenate([q1, q1])
        # targets: the circle itself (longer than half the doubled query or not), a piece of it, and the circle with an insertion
        for t in (nc.mutate(rng, circle, 0.03, 0.01), circle[: max(8, L // 3)].copy(), np.concatenate([circle, circle[: L // 2]])):
            pq, pt = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            ref.set_query(letters(q), pq)
            for rev in (0, 1):
                tt = np.array([rl[x] for x in t[::-1]], np.uint8) if rev else t
                for diag in (0, rot & 0xFFFF, (-rot) & 0xFFFF, int(rng.integers(0, 65536))):
                    e = ref.align(letters(tt), diag, rev, pt, wrapped=True)
                    o = orc.align(q, tt, mat.reshape(-1), rl, 5, 2, 40, diag, rev, pq, pt, wrapped=True)
                    assert e == o, (L, len(tt), diag, rev, e[0], o[0])
                    n += 1
                    wrapped_seed += len(q) >= 2 * len(tt)
    ref.close()
    assert n == 60 * 3 * 2 * 4 and wrapped_seed > 300


@pytest.mark.skipif(not (po.ref_available() and po.ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_long_targets_try_every_shift_of_the_16_bit_diagonal():
    """targets of 32768 residues or more: the prefilter diagonal is a 16-bit value and computeUngappedAlignment tries every
    65536-shift that fits (DistanceCalculator.h:93-112); oracle and the kernel source (emulated lanes) against the reference"""
    from tests import test_nucl_emu as te
    rng = np.random.default_rng(9)
    ref, orc = po.RefNucl(), po.NuclOracle()
    mat, rl = ref.matrix(), ref.reverse_lookup()
    letters = lambda a: "".join(po.NUCL_LETTERS[int(x)] for x in a)
    L = te._lib()
    queries, targets, pairs, expected = [], [], [], []
    for tlen, start, qlen in ((40000, 35000, 700), (65000, 60111, 400), (33000, 100, 900)):
        contig = rng.integers(0, 4, size=tlen).astype(np.uint8)
        read = nc.mutate(rng, contig[start:start + qlen], 0.05, 0.01)
        queries.append(read)
        targets.append(contig)
        ref.set_query(letters(read), 4)
        for diag in ((-start) & 0xFFFF, (-start + 3) & 0xFFFF, 12345):
            e = ref.align(letters(contig), diag, 0, 4)
            o = orc.align(read, contig, mat.reshape(-1), rl, 5, 2, 40, diag, 0, 4, 4)
            assert e == o, (tlen, start, diag, e[0], o[0])
            pairs.append((len(queries) - 1, len(targets) - 1, diag, 0))
            expected.append(e)
    assert max(len(e[1]) for e in expected) > 600          # the true diagonal was found beyond 16 bits
    hits, strs = te.emu_align(L, mat, rl, queries, targets, pairs, 4, 4)
    for e, h, s in zip(expected, hits, strs):
        got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
        assert got == e[0][:6] and s == e[1]
    ref.close()
