"""CPU tests: the plain-C oracle against (a) the committed golden vectors generated from the real reference
and (b) the real reference itself when it is available in this container (oracle/_ref)."""
import numpy as np
import pytest

from mmseqs2_amd import workloads as wl


def _pairs(v):
    qoff, toff = v["qoff"].astype(np.int64), v["toff"].astype(np.int64)
    for i in range(len(qoff) - 1):
        yield (i, v["qres"][qoff[i]:qoff[i + 1]], v["cb"][qoff[i]:qoff[i + 1]], v["tres"][toff[i]:toff[i + 1]])


def test_oracle_matches_golden_sw_vectors(oracle, matrices, sw_vectors):
    mat = matrices["blosum62_sw"]
    go, ge = int(sw_vectors["gap_open"]), int(sw_vectors["gap_extend"])
    n_word = 0
    for i, q, cb, t in _pairs(sw_vectors):
        exp = sw_vectors["expect"][i]
        r = oracle.sw_align(q, cb, t, mat, go, ge, need_start=True, need_bt=True)
        got = [r["score"], r["q_end"], r["t_end"], r["q_start"], r["t_start"], r["word"], r["ident"]]
        assert got == list(exp), "pair %d" % i
        assert r["bt"] == sw_vectors["bt"][i], "backtrace of pair %d" % i
        n_word += r["word"]
    assert n_word > 20   # the fixture exercises the int16 branch


def test_oracle_identity_score(oracle, matrices, sw_vectors):
    mat = matrices["blosum62_sw"]
    for i, q, cb, t in _pairs(sw_vectors):
        if len(q) == len(t) and np.array_equal(q, t):
            s = oracle.sw_score_identical(q, cb, t, mat)
            full = sum(int(mat[a, a]) + int(b) for a, b in zip(q, cb))
            assert s == np.int16(full)


def test_comp_bias_oracle_vs_reference(oracle, reflib):
    import ctypes
    sub = np.zeros((21, 21), np.int16)
    pb = np.zeros(21, np.float64)
    reflib.L.mmref_get_matrix16.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    reflib.L.mmref_get_pback.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    reflib.L.mmref_get_matrix16(reflib.c, sub.ctypes.data)
    reflib.L.mmref_get_pback(reflib.c, pb.ctypes.data)
    rng = np.random.default_rng(5)
    for L in [1, 2, 19, 20, 21, 40, 41, 77, 350, 2000]:
        s = rng.integers(0, 21, L).astype(np.uint8)
        a = reflib.comp_bias(s, 1.0)
        b = oracle.comp_bias(sub, pb, s, 1.0)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), L   # bit-identical floats


def test_oracle_fuzz_vs_reference(oracle, reflib):
    """Scores, ends, starts, backtraces and the uint8/int16 decision, 300 seeded pairs x 3 modes."""
    mat = reflib.matrix()
    rng = np.random.default_rng(11)
    n_word = 0
    for it in range(300):
        Lq, Lt = int(rng.integers(1, 900)), int(rng.integers(1, 900))
        q = rng.integers(0, 21, Lq).astype(np.uint8)
        t = rng.integers(0, 21, Lt).astype(np.uint8)
        if it % 3 == 0 and Lq > 12:
            t = wl.mutate(rng, q, float(rng.uniform(0.3, 1.0)))
        reflib.sw_set_query(q)
        cb = oracle.round_comp_bias(reflib.comp_bias(q))
        for mode in (0, 1, 2):
            a = reflib.sw_align(t, mode)
            b = oracle.sw_align(q, cb, t, mat, 11, 1, need_start=mode >= 1, need_bt=mode >= 2)
            if a["score"] == 0:
                assert b["score"] == 0 and b["t_end"] == -1
                continue
            keys = ["score", "q_end", "t_end", "word"] + (["q_start", "t_start"] if mode >= 1 else []) + \
                   (["bt", "ident"] if mode >= 2 else [])
            for k in keys:
                assert a[k] == b[k], (it, mode, k)
        n_word += a["word"]
    assert n_word > 30


def test_golden_matrices_match_reference(reflib, matrices):
    assert np.array_equal(reflib.matrix(), matrices["blosum62_sw"])
    assert reflib.num2aa() == bytes(matrices["num2aa"]).decode() == wl.NUM2AA


def test_word_hits_backtrace_rescored_equals_score(sw_vectors, matrices):
    """for hits whose score left the uint8 range (word == 1) the stock reference takes start and CIGAR
    from the Rust block-aligner, which cannot be built here.  The contract that can be pinned: re-scoring the backtrace the
    fallback path produces (what the device returns too) reproduces score1 and ends at (qEnd, dbEnd) - for EVERY recorded
    pair, word == 1 or not."""
    from tests.rescore import rescore
    v = sw_vectors
    mat = matrices["blosum62_sw"]
    qoff, toff = v["qoff"].astype(np.int64), v["toff"].astype(np.int64)
    n_word = 0
    for i in range(len(qoff) - 1):
        score, q_end, t_end, q_start, t_start, word, ident = [int(x) for x in v["expect"][i]]
        if score <= 0 or not v["bt"][i]:
            continue
        q, cb, t = v["qres"][qoff[i]:qoff[i + 1]], v["cb"][qoff[i]:qoff[i + 1]], v["tres"][toff[i]:toff[i + 1]]
        s, qe, te = rescore(q, cb, t, mat, int(v["gap_open"]), int(v["gap_extend"]), q_start, t_start, v["bt"][i])
        assert (s, qe, te) == (score, q_end, t_end), (i, word, s, score)
        n_word += word
    assert n_word >= 20
