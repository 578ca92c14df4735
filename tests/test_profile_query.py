"""Profile queries (alignment half): Matcher::initQuery hands Sequence::getAlignmentProfile() to
ssw_init (src/alignment/Matcher.cpp:49-60, StripedSmithWaterman.cpp:1364-1448); the scans then read the query's own
score rows instead of matrix + composition bias.

CPU: the oracle's profile form against the REAL reference driven with profile-database entries (oracle/_ref).
GPU: the device (C-ABI `mmgpu_sw_query.profile`) against the oracle and against the golden vectors recorded from the reference."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Oracle, RefLib, ref_available, ref_matrix_available

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "profile_sw.npz")


def make_entry(rng, mat, L, sharp=1.0):
    """One profile-database entry: [L][25] int8 - 20 scores (x4 scale), query letter, consensus letter, Neff, 2 reserved
    (Sequence.cpp:301-325).  Scores: the substitution row of a random consensus letter, scaled and jittered."""
    cons = rng.integers(0, 20, L).astype(np.uint8)
    e = np.zeros((L, 25), np.int8)
    rows = mat[cons][:, :20].astype(np.int32)                      # [L][20]
    sc = np.clip(np.rint(rows * 4 * sharp + rng.integers(-6, 7, rows.shape)), -120, 120)
    e[:, :20] = sc.astype(np.int8)
    e[:, 20] = cons
    e[:, 21] = cons
    e[:, 22] = 40
    return e, cons


def mutate(rng, seq, ident):
    out = []
    for c in seq:
        r = rng.random()
        if r < 0.03:
            continue
        out.append(c if rng.random() < ident else rng.integers(0, 20))
        if rng.random() < 0.03:
            out.append(rng.integers(0, 20))
    return np.array(out if out else [0], np.uint8)


def cases(rng, mat, n_queries=6):
    out = []
    for qi in range(n_queries):
        L = int(rng.integers(20, 700)) if qi else 1200          # one multi-tile query
        e, cons = make_entry(rng, mat, L, sharp=1.0 if qi % 2 else 0.6)
        ts = [mutate(rng, cons, 0.8), mutate(rng, cons, 0.45), mutate(rng, cons[L // 3:], 0.7),
              rng.integers(0, 20, int(rng.integers(15, 500))).astype(np.uint8),
              np.concatenate([rng.integers(0, 20, 40).astype(np.uint8), mutate(rng, cons[: max(10, L // 2)], 0.6)])]
        out.append((e, ts))
    return out


@pytest.mark.skipif(not (ref_available() and ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_oracle_profile_alignment_equals_reference():
    ref = RefLib(comp_bias=False)
    orc = Oracle()
    mat = np.load(os.path.join(HERE, "golden", "matrices.npz"))["blosum62_sw"]
    rng = np.random.default_rng(5)
    n = 0
    words = 0
    for e, ts in cases(rng, mat):
        prof, cons = ref.sw_set_profile_query(e)
        assert (prof == (e[:, :20].astype(np.int32) / 4).astype(np.int8).T).all()    # mapProfile: score / 4, truncating (:334)
        for t in ts:
            for mode in (0, 1, 2):
                r = ref.sw_align(t, mode=mode)
                o = orc.sw_align_profile(prof, cons, t, 21, 11, 1, need_start=mode >= 1 and r["t_end"] != -1, need_bt=mode == 2)
                assert (o["score"], o["q_end"], o["t_end"], o["word"]) == (r["score"], r["q_end"], r["t_end"], r["word"])
                if mode >= 1 and r["t_end"] != -1:
                    assert (o["q_start"], o["t_start"]) == (r["q_start"], r["t_start"])
                if mode == 2 and r["t_end"] != -1:
                    assert o["bt"] == r["bt"] and o["ident"] == r["ident"]
                n += 1
                words += r["word"]
    assert n >= 90 and words > 0


def load_golden():
    g = np.load(GOLD, allow_pickle=False)
    out = []
    for qi in range(int(g["n_queries"])):
        e = g["entry_%d" % qi]
        ts = [g["t_%d_%d" % (qi, k)] for k in range(int(g["n_targets"][qi]))]
        exp = g["exp_%d" % qi]       # [targets][score, q_end, t_end, word, q_start, t_start, ident]
        bts = [str(x) for x in g["bt_%d" % qi]]
        out.append((e, ts, exp, bts))
    return out


def test_golden_profile_vectors_against_oracle():
    """The committed vectors (recorded from the real reference by tests/golden/make_profile_golden.py) pin the oracle on
    machines without oracle/_ref."""
    orc = Oracle()
    for e, ts, exp, bts in load_golden():
        prof = (e[:, :20].astype(np.int32) / 4).astype(np.int8).T.copy()
        cons = e[:, 20].astype(np.uint8)
        for k, t in enumerate(ts):
            o = orc.sw_align_profile(prof, cons, t, 21, 11, 1, need_start=exp[k][2] != -1, need_bt=exp[k][2] != -1)
            got = (o["score"], o["q_end"], o["t_end"], o["word"])
            assert got == tuple(int(x) for x in exp[k][:4])
            if exp[k][2] != -1:
                assert (o["q_start"], o["t_start"], o["ident"]) == tuple(int(x) for x in exp[k][4:7])
                assert o["bt"] == bts[k]


@pytest.mark.gpu
def test_device_profile_queries_equal_reference_vectors():
    import mmseqs2_amd
    from mmseqs2_amd import workloads as wl
    mat = np.load(os.path.join(HERE, "golden", "matrices.npz"))["blosum62_sw"]
    gold = load_golden()
    targets = []
    for _, ts, _, _ in gold:
        targets += ts
    tres, toff = wl.seqs_from_list(targets)
    gpu = mmseqs2_amd.MMGpu(0)
    gpu.load_targets(tres, toff, 21)
    queries = []
    base = 0
    for e, ts, exp, bts in gold:
        prof = (e[:, :20].astype(np.int32) / 4).astype(np.int8).T.copy()
        queries.append(dict(q=e[:, 20].astype(np.uint8), comp_bias=None, profile=prof,
                            targets=np.arange(base, base + len(ts), dtype=np.uint32), min_start_score=0))
        base += len(ts)
    # score + end, then + start, then CIGARs
    out0 = gpu.sw_batch(mat, 11, 1, queries, mode=0)
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    out1 = b.fetch()
    info, strs = b.traceback(np.arange(len(out1), dtype=np.uint32))
    k = 0
    for e, ts, exp, bts in gold:
        for j in range(len(ts)):
            x = exp[j]
            assert (int(out0[k]["score"]), int(out0[k]["q_end"]), int(out0[k]["t_end"]), int(out0[k]["word"])) == tuple(int(v) for v in x[:4])
            assert (int(out1[k]["score"]), int(out1[k]["q_end"]), int(out1[k]["t_end"])) == tuple(int(v) for v in x[:3])
            if x[2] != -1:
                assert (int(out1[k]["q_start"]), int(out1[k]["t_start"])) == (int(x[4]), int(x[5]))
                assert int(info[k]["status"]) == 0 and strs[k] == bts[j] and int(info[k]["ident"]) == int(x[6])
            k += 1
    b.free()


@pytest.mark.gpu
def test_device_profile_queries_equal_oracle_random():
    """Mixed batch: profile and sequence queries side by side, single- and multi-tile, against the oracle."""
    import mmseqs2_amd
    from mmseqs2_amd import workloads as wl
    from mmseqs2_amd.capi import host_comp_bias
    mats = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    mat = mats["blosum62_sw"]
    rng = np.random.default_rng(77)
    orc = Oracle()
    cs = cases(rng, mat, n_queries=8)
    targets = []
    for _, ts in cs:
        targets += ts
    tres, toff = wl.seqs_from_list(targets)
    gpu = mmseqs2_amd.MMGpu(0)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(len(targets), dtype=np.uint32)
    queries = []
    for qi, (e, ts) in enumerate(cs):
        cons = e[:, 20].astype(np.uint8)
        if qi % 3 == 2:     # a sequence query between the profile queries
            _, cb = host_comp_bias(mat.astype(np.int16), mats["blosum62_pback"], cons)
            queries.append(dict(q=cons, comp_bias=cb, targets=ids, min_start_score=0))
        else:
            prof = (e[:, :20].astype(np.int32) / 4).astype(np.int8).T.copy()
            queries.append(dict(q=cons, comp_bias=None, profile=prof, targets=ids, min_start_score=0))
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    out = b.fetch().reshape(len(queries), len(targets))
    info, strs = b.traceback(np.arange(out.size, dtype=np.uint32))
    for qi, qd in enumerate(queries):
        for k, t in enumerate(targets):
            if "profile" in qd:
                o = orc.sw_align_profile(qd["profile"], qd["q"], t, 21, 11, 1, need_start=True, need_bt=True)
            else:
                o = orc.sw_align(qd["q"], qd["comp_bias"], t, mat, 11, 1, need_start=True, need_bt=True)
            h = out[qi, k]
            assert (int(h["score"]), int(h["q_end"]), int(h["t_end"]), int(h["word"])) == (o["score"], o["q_end"], o["t_end"], o["word"])
            if o["t_end"] != -1:
                p = qi * len(targets) + k
                assert (int(h["q_start"]), int(h["t_start"])) == (o["q_start"], o["t_start"])
                assert int(info[p]["status"]) == 0 and strs[p] == o["bt"] and int(info[p]["ident"]) == o["ident"]
    b.free()
