"""Nucleotide prefilter: exact k-mer matching over the 4-letter alphabet (Search.cpp:180-198:
--exact-kmer-matching, k = 15), one nucleotide matrix for seeding and ungapped scoring (Prefiltering.cpp:62-66), and
QueryMatcher::matchQuery's isNucleotide branch (:147-177).

CPU: the oracle against the REAL reference classes.  GPU: the device against the oracle and the recorded vectors."""
import os

import numpy as np
import pytest

from oracle import pyoracle

HERE = os.path.dirname(os.path.abspath(__file__))


def nucl_case(seed, n_targets=400, n_queries=8, tlen=(300, 3000), qlen=(150, 1200), sub=0.08, repeats=True):
    """random nucleotide targets; queries = mutated pieces of targets (some reverse strand pieces would be separate
    entries made by extractframes - here plain), a few N, and tandem repeats so that one target collects several
    saturated diagonals"""
    rng = np.random.default_rng(seed)
    tl = [rng.integers(0, 4, int(rng.integers(*tlen))).astype(np.uint8) for _ in range(n_targets)]
    qs = []
    for qi in range(n_queries):
        src = int(rng.integers(0, n_targets))
        L = int(rng.integers(*qlen))
        t = tl[src]
        a = int(rng.integers(0, max(1, len(t) - L)))
        piece = t[a:a + L].copy()
        m = rng.random(len(piece)) < sub
        piece[m] = rng.integers(0, 4, int(m.sum()))
        if qi % 3 == 0 and len(piece) > 50:
            piece[int(rng.integers(0, len(piece)))] = 4          # an N
        qs.append(piece)
        if repeats and qi % 2 == 0:
            # the same piece twice more in another target: several high-scoring diagonals for one target
            other = int(rng.integers(0, n_targets))
            tl[other] = np.concatenate([tl[other][:100], piece, rng.integers(0, 4, 37).astype(np.uint8), piece, tl[other][100:]])
    from mmseqs2_amd import workloads as wl
    tres, toff = wl.seqs_from_list(tl)
    return qs, tres, toff


def nucl_oracle(k, spaced=True):
    m = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    mat = m["nucleotide"].astype(np.int8).reshape(5, 5)
    return pyoracle.PfOracle(mat.astype(np.int16), mat, k=k, spaced=spaced), mat


@pytest.mark.skipif(not (pyoracle.ref_available() and pyoracle.ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
@pytest.mark.parametrize("k,spaced", [(11, True), (13, True), (9, False)])
def test_oracle_nucleotide_prefilter_equals_reference(k, spaced):
    ref = pyoracle.RefNuclPrefilter(k, spaced)
    o, mat = nucl_oracle(k, spaced)
    assert np.array_equal(ref.matrix(), mat)
    qs, tres, toff = nucl_case(100 + k)
    ref.build_index(tres, toff)
    o.build_index(tres, toff, 0)
    ro, ri, rp = ref.index_dump()
    assert np.array_equal(ro, o.offsets) and np.array_equal(ri, o.ids[:o.n_entries]) and np.array_equal(rp, o.pos[:o.n_entries])
    n_hits = n_sat = ties = small_ties = 0
    for mh, fb in ((300, 0), (6, 16), (3, 2)):
        bins = fb if fb else 2
        for qi, q in enumerate(qs):
            r = ref.match(q, max_hits=mh, force_bins=fb)
            x = o.match(q, None, bins, max_hits=mh, exact=True, nucleotide=True)
            assert x["stats"]["rc"] == 0 and r["db_matches"] == x["stats"]["db_matches"], (mh, qi)
            if x["stats"]["sat_tie"]:
                ties += 1
                # the reference's answer depends on the order its std::sort leaves equal ids in: up to 16 elements libstdc++ sorts
                # by insertion (stable) and the restatement's stable choice must be the reference's; beyond, not comparable
                if x["stats"]["sat_len"] > 16:
                    continue
                small_ties += 1
            assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]) and np.array_equal(r["diagonal"], x["diagonal"]), (mh, qi)
            n_hits += len(r["id"])
            n_sat += int((r["score"] > 255).sum())
    assert n_hits > 30 and n_sat > 10 and ties < 6


GOLD = os.path.join(HERE, "golden", "nucl_pf.npz")
NUCL_SETTINGS = [(300, 2), (6, 16), (3, 2)]


def load_golden():
    g = np.load(GOLD, allow_pickle=False)
    qs = [g["q_%d" % i] for i in range(int(g["n_queries"]))]
    return g, qs


def test_golden_nucleotide_prefilter_against_oracle():
    """the committed vectors (recorded from the real reference, tests/golden/make_nucl_pf_golden.py) pin the oracle where
    oracle/_ref is absent"""
    g, qs = load_golden()
    k = int(g["k"])
    o, _ = nucl_oracle(k, True)
    o.build_index(g["tres"], g["toff"], 0)
    for si, (mh, bins) in enumerate(NUCL_SETTINGS):
        for qi, q in enumerate(qs):
            x = o.match(q, None, bins, max_hits=mh, exact=True, nucleotide=True)
            if int(g["tie_%d_%d" % (si, qi)]):
                assert x["stats"]["sat_tie"]
                continue
            exp = g["hits_%d_%d" % (si, qi)]
            assert np.array_equal(x["id"], exp[0]) and np.array_equal(x["score"], exp[1]) and np.array_equal(x["diagonal"], exp[2]), (si, qi)


def _device_lists(gpu, qs, mh, bins):
    queries = [dict(q=q, comp_bias=None, identity_id=None) for q in qs]
    return gpu.pf_batch(queries, 0, max_hits=mh, ref_bins=bins, exact=True, nucleotide=True)[:3]


@pytest.mark.gpu
def test_device_nucleotide_prefilter_equals_reference_vectors(gpu):
    g, qs = load_golden()
    k = int(g["k"])
    m = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    mat = m["nucleotide"].astype(np.int8).reshape(5, 5)
    gpu.load_targets(g["tres"], g["toff"], 5)
    gpu.pf_build_index(k, 5, True, None, None, mat.astype(np.int16), 0, mat)
    # the index built on the device = the oracle's (= the reference's, CPU test above)
    o, _ = nucl_oracle(k, True)
    off, ids, pos = o.build_index(g["tres"], g["toff"], 0)
    doff, dids, dpos = gpu.pf_debug_index(k, 5)
    assert np.array_equal(off, doff) and np.array_equal(ids, dids) and np.array_equal(pos, dpos)
    n_ok = 0
    for si, (mh, bins) in enumerate(NUCL_SETTINGS):
        hits, counts, status = _device_lists(gpu, qs, mh, bins)
        for qi in range(len(qs)):
            if int(g["tie_%d_%d" % (si, qi)]):      # (the committed vectors hold none)
                assert int(status[qi]) in (0, 3)
                continue
            assert int(status[qi]) == 0, (si, qi)
            exp = g["hits_%d_%d" % (si, qi)]
            h = hits[qi][: int(counts[qi])]
            assert np.array_equal(h["id"], exp[0]) and np.array_equal(h["score"], exp[1]) and np.array_equal(h["diagonal"], exp[2]), (si, qi)
            n_ok += 1
    assert n_ok >= 3 * len(qs) - 6


@pytest.mark.gpu
def test_device_nucleotide_prefilter_k15_equals_oracle(gpu):
    """the search's own parameters: k = 15 spaced (4^15 offsets), reads against contigs split at 10 kb (Search.cpp:194-198)"""
    m = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    mat = m["nucleotide"].astype(np.int8).reshape(5, 5)
    qs, tres, toff = nucl_case(515, n_targets=300, n_queries=10, tlen=(2000, 10000), qlen=(500, 5000), sub=0.05)
    gpu.load_targets(tres, toff, 5)
    gpu.pf_build_index(15, 5, True, None, None, mat.astype(np.int16), 0, mat)
    o, _ = nucl_oracle(15, True)
    o.build_index(tres, toff, 0)
    hits, counts, status = _device_lists(gpu, qs, 300, 2)
    n_hits = 0
    for qi, q in enumerate(qs):
        x = o.match(q, None, 2, max_hits=300, exact=True, nucleotide=True)
        if x["stats"]["sat_tie"] and x["stats"]["sat_len"] > 16:
            assert int(status[qi]) in (0, 3)      # beyond std::sort's insertion range: replayed on the host (0) or handed back (3)
            continue
        assert int(status[qi]) == 0      # (ties among up to 16 saturated elements: the stable choice, on the device as in the oracle)
        h = hits[qi][: int(counts[qi])]
        assert np.array_equal(h["id"], x["id"]) and np.array_equal(h["score"], x["score"]) and np.array_equal(h["diagonal"], x["diagonal"]), qi
        n_hits += len(x["id"])
    assert n_hits >= 10


@pytest.mark.gpu
def test_device_exact_kmer_matching_amino_acids(gpu):
    """--exact-kmer-matching 1 for amino-acid searches: the same index, every window matches its own k-mer only"""
    from mmseqs2_amd import capi
    from tests import pf_common as pc
    g = pc.golden()
    o = pc.pf_oracle()
    thr = int(g["kmer_thr"])
    o.build_index(g["tres"], g["toff"], thr)
    km16 = g["vtml80_kmer16"]
    s3, i3 = capi.host_score_matrix(km16, 3)
    gpu.load_targets(g["tres"], g["toff"], 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, g["blosum62_ungapped"])
    qs = pc.golden_queries(g)
    for qd in qs:
        qd["identity_id"] = None
    hits, counts, status = gpu.pf_batch(qs, thr, max_hits=300, ref_bins=2, exact=True)[:3]
    for qi, qd in enumerate(qs):
        x = o.match(qd["q"], qd["comp_bias"], 2, max_hits=300, exact=True)
        h = hits[qi][: int(counts[qi])]
        assert int(status[qi]) == 0
        assert np.array_equal(h["id"], x["id"]) and np.array_equal(h["score"], x["score"]) and np.array_equal(h["diagonal"], x["diagonal"]), qi


def _tie_case(seed, k, spaced, n_targets=400, n_queries=60):
    """queries = a stretch S twice; S sits in m targets: every such target collects two saturated diagonals with (often) the same
    exact score, the query 2 m saturated elements"""
    from mmseqs2_amd import workloads as wl
    rng = np.random.default_rng(seed)
    tl = [rng.integers(0, 4, int(rng.integers(600, 3000))).astype(np.uint8) for _ in range(n_targets)]
    qs = []
    for _ in range(n_queries):
        m = int(rng.choice([1, 2, 3, 5, 8, 10, 14, 20]))
        S = rng.integers(0, 4, int(rng.integers(180, 400))).astype(np.uint8)
        for tid in rng.choice(n_targets, m, replace=False):
            a = int(rng.integers(0, len(tl[tid])))
            tl[tid] = np.concatenate([tl[tid][:a], S, tl[tid][a:]])
        qs.append(np.concatenate([S, S]))
    tres, toff = wl.seqs_from_list(tl)
    return qs, tres, toff


def _diag_score(q, t, diag16, mat):
    """UngappedAlignment's exact score of one diagonal (no composition correction for nucleotides): best local sum"""
    d = int(np.int16(np.uint16(diag16)))
    qs_, ts_ = (d, 0) if d >= 0 else (0, -d)
    n = min(len(q) - qs_, len(t) - ts_)
    if n <= 0:
        return 0
    s = mat[q[qs_:qs_ + n].astype(np.int64), t[ts_:ts_ + n].astype(np.int64)].astype(np.int64)
    best = run = 0
    for v in s:
        run = max(0, run + int(v))
        best = max(best, run)
    return best


@pytest.mark.skipif(not (pyoracle.ref_available() and pyoracle.ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_saturated_ties_stable_up_to_16_and_std_sort_replay_beyond(tmp_path):
    """QueryMatcher.cpp:147-177 sorts a query's saturated elements by id with std::sort.  (1) With at most 16 of them libstdc++
    sorts by insertion - stable - so the oracle's (and the device's) stable choice must be the REAL reference's.  (2) Beyond, the
    library replays the same std::sort over the elements in the reference's array order (mmseqs2_amd/csrc/sat_ties.h, compiled
    here for the host): the diagonals it keeps must be the real reference's, which the stable choice is not."""
    import ctypes
    import subprocess
    so = str(tmp_path / "libsatties.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(HERE, "..", "mmseqs2_amd", "csrc"),
                           os.path.join(HERE, "sat_ties_check.cpp"), "-o", so])
    L = ctypes.CDLL(so)
    k, spaced = 11, True
    ref = pyoracle.RefNuclPrefilter(k, spaced)
    o, mat = nucl_oracle(k, spaced)
    qs, tres, toff = _tie_case(78, k, spaced)
    ref.build_index(tres, toff)
    o.build_index(tres, toff, 0)
    small = large = large_unstable = 0
    for q in qs:
        r = ref.match(q, max_hits=300, force_bins=0)
        x = o.match(q, None, 2, max_hits=300, exact=True, nucleotide=True, dump=True)
        if not x["stats"]["sat_tie"]:
            continue
        stable_is_reference = np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]) and np.array_equal(r["diagonal"], x["diagonal"])
        if x["stats"]["sat_len"] <= 16:
            assert stable_is_reference, x["stats"]["sat_len"]
            small += 1
            continue
        large += 1
        large_unstable += not stable_is_reference
        # the saturated elements in the reference's array order (the dump is foundDiagonals after findDuplicates + scoring)
        n = int(x["stats"]["double_hits"])
        sel = np.nonzero(x["dd_count"][:n] >= 255)[0]
        assert len(sel) == x["stats"]["sat_len"]
        ids = np.ascontiguousarray(x["dd_id"][sel], np.uint32)
        dg = np.ascontiguousarray(x["dd_diag"][sel], np.uint16)
        sc = np.array([_diag_score(q, tres[int(toff[t]):int(toff[t + 1])], d, mat) for t, d in zip(ids, dg)], np.uint32)
        arr = np.arange(len(sel), dtype=np.uint32)
        out_id, out_dg = np.zeros(len(sel), np.uint32), np.zeros(len(sel), np.uint16)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        g = L.sat_ties_resolve(p(ids), p(arr), p(sc), p(dg), len(sel), 0, p(out_id), p(out_dg))
        kept = dict(zip(out_id[:g].tolist(), out_dg[:g].tolist()))
        # everything else of the hit list is the stable run's; the diagonal of a saturated target is the replay's
        assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"])
        for t, d in zip(r["id"].tolist(), r["diagonal"].tolist()):
            if t in kept:
                assert d == kept[t], (t, d, kept[t])
    assert small >= 10 and large >= 10 and large_unstable >= 5, (small, large, large_unstable)
