"""CPU tests of the prefilter: the plain-C restatement (oracle/prefilter_oracle.c) against the golden hit lists
recorded from the real reference classes, the product's host-side table builders against the restatement, and -
where /root/reference and oracle/_ref are present - the restatement against the reference itself."""
import numpy as np
import pytest

from mmseqs2_amd import capi
from tests import pf_common as pc


def test_oracle_matches_golden_prefilter_hits():
    g = pc.golden()
    o = pc.pf_oracle()
    o.build_index(g["tres"], g["toff"], int(g["kmer_thr"]))
    qs = pc.golden_queries(g)
    n_sat = n_cut = 0
    for si, (mh, bins) in enumerate(g["settings"].tolist()):
        exp = pc.expected_hits(g, si)
        for qi, qd in enumerate(qs):
            r = o.match(qd["q"], qd["comp_bias"], bins, max_hits=mh, min_diag_score=int(g["min_diag_score"]),
                        identity_id=qd["identity_id"])
            assert r["stats"]["rc"] == 0
            assert r["stats"]["db_matches"] == int(g["db_matches"][qi])
            assert np.array_equal(r["id"], exp[qi][0]), (si, qi)
            assert np.array_equal(r["score"], exp[qi][1]), (si, qi)
            assert np.array_equal(r["diagonal"], exp[qi][2]), (si, qi)
            n_sat += int((r["score"] > 255).sum())
            n_cut += r["stats"]["truncated"]
    assert n_sat > 500 and n_cut > 20   # the fixture exercises exact rescoring and the truncated-threshold path


def test_host_score_matrix_matches_oracle():
    g = pc.golden()
    o = pc.pf_oracle()
    s3, i3 = capi.host_score_matrix(g["vtml80_kmer16"], 3)
    s2, i2 = capi.host_score_matrix(g["vtml80_kmer16"], 2)
    assert np.array_equal(s3, o.s3) and np.array_equal(i3, o.i3)
    assert np.array_equal(s2, o.s2) and np.array_equal(i2, o.i2)


@pytest.mark.parametrize("spaced", [True, False])
def test_host_index_build_matches_oracle(spaced):
    from oracle.pyoracle import PfOracle
    g = pc.golden()
    base = pc.pf_oracle()
    o = PfOracle.__new__(PfOracle)          # share the score matrices, change the pattern
    o.__dict__.update(base.__dict__)
    o.spaced = int(spaced)
    thr = int(g["kmer_thr"])
    off, ids, pos = o.build_index(g["tres"], g["toff"], thr)
    hoff, hids, hpos = capi.host_index_build(g["tres"], g["toff"], g["vtml80_kmer16"], int(g["k"]), spaced, thr)
    assert np.array_equal(off, hoff) and np.array_equal(ids, hids) and np.array_equal(pos, hpos)
    assert len(ids) > 100000


def test_oracle_fuzz_vs_reference():
    """Where the real reference is available: index, similar k-mers and matchQuery for forced bin counts."""
    from oracle import pyoracle
    if not (pyoracle.ref_available() and pyoracle.ref_matrix_available()):
        pytest.skip("real reference (oracle/_ref + /root/reference/data) not available here")
    ref = pyoracle.RefPrefilter(6)
    km8, um8, km16, pback = ref.matrices()
    g = pc.golden()
    assert np.array_equal(km16, g["vtml80_kmer16"]) and np.array_equal(um8, g["blosum62_ungapped"])
    o = pc.pf_oracle()
    rng = np.random.default_rng(3)
    for _ in range(100):
        kmer = rng.integers(0, 20, 6).astype(np.uint8)
        thr = int(rng.integers(70, 140))
        a, na = o.kmer_list(kmer, thr)
        b, nb = ref.kmer_list(kmer, thr)
        assert na == nb and np.array_equal(a, b)
    # k = 7: (2,2,3) divide strategy
    from oracle.pyoracle import PfOracle, PfGen
    ref7 = pyoracle.RefPrefilter(7)
    o7 = PfOracle.__new__(PfOracle)
    o7.__dict__.update(o.__dict__)
    o7.k = 7
    o7.gen = PfGen(7, o.kalph, o.s3.ctypes.data, o.i3.ctypes.data, o.s2.ctypes.data, o.i2.ctypes.data)
    for _ in range(100):
        kmer = rng.integers(0, 20, 7).astype(np.uint8)
        thr = int(rng.integers(80, 160))
        a, na = o7.kmer_list(kmer, thr)
        b, nb = ref7.kmer_list(kmer, thr)
        assert na == nb and np.array_equal(a, b)
    # k = 5: (2,3) divide strategy (KmerGenerator.cpp:73-86) - the k-mer lists, then whole matchQuery runs on a k = 5 index
    ref5 = pyoracle.RefPrefilter(5)
    o5 = PfOracle.__new__(PfOracle)
    o5.__dict__.update(o.__dict__)
    o5.k = 5
    o5.gen = PfGen(5, o.kalph, o.s3.ctypes.data, o.i3.ctypes.data, o.s2.ctypes.data, o.i2.ctypes.data)
    for _ in range(100):
        kmer = rng.integers(0, 20, 5).astype(np.uint8)
        thr = int(rng.integers(50, 120))
        a, na = o5.kmer_list(kmer, thr)
        b, nb = ref5.kmer_list(kmer, thr)
        assert na == nb and np.array_equal(a, b)
    (qres5, qoff5), (tres5, toff5) = pc.synthetic_case(6, 500, seed=78, planted=0.5)
    thr5 = pyoracle.kmer_threshold(5.7, 5)
    ref5.build_index(tres5, toff5, thr5)
    o5.build_index(tres5, toff5, thr5)
    ro, ri, rp = ref5.index_dump()
    assert np.array_equal(ro, o5.offsets) and np.array_equal(ri, o5.ids[:o5.n_entries])
    swo5 = pyoracle.Oracle()
    from mmseqs2_amd import workloads as wl5
    bins5 = ref5.make_matcher(max_hits=300, force_bins=0)
    for qi, q in enumerate(wl5.split(qres5, qoff5)):
        cb = swo5.comp_bias(km16, pback, q)
        r = ref5.match(q, None)
        x = o5.match(q, cb, bins5, max_hits=300, identity_id=None)
        assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"])
        assert np.array_equal(r["diagonal"], x["diagonal"]) and r["db_matches"] == x["stats"]["db_matches"]
    (qres, qoff), (tres, toff) = pc.synthetic_case(10, 1200, seed=77, planted=0.5)
    thr = pyoracle.kmer_threshold(5.7, 6)
    ref.build_index(tres, toff, thr)
    o.build_index(tres, toff, thr)
    ro, ri, rp = ref.index_dump()
    assert np.array_equal(ro, o.offsets) and np.array_equal(ri, o.ids[:o.n_entries])
    swo = pyoracle.Oracle()
    from mmseqs2_amd import workloads as wl
    for mh, fb in ((300, 0), (12, 32), (5, 2)):
        bins = ref.make_matcher(max_hits=mh, force_bins=fb)
        for qi, q in enumerate(wl.split(qres, qoff)):
            cb = swo.comp_bias(km16, pback, q)
            ident = None if qi % 2 else qi
            r = ref.match(q, ident)
            x = o.match(q, cb, bins, max_hits=mh, identity_id=ident)
            assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"])
            assert np.array_equal(r["diagonal"], x["diagonal"]) and r["db_matches"] == x["stats"]["db_matches"]


def test_oracle_long_sequences_vs_reference():
    """Sequences of 32768 residues or more (UngappedAlignment.cpp:187-312): computeLongScore for the elements of a long query and for
    long targets in batches that are not full, and the full batch of eight elements of one diagonal, where a long target receives the
    long score of ANOTHER element's target or 0 (the restatement's mmo_score_batch says how).  Index, hit lists and statistics of
    the restatement against the reference's own classes; the counters say that every one of those paths ran."""
    import ctypes
    from oracle import pyoracle
    from mmseqs2_amd import workloads as wl
    if not (pyoracle.ref_available() and pyoracle.ref_matrix_available()):
        pytest.skip("real reference (oracle/_ref + /root/reference/data) not available here")
    ref = pyoracle.RefPrefilter(6)
    km8, um8, km16, pback = ref.matrices()
    o = pc.pf_oracle()
    swo = pyoracle.Oracle()
    thr = pyoracle.kmer_threshold(5.7, 6)
    stats = (ctypes.c_uint64 * 4)()
    o.L.mmo_pf_long_stats(stats)
    seen = np.zeros(4, np.int64)
    for seed in (1, 2):
        qs, tl = pc.long_case(seed, long_query=True)
        tres, toff = wl.seqs_from_list(tl)
        ref.build_index(tres, toff, thr)
        o.build_index(tres, toff, thr)
        ro, ri, rp = ref.index_dump()
        assert np.array_equal(ro, o.offsets) and np.array_equal(ri, o.ids[:o.n_entries]) and np.array_equal(rp, o.pos[:o.n_entries])
        for max_hits in (300, 12):
            bins = ref.make_matcher(max_hits=max_hits, force_bins=0, max_seq_len=70000)
            for qi, q in enumerate(qs):
                cb = swo.comp_bias(km16, pback, q)
                r = ref.match(q, None)
                x = o.match(q, cb, bins, max_hits=max_hits, identity_id=None)
                assert x["stats"]["rc"] == 0
                assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]), (seed, max_hits, qi)
                assert np.array_equal(r["diagonal"], x["diagonal"]) and r["db_matches"] == x["stats"]["db_matches"]
                o.L.mmo_pf_long_stats(stats)
                seen += np.array(list(stats), np.int64)
    assert (seen > 0).all(), seen
