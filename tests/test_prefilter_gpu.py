"""GPU parity tests of the prefilter (through the C-ABI): every stage of the device pipeline and the final hit_t
lists against the oracle, and the final lists against the golden vectors recorded from the real reference."""
import os

import numpy as np
import pytest

from mmseqs2_amd import capi, workloads as wl
from tests import pf_common as pc
from tests import pf_gpu_check as chk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_case(gpu):
    g = pc.golden()
    orc = pc.pf_oracle()
    orc.build_index(g["tres"], g["toff"], int(g["kmer_thr"]))
    chk.load_case(gpu, g, g["tres"], g["toff"], int(g["kmer_thr"]))
    return g, orc


def test_prefilter_stages_match_oracle(gpu, golden_case):
    g, orc = golden_case
    ok, rep = chk.check(gpu, orc, pc.golden_queries(g), 300, 2, stages=True, label="golden/stages")
    assert ok, "\n".join(rep)


def test_prefilter_hits_match_golden(gpu, golden_case):
    """Final hit lists against the lists the REAL reference produced (tests/golden/prefilter_vectors.npz)."""
    g, orc = golden_case
    qs = pc.golden_queries(g)
    for si, (mh, bins) in enumerate(g["settings"].tolist()):
        hits, counts, status, stats = gpu.pf_batch(qs, int(g["kmer_thr"]), max_hits=mh, min_diag_score=15, ref_bins=bins)
        exp = pc.expected_hits(g, si)
        for qi in range(len(qs)):
            n = int(counts[qi])
            assert status[qi] == 0
            assert int(stats[qi]["db_matches"]) == int(g["db_matches"][qi])
            assert np.array_equal(hits[qi]["id"][:n], exp[qi][0]), (si, qi)
            assert np.array_equal(hits[qi]["score"][:n], exp[qi][1]), (si, qi)
            assert np.array_equal(hits[qi]["diagonal"][:n], exp[qi][2]), (si, qi)


def test_prefilter_larger_db_and_bins(gpu):
    """20 000 targets (5 device bins), ragged query lengths, several max_hits / reference-bin settings."""
    g = pc.golden()
    (qres, qoff), (tres, toff) = pc.synthetic_case(48, 20000, seed=5, planted=0.1)
    orc = pc.pf_oracle()
    thr = int(g["kmer_thr"])
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    from oracle.pyoracle import Oracle
    swo = Oracle()
    qs = []
    for i, q in enumerate(wl.split(qres, qoff)):
        if i % 7 == 3:
            q = q[: 30 + i]
        qs.append(dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q),
                       identity_id=(i * 131) % 20000 if i % 4 == 0 else None))
    for mh, rb, st in ((300, 2, True), (40, 2, False), (40, 64, False), (10, 8, False)):
        ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=st, label="20k/%d/%d" % (mh, rb))
        assert ok, "\n".join(rep)
    # restore the golden case for tests that run after this one in the same module
    orc.build_index(g["tres"], g["toff"], thr)
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_prefilter_empty_and_tiny(gpu, golden_case):
    g, orc = golden_case
    hits, counts, status, stats = gpu.pf_batch([], int(g["kmer_thr"]))
    assert len(counts) == 0
    q = np.array([0, 1, 2], np.uint8)
    hits, counts, status, stats = gpu.pf_batch([dict(q=q, comp_bias=None, identity_id=7)], int(g["kmer_thr"]))
    assert counts[0] == 1 and hits[0]["id"][0] == 7 and hits[0]["score"][0] == 65535


def test_split_merge_on_device(gpu):
    """Two target shards processed one after the other on this GPU, lists handed over in device memory
    (mmgpu_pf_fetch_device), merged by mmgpu_pf_merge_splits: equals the host mirror and the oracle per shard."""
    import torch
    from mmseqs2_amd import capi, distributed as D
    g = pc.golden()
    orc = pc.pf_oracle()
    thr = int(g["kmer_thr"])
    world = 2
    sh, sizes = pc.shards(g, world)
    mh = capi.split_max_hits(300, world)
    qs = pc.golden_queries(g)
    for qd in qs:
        qd["identity_id"] = None
    nq = len(qs)
    dev = torch.device("cuda", 0)
    gh = torch.zeros((world, nq, mh, 3), dtype=torch.int32, device=dev)
    gc = torch.zeros((world, nq), dtype=torch.int32, device=dev)
    host = []
    for r in range(world):
        chk.load_case(gpu, g, sh[r][0], sh[r][1], thr)
        b = gpu.pf_prepare(qs, thr, max_hits=mh, ref_bins=2)
        b.run()
        b.fetch_device(gh[r].data_ptr(), mh, gc[r].data_ptr())
        hits, counts, status, stats = b.fetch()
        host.append((hits.copy(), counts.copy()))
        orc.build_index(sh[r][0], sh[r][1], thr)
        for qi, qd in enumerate(qs):          # each shard against the oracle
            o = orc.match(qd["q"], qd["comp_bias"], 2, max_hits=mh)
            n = int(counts[qi])
            assert np.array_equal(hits[qi]["id"][:n], o["id"]) and np.array_equal(hits[qi]["score"][:n], o["score"])
        b.free()
    out_h = torch.zeros((nq, world * mh, 3), dtype=torch.int32, device=dev)
    out_c = torch.zeros((nq,), dtype=torch.int32, device=dev)
    gpu.pf_merge_splits(gh.data_ptr(), gc.data_ptr(), world, nq, mh, D.shard_id_offsets(sizes), out_h.data_ptr(), out_c.data_ptr())
    gpu.synchronize()
    oh = out_h.cpu().numpy().reshape(nq, world * mh * 3).view(capi.PF_HIT_DTYPE).reshape(nq, world * mh)
    oc = out_c.cpu().numpy()
    off = D.shard_id_offsets(sizes)
    for qi in range(nq):
        exp = capi.merge_hit_lists_host([host[r][0][qi, :host[r][1][qi]] for r in range(world)], off)
        n = int(oc[qi])
        assert n == len(exp)
        assert np.array_equal(oh[qi]["id"][:n], exp["id"]) and np.array_equal(oh[qi]["score"][:n], exp["score"])
        assert np.array_equal(oh[qi]["diagonal"][:n], exp["diagonal"])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_prefilter_100k_targets_properties_and_oracle_sample(gpu):
    """BASELINE configs[2] at 1/10 scale (1000 queries x 100 000 targets, 25 device bins): size-independent properties
    over all queries - the result does not depend on the batch size, lists are sorted by (score desc, id asc), ids are
    unique and in range, counts <= max_hits, db_matches identical across batchings - and a seeded sample of queries
    against the oracle over the full database."""
    g = pc.golden()
    m = g
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(2000, 50, 1000, seed=10)
    thr = int(g["kmer_thr"])
    chk.load_case(gpu, g, tres, toff, thr)
    from mmseqs2_amd import capi
    qs = wl.split(qres, qoff)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q, lib=gpu.L)[0],
                    identity_id=None) for q in qs]
    runs = []
    for bsz in (1000, 173):
        hs, cs, dm = [], [], []
        for i in range(0, len(queries), bsz):
            h, c, st, stats = gpu.pf_batch(queries[i:i + bsz], thr, max_hits=300, ref_bins=2)
            assert np.all(st == 0)
            hs.append(h); cs.append(c); dm.append(stats["db_matches"])
        runs.append((np.concatenate(hs), np.concatenate(cs), np.concatenate(dm)))
    (h0, c0, d0), (h1, c1, d1) = runs
    assert np.array_equal(c0, c1) and np.array_equal(d0, d1)
    nt = len(toff) - 1
    for qi in range(len(qs)):
        n = int(c0[qi])
        assert n <= 300
        a, b = h0[qi][:n], h1[qi][:n]
        assert np.array_equal(a["id"], b["id"]) and np.array_equal(a["score"], b["score"]) and np.array_equal(a["diagonal"], b["diagonal"])
        sc = a["score"].astype(np.int64)
        assert np.all(sc[:-1] >= sc[1:])
        same = sc[:-1] == sc[1:]
        assert np.all(a["id"][:-1][same] < a["id"][1:][same])
        assert len(np.unique(a["id"])) == n and (n == 0 or int(a["id"].max()) < nt)
        assert np.all(sc >= 15)
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    rng = np.random.default_rng(4)
    for qi in rng.choice(len(qs), 12, replace=False):
        o = orc.match(qs[qi], queries[qi]["comp_bias"], 2, max_hits=300)
        n = int(c0[qi])
        assert int(d0[qi]) == o["stats"]["db_matches"]
        assert np.array_equal(h0[qi]["id"][:n], o["id"]) and np.array_equal(h0[qi]["score"][:n], o["score"])
        assert np.array_equal(h0[qi]["diagonal"][:n], o["diagonal"])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_prefilter_k7(gpu):
    """k = 7 (the reference's choice for target sets of 3.35e9 residues and more, IndexTable.h:441-449): the (2,2,3)
    similar-k-mer generator, a 20^7-entry offset table, hit lists against the oracle (itself pinned against the real
    reference for k = 7, scripts/fuzz and tests/test_prefilter_oracle.py)."""
    from oracle.pyoracle import PfOracle, kmer_threshold
    g = pc.golden()
    thr7 = kmer_threshold(5.7, 7)
    base = pc.pf_oracle()
    orc = PfOracle.__new__(PfOracle)
    orc.__dict__.update(base.__dict__)
    orc.k = 7
    import ctypes
    from oracle.pyoracle import PfGen
    orc.gen = PfGen(7, orc.kalph, orc.s3.ctypes.data, orc.i3.ctypes.data, orc.s2.ctypes.data, orc.i2.ctypes.data)
    n = 400
    tres, toff = g["tres"][:int(g["toff"][n])], g["toff"][:n + 1]
    orc.build_index(tres, toff, thr7)
    chk.load_case(gpu, g, tres, toff, thr7, k=7)
    qs = pc.golden_queries(g)[:10]
    for qd in qs:
        qd["identity_id"] = None
    ok, rep = chk.check(gpu, orc, qs, 300, 2, stages=True, label="k7")
    assert ok, "\n".join(rep)
    thr = int(g["kmer_thr"])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_prefilter_k5(gpu):
    """k = 5 (round 6; `-k 5` took the CPU path before): the (2,3) similar-k-mer generator - row A from the 2-mer table -, a 20^5
    offset table, every stage and the hit lists against the oracle (pinned against the real reference for k = 5 in
    tests/test_prefilter_oracle.py); the index is also built on the device and compared with the host builder's."""
    from oracle.pyoracle import PfOracle, kmer_threshold, PfGen
    g = pc.golden()
    thr5 = kmer_threshold(5.7, 5)
    base = pc.pf_oracle()
    orc = PfOracle.__new__(PfOracle)
    orc.__dict__.update(base.__dict__)
    orc.k = 5
    orc.gen = PfGen(5, orc.kalph, orc.s3.ctypes.data, orc.i3.ctypes.data, orc.s2.ctypes.data, orc.i2.ctypes.data)
    n = 600
    tres, toff = g["tres"][:int(g["toff"][n])], g["toff"][:n + 1]
    orc.build_index(tres, toff, thr5)
    try:
        tab = chk.load_case(gpu, g, tres, toff, thr5, k=5)
        assert np.array_equal(tab["offsets"], orc.offsets)
        qs = pc.golden_queries(g)[:12]
        for qd in qs:
            qd["identity_id"] = None
        for mh, rb in ((300, 2), (20, 8)):
            ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=True, label="k5/%d/%d" % (mh, rb))
            assert ok, "\n".join(rep)
        hits_a, counts_a, _, _ = gpu.pf_batch(qs, thr5, max_hits=300, ref_bins=2)
        km16, um8 = g["vtml80_kmer16"], g["blosum62_ungapped"]
        s3, i3 = capi.host_score_matrix(km16, 3, lib=gpu.L)
        s2, i2 = capi.host_score_matrix(km16, 2, lib=gpu.L)
        gpu.pf_build_index(5, 21, True, s3, i3, km16, thr5, um8, score2=s2, index2=i2)
        off_b, ids_b, pos_b = gpu.pf_debug_index(5, 21)
        assert np.array_equal(off_b, tab["offsets"]) and np.array_equal(ids_b, tab["ids"]) and np.array_equal(pos_b, tab["pos"])
        hits_b, counts_b, _, _ = gpu.pf_batch(qs, thr5, max_hits=300, ref_bins=2)
        assert np.array_equal(counts_a, counts_b)
        for qi in range(len(qs)):
            assert np.array_equal(hits_a[qi][:int(counts_a[qi])], hits_b[qi][:int(counts_b[qi])])
    finally:
        chk.load_case(gpu, g, g["tres"], g["toff"], int(g["kmer_thr"]))


def _long_query(rng, tl, ql, homolog_frac):
    parts = []
    while sum(len(p) for p in parts) < ql:
        r = rng.random()
        if r < homolog_frac:
            parts.append(wl.mutate(rng, tl[int(rng.integers(0, len(tl)))], 0.7))
        elif r < 0.8:
            parts.append(rng.choice(20, size=200, p=wl.BACKGROUND).astype(np.uint8))
        else:
            parts.append(np.tile(rng.integers(0, 20, 3).astype(np.uint8), 30))
    return np.concatenate(parts)[:ql]


@pytest.mark.parametrize("nt,ql,frac", [(60000, 30000, 0.5), (100000, 31000, 0.02), (200000, 31000, 0.5)])
def test_prefilter_overflow_path(gpu, nt, ql, frac):
    """Queries that gather >= 2*max(1e6, dbSize) index entries: the reference flushes its databaseHits buffer and merges
    the segments (QueryMatcher.cpp:310-346; 1, 2 and 5 flushes here, saturated and low-score cuts, several CPU bin
    counts).  Device emulation against the oracle (pinned against the real reference for exactly these shapes)."""
    from oracle.pyoracle import Oracle
    g = pc.golden()
    rng = np.random.default_rng(5)
    (qres, qoff), (tres, toff) = wl.config2_align_only(4, nt, 0.2, seed=9)
    thr = int(g["kmer_thr"])
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    swo = Oracle()
    tl = wl.split(tres, toff)
    qs = []
    for q in (_long_query(rng, tl, ql, frac), wl.split(qres, qoff)[0], _long_query(rng, tl, ql - 777, frac)):
        qs.append(dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q), identity_id=None))
    for mh, rb, st in ((300, 2, True), (50, 16, False), (300, 128, False)):
        ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=st, label="overflow/%d/%d/%d" % (nt, mh, rb))
        assert ok, "\n".join(rep)
        assert any("status" not in r for r in rep)
    orc.build_index(g["tres"], g["toff"], thr)
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_index_built_on_device_equals_host_index(gpu):
    """mmgpu_pf_build_index (IndexBuilder::fillDatabase in HBM) against the host builder, which is itself identical to
    the reference's index (tests/test_prefilter_oracle.py): offsets and every (seqId, position) entry, for the golden
    DB, for a DB containing targets with more than 4096 windows (global-scratch path) and a repetitive one (long
    lists), and for k = 7; then a search on the device-built index."""
    from mmseqs2_amd import capi
    g = pc.golden()
    km16, um8 = g["vtml80_kmer16"], g["blosum62_ungapped"]
    thr = int(g["kmer_thr"])
    s3, i3 = capi.host_score_matrix(km16, 3, lib=gpu.L)
    rng = np.random.default_rng(8)
    tl = wl.split(g["tres"], g["toff"])[:300]
    tl.append(rng.choice(20, size=9000, p=wl.BACKGROUND).astype(np.uint8))            # > 4096 windows
    tl.append(np.tile(rng.choice(20, size=37, p=wl.BACKGROUND).astype(np.uint8), 200))  # repeats: few distinct k-mers
    for _ in range(60):                                                                 # one k-mer in many targets
        t = rng.choice(20, size=120, p=wl.BACKGROUND).astype(np.uint8)
        t[40:52] = np.array([9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9], np.uint8)
        tl.append(t)
    big_res, big_off = wl.seqs_from_list(tl)
    for tres, toff, k, kthr in ((g["tres"], g["toff"], 6, thr), (big_res, big_off, 6, thr), (big_res, big_off, 6, 0),
                                (g["tres"][:int(g["toff"][200])], g["toff"][:201], 7, 122)):
        gpu.load_targets(tres, toff, 21)
        s2, i2 = capi.host_score_matrix(km16, 2, lib=gpu.L) if k == 7 else (None, None)
        gpu.pf_build_index(k, 21, True, s3, i3, km16, kthr, um8, score2=s2, index2=i2)
        off, ids, pos = gpu.pf_debug_index(k, 21)
        hoff, hids, hpos = capi.host_index_build(tres, toff, km16, k, True, kthr, lib=gpu.L)
        assert np.array_equal(off, hoff), (k, kthr)
        assert np.array_equal(ids, hids) and np.array_equal(pos, hpos), (k, kthr)
        assert len(ids) > 1000
    # search on a device-built index: golden hit lists
    gpu.load_targets(g["tres"], g["toff"], 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    qs = pc.golden_queries(g)
    hits, counts, status, stats = gpu.pf_batch(qs, thr, max_hits=300, ref_bins=2)
    exp = pc.expected_hits(g, 0)
    for qi in range(len(qs)):
        n = int(counts[qi])
        assert np.array_equal(hits[qi]["id"][:n], exp[qi][0]) and np.array_equal(hits[qi]["score"][:n], exp[qi][1])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_fused_prefilter_to_align_handover(gpu, matrices, oracle):
    """mmgpu_sw_prepare_from_pf: the alignment of the prefilter's hit lists without the lists leaving the device gives,
    slot by slot, what the two-call path through the host gives (and therefore what the oracle gives), incl. backtraces."""
    from mmseqs2_amd import capi
    g = pc.golden()
    thr = int(g["kmer_thr"])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)
    qs = pc.golden_queries(g)
    for qd in qs:
        qd["identity_id"] = None
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    pfb = gpu.pf_prepare(qs, thr, max_hits=300, ref_bins=2)
    pfb.run()
    hits, counts, status, _ = pfb.fetch()
    swq = [dict(q=qd["q"], comp_bias=capi.host_comp_bias(sub16, matrices["blosum62_pback"], qd["q"], lib=gpu.L)[1],
                min_start_score=40) for qd in qs]
    fused = gpu.sw_prepare_from_pf(mat, 11, 1, swq, pfb, mode=1)
    # the alignment batch owns copies of what it needs (list lengths, slot -> target): the prefilter batch may be
    # re-run with other queries or freed before the alignment runs (it used to keep raw pointers into it)
    max_hits = pfb.max_hits
    pfb.free()
    other = gpu.pf_prepare(qs[::-1], thr, max_hits=300, ref_bins=2)
    other.run()
    fused.run()
    fr = fused.fetch().reshape(len(qs), max_hits)
    assert fused.pairs == int(counts.sum())
    host_q = [dict(q=x["q"], comp_bias=x["comp_bias"], targets=hits[i]["id"][:counts[i]].copy(), min_start_score=40)
              for i, x in enumerate(swq)]
    sep = gpu.sw_prepare(mat, 11, 1, host_q, mode=1)
    sep.run()
    sr = sep.fetch()
    assert sep.cells == fused.cells
    off = 0
    pick_f, pick_s = [], []
    for i in range(len(qs)):
        n = int(counts[i])
        a, b = fr[i, :n], sr[off:off + n]
        for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
            assert np.array_equal(a[f], b[f]), (i, f)
        assert np.all(fr[i, n:]["score"] == 0)
        for k in range(min(n, 3)):
            pick_f.append(i * max_hits + k)
            pick_s.append(off + k)
        off += n
    fi, fs = fused.traceback(np.array(pick_f, np.uint32))
    si, ss = sep.traceback(np.array(pick_s, np.uint32))
    assert fs == ss and np.array_equal(fi["ident"], si["ident"]) and np.array_equal(fi["status"], si["status"])
    assert sum(1 for x in fs if x) > 20
    for b in (fused, sep, other):
        b.free()


def test_fused_handover_with_lists_above_4096(gpu, matrices):
    """mmgpu_sw_prepare_from_pf with --max-seqs above 4096 (lists up to 16384 are ordered in LDS by sw_from_pf_kernel; the prefilter's
    select sorts such lists in global scratch): slot by slot what the two-call path through the host gives"""
    from mmseqs2_amd import capi
    g = pc.golden()
    thr = int(g["kmer_thr"])
    rng = np.random.default_rng(17)
    (qres, qoff), (tres, toff) = pc.synthetic_case(3, 3000, seed=21, planted=0.1)
    qs0 = wl.split(qres, qoff)
    tl = wl.split(tres, toff) + [wl.mutate(rng, qs0[0], float(rng.uniform(0.6, 0.95))) for _ in range(5200)]
    tl = [tl[i] for i in rng.permutation(len(tl))]
    tres, toff = wl.seqs_from_list(tl)
    chk.load_case(gpu, g, tres, toff, thr)
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    qs = [dict(q=q, comp_bias=capi.host_comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs0]
    pfb = gpu.pf_prepare(qs, thr, max_hits=6000, ref_bins=2)
    pfb.run()
    hits, counts, status, _ = pfb.fetch()
    assert int(counts[0]) > 4096 and (status == 0).all()
    swq = [dict(q=qd["q"], comp_bias=capi.host_comp_bias(sub16, matrices["blosum62_pback"], qd["q"], lib=gpu.L)[1], min_start_score=40) for qd in qs]
    fused = gpu.sw_prepare_from_pf(mat, 11, 1, swq, pfb, mode=1)
    max_hits = pfb.max_hits
    fused.run()
    fr = fused.fetch().reshape(len(qs), max_hits)
    host_q = [dict(q=x["q"], comp_bias=x["comp_bias"], targets=hits[i]["id"][:counts[i]].copy(), min_start_score=40) for i, x in enumerate(swq)]
    sep = gpu.sw_prepare(mat, 11, 1, host_q, mode=1)
    sep.run()
    sr = sep.fetch()
    assert sep.cells == fused.cells and fused.pairs == int(counts.sum())
    off = 0
    for i in range(len(qs)):
        n = int(counts[i])
        a, b = fr[i, :n], sr[off:off + n]
        for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
            assert np.array_equal(a[f], b[f]), (i, f)
        assert np.all(fr[i, n:]["score"] == 0)
        off += n
    for b in (fused, sep, pfb):
        b.free()
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_max_seqs_above_4096(gpu):
    """--max-seqs above 4096 (unsupported before): the final sort of a list runs in global scratch instead of LDS.
    One family of 7000 members so that a query collects thousands of hits; truncation (5000 of ~7000) and the full list."""
    from mmseqs2_amd import workloads as wl
    g = pc.golden()
    rng = np.random.default_rng(12)
    base = rng.choice(20, size=260, p=wl.BACKGROUND).astype(np.uint8)
    tl = [wl.mutate(rng, base, float(rng.uniform(0.55, 0.95))) for _ in range(7000)]
    tl += [rng.choice(20, size=int(rng.integers(80, 400)), p=wl.BACKGROUND).astype(np.uint8) for _ in range(1500)]
    tres, toff = wl.seqs_from_list(tl)
    km16 = g["vtml80_kmer16"]
    thr = int(g["kmer_thr"])
    s3, i3 = capi.host_score_matrix(km16, 3)
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, g["blosum62_ungapped"])
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    qs = [base, wl.mutate(rng, base, 0.8), tl[7100]]
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q)[0], identity_id=None) for q in qs]
    for mh in (5000, 8500):
        hits, counts, status, _ = gpu.pf_batch(queries, thr, max_hits=mh, ref_bins=2)
        for qi, qd in enumerate(queries):
            x = orc.match(qd["q"], qd["comp_bias"], 2, max_hits=mh)
            assert int(status[qi]) == 0
            h = hits[qi][: int(counts[qi])]
            assert len(x["id"]) == len(h), (mh, qi, len(x["id"]), len(h))
            assert np.array_equal(h["id"], x["id"]) and np.array_equal(h["score"], x["score"]) and np.array_equal(h["diagonal"], x["diagonal"]), (mh, qi)
        assert int(counts[0]) > 4096


def test_stage_chunks_give_the_same_lists(gpu, monkeypatch):
    """Stages 2-3 of a batch run over chunks of queries that share one candidate / survivor array (MMGPU_PF_STAGE_GB, 16 GB
    by default: one chunk for every test batch).  Forced down to one query per chunk and to a few queries per chunk, with
    overflow-path queries at chunk borders: same lists, same statistics; the unchunked run is checked against the oracle."""
    from oracle.pyoracle import Oracle
    g = pc.golden()
    rng = np.random.default_rng(21)
    (qres, qoff), (tres, toff) = wl.config2_align_only(12, 60000, 0.2, seed=19)
    thr = int(g["kmer_thr"])
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    swo = Oracle()
    tl = wl.split(tres, toff)
    ql = wl.split(qres, qoff)
    seqs = [ql[0], _long_query(rng, tl, 30000, 0.5)] + ql[1:7] + [_long_query(rng, tl, 29000, 0.5), ql[0][:5].copy()] + ql[7:]
    qs = [dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q), identity_id=None) for q in seqs]
    monkeypatch.delenv("MMGPU_PF_STAGE_GB", raising=False)
    ok, rep = chk.check(gpu, orc, qs, 300, 2, stages=False, label="one chunk")
    assert ok, "\n".join(rep)

    def run():
        b = gpu.pf_prepare(qs, thr, max_hits=300, ref_bins=2)
        b.run()
        out = b.fetch()
        ms = b.stage_ms()
        b.free()
        return out, ms

    (h0, c0, s0, t0), ms0 = run()
    assert int(c0.sum()) > 0 and int(s0[1]) == 0 and int(s0[8]) == 0      # the overflow queries ran on the device
    entries = int(sum(int(x["db_matches"]) for x in t0))
    for gb in (1e-7, 32.0 * entries / 5 / 2 ** 30):        # every query alone; ~5 chunks
        monkeypatch.setenv("MMGPU_PF_STAGE_GB", repr(gb))
        (h1, c1, s1, t1), ms1 = run()
        assert np.array_equal(c0, c1) and np.array_equal(s0, s1)
        assert np.array_equal(t0, t1)
        for qi in range(len(qs)):
            n = int(c0[qi])
            assert np.array_equal(h0[qi][:n], h1[qi][:n]), (gb, qi)
        assert len(ms1) == 7 and all(x >= 0 for x in ms1)
    monkeypatch.delenv("MMGPU_PF_STAGE_GB", raising=False)
    orc.build_index(g["tres"], g["toff"], thr)
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_candidate_sets_of_half_the_diagonal_array_go_to_the_host(gpu, golden_case, monkeypatch):
    """QueryMatcher.cpp:188,204-214: when keepMaxScoreElementOnly leaves foundDiagonalsSize / 2 = max(1M, dbSize) / 2 elements
    or more, the reference filters and sorts with an unstable std::sort instead of the radix pass, without rescoring - a branch
    the device does not restate.  The double-diagonal candidate total bounds that size from above; queries that reach the
    limit come back MMGPU_PF_SAT_TIE (the host runs matchQuery for them), the others are untouched.  The limit is lowered to
    the median candidate count of the golden queries (MMGPU_PF_SORT_CAP) so that a small database reaches the branch."""
    g, orc = golden_case
    qs = pc.golden_queries(g)
    thr = int(g["kmer_thr"])
    monkeypatch.delenv("MMGPU_PF_SORT_CAP", raising=False)
    h0, c0, s0, _ = gpu.pf_batch(qs, thr, max_hits=300, min_diag_score=15, ref_bins=2)
    assert int((s0 != 0).sum()) == 0
    cands = np.array([orc.match(q["q"], q.get("comp_bias"), 2, max_hits=300, min_diag_score=15,
                                identity_id=q.get("identity_id"))["stats"]["double_hits"] for q in qs], np.int64)
    cap = int(np.median(cands))
    assert cands.min() < cap <= cands.max()
    monkeypatch.setenv("MMGPU_PF_SORT_CAP", str(cap))
    h1, c1, s1, _ = gpu.pf_batch(qs, thr, max_hits=300, min_diag_score=15, ref_bins=2)
    monkeypatch.delenv("MMGPU_PF_SORT_CAP", raising=False)
    for qi in range(len(qs)):
        if cands[qi] >= cap:
            assert s1[qi] == 3 and c1[qi] == 0, (qi, int(cands[qi]), cap)      # MMGPU_PF_SAT_TIE
        else:
            assert s1[qi] == 0 and c1[qi] == c0[qi] and np.array_equal(h1[qi][:c1[qi]], h0[qi][:c0[qi]]), qi


def test_lds_applies_the_lanes_of_one_atomic_in_lane_order(gpu, tmp_path):
    """What the replay kernel rests on since round 6 (pf_kernels.hip, lds_byte_exchange): lanes of ONE ds_mskor_rtn_b32 / ds_or_rtn_b32
    that name the same LDS word are applied in ascending lane order, so that `prev = tmp[id]; tmp[id] = diagonal`
    (CacheFriendlyOperations.cpp:194-208) for the 64 entries of a round is one instruction whatever targets repeat inside it.  The
    probe replays 16.7 M lanes under every degree of conflict (2 ... 4096 distinct keys) against the sequential loop on the host."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "probes", "lds_atomic_order.hip")
    exe = str(tmp_path / "lds_atomic_order")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-o", exe, src], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "mismatches 0, test-and-set mismatches 0" in r.stdout, r.stdout


def test_replay_of_buckets_full_of_repeated_targets(gpu):
    """Rounds in which nearly every lane names a target another lane names too (few targets, nearly all of them planted homologs of the
    queries: hit on every query position): every stage against the oracle.  (With segments - databaseHits flushes - the same is
    test_prefilter_overflow_path's long queries.)"""
    from oracle.pyoracle import Oracle
    swo = Oracle()
    g = pc.golden()
    thr = int(g["kmer_thr"])
    (qres, qoff), (tres, toff) = pc.synthetic_case(24, 600, seed=77, planted=0.9)
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    qs = [dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q), identity_id=None) for q in wl.split(qres, qoff)]
    try:
        for mh, rb in ((300, 2), (40, 8)):
            ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=True, label="dense repeats/%d/%d" % (mh, rb))
            assert ok, "\n".join(rep)
    finally:
        orc.build_index(g["tres"], g["toff"], thr)
        chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_compact_offset_table_with_long_lists(gpu):
    """The similar-k-mer kernels look lists up in the compact offset table (blocks of 28 k-mers: base + one-byte lengths); a
    block holding a list of 255 entries or more is answered by the full table.  400 targets share a 14-residue motif (lists of
    400), 250 another one (lists just below the limit): every stage against the oracle, with queries that contain the motifs,
    and the same lists with the compact table switched off."""
    import os
    g = pc.golden()
    rng = np.random.default_rng(31)
    thr = int(g["kmer_thr"])
    tl = wl.split(g["tres"], g["toff"])[:600]
    motif_a = rng.choice(20, size=14, p=wl.BACKGROUND).astype(np.uint8)
    motif_b = rng.choice(20, size=14, p=wl.BACKGROUND).astype(np.uint8)
    for i in range(400):
        t = rng.choice(20, size=int(rng.integers(60, 300)), p=wl.BACKGROUND).astype(np.uint8)
        a = int(rng.integers(0, len(t) - 14))
        t[a:a + 14] = motif_a
        tl.append(t)
    for i in range(250):
        t = rng.choice(20, size=int(rng.integers(60, 300)), p=wl.BACKGROUND).astype(np.uint8)
        a = int(rng.integers(0, len(t) - 14))
        t[a:a + 14] = motif_b
        tl.append(t)
    tres, toff = wl.seqs_from_list(tl)
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    tab = chk.load_case(gpu, g, tres, toff, thr)
    lens = np.diff(tab["offsets"].astype(np.int64))
    assert lens.max() >= 400 and np.any((lens >= 200) & (lens < 255))
    from oracle.pyoracle import Oracle
    swo = Oracle()
    qs = []
    for m in (motif_a, motif_b, np.concatenate([motif_a, motif_b])):
        q = rng.choice(20, size=180, p=wl.BACKGROUND).astype(np.uint8)
        q[50:50 + len(m)] = m
        qs.append(dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q), identity_id=None))
    qs += pc.golden_queries(g)[:4]
    for qd in qs:
        qd["identity_id"] = None
    ok, rep = chk.check(gpu, orc, qs, 300, 2, stages=True, label="compact offsets / long lists")
    assert ok, "\n".join(rep)
    hits, counts, _, _ = gpu.pf_batch(qs, thr, max_hits=300, ref_bins=2)
    os.environ["MMGPU_PF_COFS"] = "0"
    try:
        chk.load_case(gpu, g, tres, toff, thr)
        hits0, counts0, _, _ = gpu.pf_batch(qs, thr, max_hits=300, ref_bins=2)
    finally:
        del os.environ["MMGPU_PF_COFS"]
    assert np.array_equal(counts, counts0)
    for qi in range(len(qs)):
        n = int(counts[qi])
        assert np.array_equal(hits[qi][:n], hits0[qi][:n])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_batch_with_more_than_2_32_index_entries_and_tile_slots(gpu):
    """One batch whose queries meet more than 2^32 index entries (the split tiles then hold more than 2^32 slots): the 32-bit
    places of the replay kernel are relative to the query's own tiles and the 32-bit candidate bases only meet inside a stage
    chunk, so the batch runs as one.  2 000 targets that are lightly mutated copies of one sequence, every query that sequence:
    each query meets ~10^6 entries, as many queries as it takes to pass 2^32 + 10 %.  The first query is checked against the
    oracle stage by stage; every other query - those standing beyond the 2^32nd entry among them - must give the same list."""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    rng = np.random.default_rng(5)
    base = rng.choice(20, size=220, p=wl.BACKGROUND).astype(np.uint8)
    tl = [wl.mutate(rng, base, 0.9) for _ in range(2000)]
    tres, toff = wl.seqs_from_list(tl)
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    try:
        one = [dict(q=base, comp_bias=None, identity_id=None)]
        ok, rep = chk.check(gpu, orc, one, 300, 2, stages=True, label="one query of the large batch")
        assert ok, "\n".join(rep)
        _, _, _, stats = gpu.pf_batch(one, thr, max_hits=300, ref_bins=2)
        per_query = int(stats[0]["db_matches"])
        assert 1e5 < per_query < 2e6      # (below maxDbMatches: the ordinary path)
        nq = int(1.1 * 2 ** 32 / per_query) + 1
        hits, counts, status, stats = gpu.pf_batch(one * nq, thr, max_hits=300, ref_bins=2)
        assert int(stats["db_matches"].astype(np.int64).sum()) > 2 ** 32
        assert np.all(status == 0) and np.all(counts == counts[0]) and counts[0] > 0
        n = int(counts[0])
        for f in ("id", "score", "diagonal"):
            assert np.all(hits[:, :n][f] == hits[0, :n][f][None, :]), f
    finally:
        orc.build_index(g["tres"], g["toff"], thr)
        chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_targets_of_32768_residues_or_more(gpu):
    """Candidates on targets of 32768 residues or more are scored on the device (pf_long_kernel): computeLongScore over every
    65536-shift of the 16-bit diagonal, and the reference's batches of eight elements of one diagonal, where a long target of a
    FULL batch takes the long score of another element's target or 0 (UngappedAlignment.cpp:187-312).  The restatement is pinned
    against the reference on cases of the same make (tests/test_prefilter_oracle.py); its counters say the cases run every path.
    A query of 32768 residues or more: computeLongScore for every element (pf_longq_kernel).  No query is handed back; the index
    built on the device equals the host builder's."""
    import ctypes
    from mmseqs2_amd import capi
    from oracle.pyoracle import Oracle
    g = pc.golden()
    km16, um8 = g["vtml80_kmer16"], g["blosum62_ungapped"]
    thr = int(g["kmer_thr"])
    orc = pc.pf_oracle()
    swo = Oracle()
    stats = (ctypes.c_uint64 * 4)()
    orc.L.mmo_pf_long_stats(stats)
    seen = np.zeros(4, np.int64)
    s3, i3 = capi.host_score_matrix(km16, 3, lib=gpu.L)
    for seed in (1, 2, 5):
        qs, tl = pc.long_case(seed, long_query=seed != 1, device=True)      # (seeds 2, 5: with a 33 000-residue query)
        tres, toff = wl.seqs_from_list(tl)
        orc.build_index(tres, toff, thr)
        tab = chk.load_case(gpu, g, tres, toff, thr)
        queries = [dict(q=q, comp_bias=swo.comp_bias(km16, g["vtml80_pback"], q), identity_id=None) for q in qs]
        for mh, rb in ((300, 2), (12, 16)):
            ok, rep = chk.check(gpu, orc, queries, mh, rb, stages=False, label="long/%d/%d/%d" % (seed, mh, rb))
            assert ok, "\n".join(rep)
            orc.L.mmo_pf_long_stats(stats)
            seen += np.array(list(stats), np.int64)
        if seed == 1:
            gpu.load_targets(tres, toff, 21)
            gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
            off, ids, pos = gpu.pf_debug_index(6, 21)
            assert np.array_equal(off, tab["offsets"]) and np.array_equal(ids, tab["ids"]) and np.array_equal(pos, tab["pos"])
            hits, counts, status, _ = gpu.pf_batch(queries, thr, max_hits=300, ref_bins=2)
            for qi, qd in enumerate(queries):
                o = orc.match(qd["q"], qd["comp_bias"], 2, max_hits=300, identity_id=None)
                n = int(counts[qi])
                assert status[qi] == 0 and np.array_equal(hits[qi]["id"][:n], o["id"]) and np.array_equal(hits[qi]["score"][:n], o["score"])
    assert (seen > 0).all(), seen      # elements of a long query; long targets in batches that are not full, in full batches, scored by another element's target
    # more candidates on the diagonals of long targets than pf_long_kernel pools (4096): that query - and only that one - is
    # handed back with MMGPU_PF_LONG_SEQ
    rng = np.random.default_rng(9)
    qs, tl = pc.long_case(1, device=True)
    tl = tl + [wl.mutate(rng, qs[3], 0.9, max_indels=0) for _ in range(4200)]
    tres, toff = wl.seqs_from_list(tl)
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    queries = [dict(q=q, comp_bias=swo.comp_bias(km16, g["vtml80_pback"], q), identity_id=None) for q in qs]
    hits, counts, status, _ = gpu.pf_batch(queries, thr, max_hits=300, ref_bins=2)
    for qi, qd in enumerate(queries):
        if qi == 3:
            assert status[qi] == capi.PF_LONG_SEQ
            continue
        o = orc.match(qd["q"], qd["comp_bias"], 2, max_hits=300, identity_id=None)
        n = int(counts[qi])
        assert status[qi] == 0 and np.array_equal(hits[qi]["id"][:n], o["id"]) and np.array_equal(hits[qi]["score"][:n], o["score"])
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)
