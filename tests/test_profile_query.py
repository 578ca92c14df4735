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
def test_device_profile_queries_equal_reference_vectors(gpu):
    import mmseqs2_amd
    from mmseqs2_amd import workloads as wl
    mat = np.load(os.path.join(HERE, "golden", "matrices.npz"))["blosum62_sw"]
    gold = load_golden()
    targets = []
    for _, ts, _, _ in gold:
        targets += ts
    tres, toff = wl.seqs_from_list(targets)
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
def test_device_profile_queries_equal_oracle_random(gpu):
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


@pytest.mark.gpu
def test_device_block_aligner_for_profile_queries_equals_the_restatement(gpu):
    """a15 for PROFILE queries (round 5): mmgpu_sw_block_backtrace on the int16-range pairs of profile queries - the query's score rows
    in place of matrix + bias, the query on the crate's reference side (block_kernel.hip, BkSeq::prof) - against the restated
    alignStartPosBacktraceBlock<PROFILE_SEQ> (oracle/block_oracle.c, itself equal to the reference's compiled glue over the restated
    crate, tests/test_block_oracle.py); a sequence query rides in the same batch."""
    from mmseqs2_amd.capi import host_comp_bias
    mats = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    mat = mats["blosum62_sw"]
    orc = Oracle()
    rng = np.random.default_rng(91)
    cs = cases(rng, mat, n_queries=7)
    targets = [t for _, ts in cs for t in ts]
    from mmseqs2_amd import workloads as wl
    tres, toff = wl.seqs_from_list(targets)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(len(targets), dtype=np.uint32)
    queries = []
    for qi, (e, ts) in enumerate(cs):
        cons = e[:, 20].astype(np.uint8)
        if qi == 3:
            _, cb = host_comp_bias(mat.astype(np.int16), mats["blosum62_pback"], cons)
            queries.append(dict(q=cons, comp_bias=cb, targets=ids, min_start_score=0))
        else:
            prof = (e[:, :20].astype(np.int32) / 4).astype(np.int8).T.copy()
            queries.append(dict(q=cons, comp_bias=None, profile=prof, targets=ids, min_start_score=0))
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    out = b.fetch().reshape(len(queries), len(targets))
    blk, strs = b.block_backtrace(np.arange(out.size, dtype=np.uint32))
    n_prof = n_seq = 0
    for qi, qd in enumerate(queries):
        for k, t in enumerate(targets):
            p = qi * len(targets) + k
            h = out[qi, k]
            if int(h["word"]) != 1 or int(h["score"]) <= 0:
                assert int(blk[p]["status"]) == 3      # MMGPU_BLOCK_NOT_WORD
                continue
            if "profile" in qd:
                prof21 = np.concatenate([qd["profile"], np.zeros((1, qd["profile"].shape[1]), np.int8)])
                w = orc.sw_block_backtrace_profile(prof21, qd["q"], t, 11, 1, int(h["score"]), int(h["q_end"]), int(h["t_end"]))
                n_prof += 1
            else:
                r = orc.block_backtrace(qd["q"], qd["comp_bias"], t, mat, 11, 1, int(h["score"]), int(h["q_end"]), int(h["t_end"]))
                w = dict(q_start=r["q_start"], t_start=r["t_start"], ident=r["ident"], bt=r["bt"]) if r["ok"] else None
                n_seq += 1
            if w is None:
                assert int(blk[p]["status"]) == 1, (qi, k)      # MMGPU_BLOCK_DECLINED: "Block alignment failed"
                continue
            assert int(blk[p]["status"]) == 0, (qi, k, int(blk[p]["status"]))
            assert (int(blk[p]["q_start"]), int(blk[p]["t_start"]), int(blk[p]["ident"]), strs[p]) == (w["q_start"], w["t_start"], w["ident"], w["bt"]), (qi, k)
    assert n_prof >= 10 and n_seq >= 1, (n_prof, n_seq)
    b.free()


PF_PROFILE_THR = 99                                      # getKmerThreshold(5.7, profile, k = 6): 134.35 - 6.15 * 5.7
PF_PROFILE_SETTINGS = [(300, 2), (10, 32), (4, 2)]       # (max_hits, CacheFriendlyOperations bins)
GOLD_PF = os.path.join(HERE, "golden", "profile_pf.npz")


# ---- prefilter of profile queries (QueryMatcher::matchQuery with Sequence::profile_matrix, Prefiltering.cpp:832-834) -----------
def pf_profile_case(seed, n_queries=6, n_targets=1500):
    """profile entries + a target set with planted homologs of their consensus sequences"""
    from mmseqs2_amd import workloads as wl
    mat = np.load(os.path.join(HERE, "golden", "matrices.npz"))["blosum62_sw"]
    rng = np.random.default_rng(seed)
    entries = []
    tl = [rng.choice(20, size=int(rng.integers(40, 500)), p=wl.BACKGROUND).astype(np.uint8) for _ in range(n_targets)]
    for qi in range(n_queries):
        L = int(rng.integers(30, 400))
        e, cons = make_entry(rng, mat, L, sharp=1.6 if qi % 2 else 1.1)     # real profiles score on the x4 scale up to ~60
        entries.append(e)
        for k in rng.choice(n_targets, 25, replace=False):
            tl[k] = mutate(rng, cons, float(rng.uniform(0.35, 0.9)))
    tres, toff = wl.seqs_from_list(tl)
    return entries, tres, toff


@pytest.mark.skipif(not (ref_available() and ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_oracle_profile_prefilter_equals_reference():
    from oracle import pyoracle
    from tests import pf_common as pc
    ref = pyoracle.RefPrefilter(6)
    o = pc.pf_oracle()
    entries, tres, toff = pf_profile_case(31)
    ref.build_index(tres, toff, 0)          # profile searches index every target k-mer (Prefiltering.cpp:555-557)
    o.build_index(tres, toff, 0)
    thr = 99                                # getKmerThreshold(5.7, profile, k = 6): 134.35 - 6.15 * 5.7
    n_hits = n_sat = 0
    for mh, fb in ((300, 0), (10, 32), (4, 2)):
        for qi, e in enumerate(entries):
            ident = None if qi % 2 else qi
            r = ref.match_profile(e, thr, max_hits=mh, force_bins=fb, identity_id=ident)
            assert (r["aln"] == (e[:, :20].astype(np.int32) / 4).astype(np.int8).T).all()
            bins = fb if fb else 2          # a 1500-target database picks CacheFriendlyOperations<2> (QueryMatcher.cpp:460-488)
            x = o.match_profile(r["letters"], r["pscore"], r["pindex"], r["aln"], bins, thr, max_hits=mh, identity_id=ident)
            assert x["stats"]["rc"] == 0 and r["db_matches"] == x["stats"]["db_matches"], (mh, qi)
            assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]) and np.array_equal(r["diagonal"], x["diagonal"]), (mh, qi)
            n_hits += len(r["id"])
            n_sat += int((r["score"] > 255).sum())
    assert n_hits > 200 and n_sat > 5


def load_pf_golden():
    g = np.load(GOLD_PF, allow_pickle=False)
    qs = []
    for qi in range(int(g["n_queries"])):
        qs.append(dict(q=g["letters_%d" % qi], profile_score=g["pscore_%d" % qi], profile_index=g["pindex_%d" % qi],
                       profile=g["aln_%d" % qi], comp_bias=None, identity_id=None if qi % 2 else qi))
    return g, qs


def test_golden_profile_prefilter_against_oracle():
    from tests import pf_common as pc
    g, qs = load_pf_golden()
    o = pc.pf_oracle()
    o.build_index(g["tres"], g["toff"], 0)
    for si, (mh, bins) in enumerate(g["settings"].tolist()):
        for qi, qd in enumerate(qs):
            x = o.match_profile(qd["q"], qd["profile_score"], qd["profile_index"], qd["profile"], bins, int(g["thr"]), max_hits=mh,
                                identity_id=qd["identity_id"])
            exp = g["hits_%d_%d" % (si, qi)]
            assert np.array_equal(x["id"], exp[0]) and np.array_equal(x["score"], exp[1]) and np.array_equal(x["diagonal"], exp[2]), (si, qi)


@pytest.mark.gpu
def test_device_profile_prefilter_equals_reference_vectors(gpu):
    """device prefilter with profile queries = the hit lists the real reference produced (ids, scores, diagonals, order)"""
    import mmseqs2_amd
    from mmseqs2_amd import capi
    from tests import pf_common as pc
    g, qs = load_pf_golden()
    m = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    km16 = m["vtml80_kmer"].astype(np.int16)
    gpu.load_targets(g["tres"], g["toff"], 21)
    s3, i3 = capi.host_score_matrix(km16, 3)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, 0, m["blosum62_ungapped"])     # threshold 0: every target k-mer is indexed
    n_sat = 0
    for si, (mh, bins) in enumerate(g["settings"].tolist()):
        hits, counts, status = gpu.pf_batch(qs, int(g["thr"]), max_hits=mh, ref_bins=bins)[:3]
        for qi in range(len(qs)):
            exp = g["hits_%d_%d" % (si, qi)]
            assert int(status[qi]) == 0
            h = hits[qi][: int(counts[qi])]
            assert np.array_equal(h["id"], exp[0]) and np.array_equal(h["score"], exp[1]) and np.array_equal(h["diagonal"], exp[2]), (si, qi)
            n_sat += int((h["score"] > 255).sum())
    assert n_sat > 5


@pytest.mark.gpu
def test_device_mixed_profile_and_sequence_prefilter_equals_oracle(gpu):
    """profile and sequence queries in one batch, against the oracle, on a larger target set"""
    import mmseqs2_amd
    from mmseqs2_amd import capi, workloads as wl
    from oracle import pyoracle
    from tests import pf_common as pc
    m = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    km16 = m["vtml80_kmer"].astype(np.int16)
    g, qs = load_pf_golden()
    rng = np.random.default_rng(9)
    # sequence queries: consensus of the profiles' letters, mutated
    seqq = [dict(q=mutate(rng, qd["q"], 0.9), identity_id=None) for qd in qs[:4]]
    for d in seqq:
        d["comp_bias"] = capi.host_comp_bias(km16, m["vtml80_pback"], d["q"])[0]
    batch = []
    for a, b_ in zip(qs, seqq + [None] * len(qs)):
        batch.append(a)
        if b_ is not None:
            batch.append(b_)
    gpu.load_targets(g["tres"], g["toff"], 21)
    s3, i3 = capi.host_score_matrix(km16, 3)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, 0, m["blosum62_ungapped"])
    o = pc.pf_oracle()
    o.build_index(g["tres"], g["toff"], 0)
    thr = int(g["thr"])
    hits, counts, status = gpu.pf_batch(batch, thr, max_hits=50, ref_bins=2)[:3]
    for qi, qd in enumerate(batch):
        if qd.get("profile") is not None:
            x = o.match_profile(qd["q"], qd["profile_score"], qd["profile_index"], qd["profile"], 2, thr, max_hits=50, identity_id=qd["identity_id"])
        else:
            o.kmer_thr = thr
            x = o.match(qd["q"], qd["comp_bias"], 2, max_hits=50, identity_id=None)
        h = hits[qi][: int(counts[qi])]
        assert int(status[qi]) == 0
        assert np.array_equal(h["id"], x["id"]) and np.array_equal(h["score"], x["score"]) and np.array_equal(h["diagonal"], x["diagonal"]), qi


def pf_profile_long_case(seed):
    """pf_profile_case plus targets of 32768 residues or more that carry homologs of the profiles' consensus sequences: one at
    1 000 and one at 34 000 of a 40 000-residue target, and - for the batches of eight elements of one diagonal
    (UngappedAlignment.cpp:187-293) - substitution-only copies of profile 3's consensus as ordinary targets and at the start of three
    long ones."""
    from mmseqs2_amd import workloads as wl
    entries, tres, toff = pf_profile_case(seed, n_targets=600)
    rng = np.random.default_rng(seed + 1000)
    tl = wl.split(tres, toff)
    bg = lambda n: rng.choice(20, size=n, p=wl.BACKGROUND).astype(np.uint8)
    cons = [e[:, 20].astype(np.uint8) for e in entries]
    big = bg(40000)
    for k, at in ((0, 1000), (1, 34000)):
        h = wl.mutate(rng, cons[k], 0.85)
        big[at:at + len(h)] = h
    extra = [big]
    for _ in range(int(rng.integers(10, 18))):
        extra.append(wl.mutate(rng, cons[3], 0.9, max_indels=0))
    for n in (33000, 37000, 52000):
        b = bg(n)
        b[:len(cons[3])] = wl.mutate(rng, cons[3], 0.9, max_indels=0)
        extra.append(b)
    tl = tl + extra
    perm = rng.permutation(len(tl))
    tres, toff = wl.seqs_from_list([tl[i] for i in perm])
    return entries, tres, toff


@pytest.mark.skipif(not (ref_available() and ref_matrix_available()), reason="needs oracle/_ref and /root/reference/data")
def test_oracle_profile_prefilter_with_long_targets_equals_reference():
    """profile queries against targets of 32768 residues or more: computeLongScore and the batches of scoreDiagonalAndUpdateHits read
    the profile's own score rows (UngappedAlignment::createProfile's profile branch) - restatement against the reference"""
    import ctypes
    from oracle import pyoracle
    from tests import pf_common as pc
    ref = pyoracle.RefPrefilter(6)
    o = pc.pf_oracle()
    stats = (ctypes.c_uint64 * 4)()
    o.L.mmo_pf_long_stats(stats)
    seen = np.zeros(4, np.int64)
    for seed in (41, 42):
        entries, tres, toff = pf_profile_long_case(seed)
        ref.build_index(tres, toff, 0)
        o.build_index(tres, toff, 0)
        for mh in (300, 8):
            for qi, e in enumerate(entries):
                r = ref.match_profile(e, 99, max_hits=mh, max_seq_len=65535, identity_id=None)
                x = o.match_profile(r["letters"], r["pscore"], r["pindex"], r["aln"], 2, 99, max_hits=mh, identity_id=None)
                assert x["stats"]["rc"] == 0 and r["db_matches"] == x["stats"]["db_matches"], (seed, mh, qi)
                assert np.array_equal(r["id"], x["id"]) and np.array_equal(r["score"], x["score"]) and np.array_equal(r["diagonal"], x["diagonal"]), (seed, mh, qi)
                o.L.mmo_pf_long_stats(stats)
                seen += np.array(list(stats), np.int64)
    assert (seen[1:3] > 0).all(), seen      # long targets in batches that are not full and in full batches


PF_LONG_SEEDS = (41, 42)
PF_LONG_MAX_HITS = (300, 8)
GOLD_PF_LONG = os.path.join(HERE, "golden", "profile_long_pf.npz")


def load_pf_long_golden(seed):
    """the profile arrays the reference derived from the entries of pf_profile_long_case(seed) and its hit lists
    (tests/golden/make_profile_long_pf_golden.py); the targets are regenerated here and checked against the recorded CRC"""
    import zlib
    g = np.load(GOLD_PF_LONG, allow_pickle=False)
    entries, tres, toff = pf_profile_long_case(seed)
    assert zlib.crc32(tres.tobytes()) == int(g["tres_crc_%d" % seed]), "the regenerated targets differ from the recorded ones"
    qs = []
    for qi in range(int(g["n_queries_%d" % seed])):
        qs.append(dict(q=g["letters_%d_%d" % (seed, qi)], profile_score=g["pscore_%d_%d" % (seed, qi)],
                       profile_index=g["pindex_%d_%d" % (seed, qi)].astype(np.uint32), profile=g["aln_%d_%d" % (seed, qi)],
                       comp_bias=None, identity_id=None))
    return g, qs, tres, toff


def test_golden_profile_long_targets_against_oracle():
    """the recorded lists of the reference against the restatement (runs without /root/reference)"""
    from tests import pf_common as pc
    o = pc.pf_oracle()
    for seed in PF_LONG_SEEDS[:1]:
        g, qs, tres, toff = load_pf_long_golden(seed)
        o.build_index(tres, toff, 0)
        for mh in PF_LONG_MAX_HITS:
            for qi, qd in enumerate(qs):
                x = o.match_profile(qd["q"], qd["profile_score"], qd["profile_index"], qd["profile"], 2, 99, max_hits=mh, identity_id=None)
                e = g["hits_%d_%d_%d" % (seed, mh, qi)]
                assert np.array_equal(x["id"], e[0]) and np.array_equal(x["score"], e[1]) and np.array_equal(x["diagonal"], e[2]), (seed, mh, qi)


@pytest.mark.gpu
def test_device_profile_prefilter_with_long_targets_equals_reference_vectors(gpu):
    """the same cases on the device (pf_long_kernel with the profile's score rows) against the lists recorded from the reference:
    no query handed back"""
    from mmseqs2_amd import capi
    m = np.load(os.path.join(HERE, "golden", "matrices.npz"))
    km16 = m["vtml80_kmer"].astype(np.int16)
    s3, i3 = capi.host_score_matrix(km16, 3)
    n_long_hits = 0
    for seed in PF_LONG_SEEDS:
        g, batch, tres, toff = load_pf_long_golden(seed)
        lens = np.diff(toff.astype(np.int64))
        gpu.load_targets(tres, toff, 21)
        gpu.pf_build_index(6, 21, True, s3, i3, km16, 0, m["blosum62_ungapped"])
        for mh in PF_LONG_MAX_HITS:
            hits, counts, status = gpu.pf_batch(batch, 99, max_hits=mh, ref_bins=2)[:3]
            for qi in range(len(batch)):
                e = g["hits_%d_%d_%d" % (seed, mh, qi)]
                h = hits[qi][: int(counts[qi])]
                assert int(status[qi]) == 0, (seed, mh, qi)
                assert np.array_equal(h["id"], e[0]) and np.array_equal(h["score"], e[1]) and np.array_equal(h["diagonal"], e[2]), (seed, mh, qi)
                n_long_hits += int((lens[e[0]] >= 32768).sum())
    assert n_long_hits > 4      # the lists do hold targets of 32768 residues or more
