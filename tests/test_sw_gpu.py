"""GPU parity tests (pytest -m gpu): the HIP Smith-Waterman path, called through the C-ABI, against the
plain-C oracle on the same seeded inputs and against the golden vectors generated from the real reference.
Bar: bit-exact integers (score, q_end, t_end, q_start, t_start, word)."""
import numpy as np
import pytest

from mmseqs2_amd import workloads as wl
from tests.rescore import rescore

pytestmark = pytest.mark.gpu

GO, GE = 11, 1


def _round_cb(oracle, matrices, q):
    sub = matrices["blosum62_sw"].astype(np.int16)
    return oracle.round_comp_bias(oracle.comp_bias(sub, matrices["blosum62_pback"], q, 1.0))


def _check(gpu_hits, oracle, mat, q, cb, tres, toff, ids, with_start, tag):
    toff = toff.astype(np.int64)
    for k, tid in enumerate(ids):
        t = tres[toff[tid]:toff[tid + 1]]
        r = oracle.sw_align(q, cb, t, mat, GO, GE, need_start=with_start)
        h = gpu_hits[k]
        got = (int(h["score"]), int(h["q_end"]), int(h["t_end"]), int(h["word"]))
        exp = (r["score"], r["q_end"], r["t_end"], r["word"])
        assert got == exp, (tag, k, int(tid), got, exp)
        if with_start and r["score"] > 0:
            assert (int(h["q_start"]), int(h["t_start"])) == (r["q_start"], r["t_start"]), (tag, k, int(tid))


def test_golden_vectors(gpu, oracle, matrices, sw_vectors):
    """Every golden pair (reference-generated): forward and reverse scans."""
    v = sw_vectors
    mat = matrices["blosum62_sw"]
    gpu.load_targets(v["tres"], v["toff"], 21)
    qoff = v["qoff"].astype(np.int64)
    queries = []
    for i in range(len(qoff) - 1):
        queries.append(dict(q=v["qres"][qoff[i]:qoff[i + 1]], comp_bias=v["cb"][qoff[i]:qoff[i + 1]],
                            targets=np.array([i], np.uint32), min_start_score=0))
    out = gpu.sw_batch(mat, GO, GE, queries, mode=1)
    exp = v["expect"]
    for i in range(len(queries)):
        h = out[i]
        got = [int(h["score"]), int(h["q_end"]), int(h["t_end"]), int(h["q_start"]), int(h["t_start"]), int(h["word"])]
        assert got == list(exp[i][:6]), (i, got, list(exp[i][:6]))


@pytest.mark.parametrize("qlen", [1, 5, 20, 33, 64, 70, 90, 97, 128, 129, 150, 161, 185, 200, 215, 235, 256, 265, 275, 300, 315, 330, 340,
                                  360, 384, 385, 410, 420, 440, 448, 470, 512])
def test_single_tile_classes_vs_oracle(gpu, oracle, matrices, qlen):
    """One query per kernel instantiation (round 5: every R = 1 .. 28 rows per lane, tiles of 16 R rows; 470 and 512 are cut into
    two tiles) against 300 ragged targets incl. homologs."""
    rng = np.random.default_rng(100 + qlen)
    mat = matrices["blosum62_sw"]
    q = rng.choice(20, size=qlen, p=wl.BACKGROUND).astype(np.uint8)
    cb = _round_cb(oracle, matrices, q)
    tl = []
    for k in range(300):
        if k % 4 == 0 and qlen > 12:
            tl.append(wl.mutate(rng, q, float(rng.uniform(0.3, 0.95))))
        else:
            tl.append(rng.choice(21, size=int(rng.integers(1, 700)), p=np.append(wl.BACKGROUND * 0.98, 0.02)).astype(np.uint8))
    tres, toff = wl.seqs_from_list(tl)
    gpu.load_targets(tres, toff, 21)
    ids = rng.permutation(300).astype(np.uint32)
    out = gpu.sw_batch(mat, GO, GE, [dict(q=q, comp_bias=cb, targets=ids, min_start_score=0)], mode=1)
    _check(out, oracle, mat, q, cb, tres, toff, ids, True, "qlen%d" % qlen)


@pytest.mark.parametrize("qlen", [449, 500, 513, 560, 600, 640, 670, 700, 730, 760, 790, 830, 880, 1025, 1400, 2500])
def test_multi_tile_vs_oracle(gpu, oracle, matrices, qlen):
    rng = np.random.default_rng(200 + qlen)
    mat = matrices["blosum62_sw"]
    q = rng.choice(20, size=qlen, p=wl.BACKGROUND).astype(np.uint8)
    cb = _round_cb(oracle, matrices, q)
    tl = []
    for k in range(70):
        if k % 3 == 0:
            h = wl.mutate(rng, q, float(rng.uniform(0.3, 0.95)))
            a = int(rng.integers(0, len(h) // 2))
            tl.append(h[a:a + int(rng.integers(20, len(h)))])
        else:
            tl.append(rng.choice(20, size=int(rng.integers(1, 1500)), p=wl.BACKGROUND).astype(np.uint8))
    tres, toff = wl.seqs_from_list(tl)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(70, dtype=np.uint32)
    out = gpu.sw_batch(mat, GO, GE, [dict(q=q, comp_bias=cb, targets=ids, min_start_score=0)], mode=1)
    _check(out, oracle, mat, q, cb, tres, toff, ids, True, "multi%d" % qlen)


def test_many_long_queries_and_one_very_long_target(gpu, oracle, matrices):
    """The column scratch of the multi-tile jobs is a pool sized by resident workgroups, not by jobs: 600 queries of
    600..2200 residues, each with a list that contains a 60 000-residue target, would need 600 x 17 MB per-job scratch
    (the round-1 sizing) - the pool needs a few hundred MB.  Results against the oracle for a sample of the pairs."""
    rng = np.random.default_rng(77)
    mat = matrices["blosum62_sw"]
    tl = [rng.choice(20, size=60000, p=wl.BACKGROUND).astype(np.uint8)]
    qs = [rng.choice(20, size=int(rng.integers(600, 2200)), p=wl.BACKGROUND).astype(np.uint8) for _ in range(600)]
    for i in range(40):
        tl.append(wl.mutate(rng, qs[i], 0.6))
    for _ in range(60):
        tl.append(rng.choice(20, size=int(rng.integers(50, 900)), p=wl.BACKGROUND).astype(np.uint8))
    # plant a homolog of query 3 far inside the long target
    hom = wl.mutate(rng, qs[3], 0.8)
    tl[0][41000:41000 + len(hom)] = hom
    tres, toff = wl.seqs_from_list(tl)
    gpu.load_targets(tres, toff, 21)
    queries = []
    for i, q in enumerate(qs):
        ids = np.concatenate([[0], rng.integers(1, len(tl), 9)]).astype(np.uint32)
        queries.append(dict(q=q, comp_bias=None, targets=ids, min_start_score=0))
    out = gpu.sw_batch(mat, GO, GE, queries, mode=1).reshape(len(qs), 10)
    for i in [0, 3, 7, 123, 599]:
        _check(out[i], oracle, mat, qs[i], None, tres, toff, queries[i]["targets"], True, "pool%d" % i)
    assert out[3, 0]["t_end"] > 41000 and out[3, 0]["score"] > 500


def test_many_queries_ragged_lists_and_empty(gpu, oracle, matrices):
    """Several queries of different classes in one batch, lists of 0, 1, 33 and 600 targets, duplicate ids."""
    rng = np.random.default_rng(7)
    mat = matrices["blosum62_sw"]
    tres, toff = wl.random_seqs(rng, 400, 200, 120, min_len=1)
    gpu.load_targets(tres, toff, 21)
    queries, meta = [], []
    for qlen, nt in [(40, 0), (90, 1), (200, 33), (360, 600), (520, 17), (33, 64)]:
        q = rng.choice(20, size=qlen, p=wl.BACKGROUND).astype(np.uint8)
        cb = _round_cb(oracle, matrices, q) if qlen != 90 else None
        ids = rng.integers(0, 400, nt).astype(np.uint32)
        queries.append(dict(q=q, comp_bias=cb, targets=ids, min_start_score=0))
        meta.append((q, cb, ids))
    out = gpu.sw_batch(mat, GO, GE, queries, mode=0)
    assert len(out) == sum(len(m[2]) for m in meta)
    pos = 0
    for qi, (q, cb, ids) in enumerate(meta):
        _check(out[pos:pos + len(ids)], oracle, mat, q, cb, tres, toff, ids, False, "q%d" % qi)
        assert np.all(out[pos:pos + len(ids)]["q_start"] == -1)
        pos += len(ids)


def test_min_start_score_gates_reverse_scan(gpu, oracle, matrices):
    rng = np.random.default_rng(9)
    mat = matrices["blosum62_sw"]
    q = rng.choice(20, size=250, p=wl.BACKGROUND).astype(np.uint8)
    tl = [wl.mutate(rng, q, 0.8) if k % 2 else rng.choice(20, size=250, p=wl.BACKGROUND).astype(np.uint8) for k in range(40)]
    tres, toff = wl.seqs_from_list(tl)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(40, dtype=np.uint32)
    out = gpu.sw_batch(mat, GO, GE, [dict(q=q, comp_bias=None, targets=ids, min_start_score=100)], mode=1)
    for k in range(40):
        if out[k]["score"] >= 100:
            r = oracle.sw_align(q, None, tl[k], mat, GO, GE, need_start=True)
            assert (int(out[k]["q_start"]), int(out[k]["t_start"])) == (r["q_start"], r["t_start"])
        else:
            assert out[k]["q_start"] == -1 and out[k]["t_start"] == -1
    assert (out["score"] >= 100).sum() >= 15 and (out["score"] < 100).sum() >= 15


def test_int16_saturation_matches_word_pass(gpu, oracle, matrices):
    """A 4000-residue self hit scores > 32767 raw: sw_sse2_word saturates at 32767 and so must we."""
    rng = np.random.default_rng(3)
    mat = matrices["blosum62_sw"]
    q = rng.choice(20, size=7000, p=wl.BACKGROUND).astype(np.uint8)
    tres, toff = wl.seqs_from_list([q.copy(), q[:3000].copy()])
    gpu.load_targets(tres, toff, 21)
    out = gpu.sw_batch(mat, GO, GE, [dict(q=q, comp_bias=None, targets=np.array([0, 1], np.uint32))], mode=0)
    for k in range(2):
        r = oracle.sw_align(q, None, tres[int(toff[k]):int(toff[k + 1])], mat, GO, GE)
        assert (int(out[k]["score"]), int(out[k]["q_end"]), int(out[k]["t_end"]), int(out[k]["word"])) == \
               (r["score"], r["q_end"], r["t_end"], r["word"])
    assert out[0]["score"] == 32767


def test_size_independent_properties_at_scale(gpu, matrices):
    """Config-2-shaped slice (64 queries x 20k targets = 1.28 M pairs): properties that need no oracle.
    (1) a permuted hit list gives the permuted results (scheduling independence);
    (2) score(q, t) for planted self hits equals the diagonal sum bound; (3) 0 <= ends < lengths."""
    (qres, qoff), (tres, toff) = wl.config2_align_only(n_queries=64, n_targets=20000, seed=5)
    mat = matrices["blosum62_sw"]
    gpu.load_targets(tres, toff, 21)
    qs = wl.split(qres, qoff)
    rng = np.random.default_rng(1)
    ids = np.arange(20000, dtype=np.uint32)
    perm = rng.permutation(20000).astype(np.uint32)
    a = gpu.sw_batch(mat, GO, GE, [dict(q=q, comp_bias=None, targets=ids) for q in qs], mode=0)
    b = gpu.sw_batch(mat, GO, GE, [dict(q=q, comp_bias=None, targets=perm) for q in qs], mode=0)
    a = a.reshape(64, 20000)
    b = b.reshape(64, 20000)
    for f in ("score", "q_end", "t_end", "word"):
        assert np.array_equal(a[f][:, perm], b[f]), f
    tlen = (toff[1:] - toff[:-1]).astype(np.int64)
    qlen = (qoff[1:] - qoff[:-1]).astype(np.int64)
    assert np.all(a["score"] > 0)
    assert np.all(a["t_end"] < tlen[None, :]) and np.all(a["t_end"] >= 0)
    assert np.all(a["q_end"] < qlen[:, None]) and np.all(a["q_end"] >= 0)
    # checksum of checksums, recorded for the log
    print("checksum", int(a["score"].astype(np.int64).sum()), int(a["t_end"].astype(np.int64).sum()))


def test_traceback_golden_vectors(gpu, matrices, sw_vectors):
    """Backtrace strings and identity counts of every golden pair against the REAL reference's (banded_sw +
    computerBacktrace, recorded in tests/golden/sw_vectors.npz)."""
    v = sw_vectors
    mat = matrices["blosum62_sw"]
    gpu.load_targets(v["tres"], v["toff"], 21)
    qoff = v["qoff"].astype(np.int64)
    queries = [dict(q=v["qres"][qoff[i]:qoff[i + 1]], comp_bias=v["cb"][qoff[i]:qoff[i + 1]],
                    targets=np.array([i], np.uint32), min_start_score=0) for i in range(len(qoff) - 1)]
    b = gpu.sw_prepare(mat, GO, GE, queries, mode=1)
    b.run()
    res = b.fetch()
    info, strs = b.traceback(np.arange(len(queries), dtype=np.uint32))
    n_bt = n_word = 0
    for i in range(len(queries)):
        if res[i]["score"] <= 0:
            assert info[i]["status"] == 3 and strs[i] == ""
            continue
        assert info[i]["status"] == 0, (i, int(info[i]["status"]))
        assert strs[i] == v["bt"][i], (i, strs[i][:60], v["bt"][i][:60])
        assert int(info[i]["ident"]) == int(v["expect"][i][6]), i
        # SURVEY.md section 8c's contract for word == 1 hits (and every other one): the path re-scores to score1
        toff = v["toff"].astype(np.int64)
        s, qe, te = rescore(queries[i]["q"], queries[i]["comp_bias"], v["tres"][toff[i]:toff[i + 1]], mat, GO, GE,
                            int(res[i]["q_start"]), int(res[i]["t_start"]), strs[i])
        assert (s, qe, te) == (int(res[i]["score"]), int(res[i]["q_end"]), int(res[i]["t_end"])), (i, int(res[i]["word"]))
        n_word += int(res[i]["word"])
        n_bt += 1
    assert n_bt > 150 and n_word >= 20
    b.free()


def test_traceback_vs_oracle_lists(gpu, oracle, matrices):
    """A prefilter-like list (homologs of several identities and indel loads, plus unrelated targets), arbitrary
    subset and order of pairs, against the oracle's banded backtrace."""
    rng = np.random.default_rng(77)
    mat = matrices["blosum62_sw"]
    qs, tl = [], []
    for k in range(6):
        qs.append(rng.choice(20, size=int(rng.integers(40, 900)), p=wl.BACKGROUND).astype(np.uint8))
    for k in range(240):
        src = qs[k % 6]
        if k % 5 == 4:
            tl.append(rng.choice(20, size=int(rng.integers(30, 600)), p=wl.BACKGROUND).astype(np.uint8))
        else:
            h = wl.mutate(rng, src, float(rng.uniform(0.25, 0.98)), max_indels=6, max_indel_len=25)
            pre = rng.choice(20, size=int(rng.integers(0, 80)), p=wl.BACKGROUND).astype(np.uint8)
            tl.append(np.concatenate([pre, h, pre[::-1]]))
    tres, toff = wl.seqs_from_list(tl)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(240, dtype=np.uint32)
    queries = [dict(q=q, comp_bias=_round_cb(oracle, matrices, q), targets=ids, min_start_score=0) for q in qs]
    b = gpu.sw_prepare(mat, GO, GE, queries, mode=1)
    b.run()
    res = b.fetch()
    pick = rng.permutation(6 * 240)[:700].astype(np.uint32)
    info, strs = b.traceback(pick)
    n_ok = 0
    for k, p in enumerate(pick.tolist()):
        qi, ti = divmod(p, 240)
        r = oracle.sw_align(qs[qi], queries[qi]["comp_bias"], tl[ti], mat, GO, GE, need_start=True, need_bt=True)
        if r["score"] <= 0:
            assert info[k]["status"] == 3
            continue
        assert info[k]["status"] == 0, (k, p, int(info[k]["status"]))
        assert strs[k] == r["bt"], (k, p, strs[k][:50], r["bt"][:50])
        assert int(info[k]["ident"]) == r["ident"]
        n_ok += 1
    assert n_ok > 500
    b.free()


def test_very_long_query_and_target(gpu, oracle, matrices):
    """a 33 000-residue query (74 tiles) and a 40 000-residue target with planted homologs: scores, ends, starts and the
    backtraces of the planted pairs against the oracle"""
    rng = np.random.default_rng(11)
    mat = matrices["blosum62_sw"]
    small = [rng.choice(20, size=int(rng.integers(150, 400)), p=wl.BACKGROUND).astype(np.uint8) for _ in range(6)]
    big_t = rng.choice(20, size=40000, p=wl.BACKGROUND).astype(np.uint8)
    big_t[1000:1000 + len(small[0])] = small[0]
    h = wl.mutate(rng, small[1], 0.8)
    big_t[34000:34000 + len(h)] = h
    big_q = rng.choice(20, size=33000, p=wl.BACKGROUND).astype(np.uint8)
    big_q[5000:5000 + len(small[3])] = small[3]
    tl = small + [big_t]
    tres, toff = wl.seqs_from_list(tl)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(len(tl), dtype=np.uint32)
    qs = [small[0], small[1], big_q]
    queries = [dict(q=q, comp_bias=_round_cb(oracle, matrices, q), targets=ids, min_start_score=40) for q in qs]
    b = gpu.sw_prepare(mat, GO, GE, queries, mode=1)
    b.run()
    out = b.fetch().reshape(3, len(tl))
    for qi, qd in enumerate(queries):
        for k in range(len(tl)):
            r = oracle.sw_align(qd["q"], qd["comp_bias"], tl[k], mat, GO, GE, need_start=True)
            g = out[qi, k]
            assert (int(g["score"]), int(g["q_end"]), int(g["t_end"]), int(g["word"])) == (r["score"], r["q_end"], r["t_end"], r["word"]), (qi, k)
            if r["score"] >= 40:
                assert (int(g["q_start"]), int(g["t_start"])) == (r["q_start"], r["t_start"]), (qi, k)
    pick = np.array([0 * len(tl) + 6, 1 * len(tl) + 6, 2 * len(tl) + 3, 2 * len(tl) + 6], np.uint32)
    info, strs = b.traceback(pick)
    for k, p in enumerate(pick.tolist()):
        qi, ti = divmod(p, len(tl))
        if out[qi, ti]["score"] < 40:
            continue
        r = oracle.sw_align(queries[qi]["q"], queries[qi]["comp_bias"], tl[ti], mat, GO, GE, need_start=True, need_bt=True)
        assert int(info[k]["status"]) in (0, 1), (k, int(info[k]["status"]))
        if int(info[k]["status"]) == 0:
            assert strs[k] == r["bt"] and int(info[k]["ident"]) == r["ident"], k
    b.free()


def test_block_aligner_on_device_equals_restatement(gpu, matrices, oracle):
    """Row a15: start position, identities and backtrace of int16-range hits from the device's block aligner
    (mmgpu_sw_block_backtrace, block_kernel.hip) against the plain-C restatement of the crate (oracle/block_oracle.c,
    tests/test_block_oracle.py): every pair the device answers must equal it field by field; what it declines it must name."""
    from mmseqs2_amd import workloads as wl
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    (qres, qoff), (tres, toff), fam_t, fam_q = wl.config3_prefilter(160, 6, 160, seed=33)
    qs, ts = wl.split(qres, qoff), wl.split(tres, toff)
    gpu.load_targets(tres, toff, 21)
    queries = []
    for qi, q in enumerate(qs):
        own = np.nonzero(fam_t == fam_q[qi])[0][:4].astype(np.uint32)
        other = np.array([(qi * 7 + 3) % len(ts)], np.uint32)
        cb = oracle.round_comp_bias(oracle.comp_bias(sub16, matrices["blosum62_pback"], q, 1.0))
        queries.append(dict(q=q, comp_bias=cb, targets=np.concatenate([own, other]), min_start_score=0))
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    res = b.fetch()
    blk, strs = b.block_backtrace(np.arange(len(res), dtype=np.uint32))
    b.free()
    k = n_ok = n_word = n_large = 0
    for qd in queries:
        for t_id in qd["targets"]:
            r, o = res[k], blk[k]
            if r["word"] != 1 or r["score"] <= 0:
                assert o["status"] == 3, k
            else:
                n_word += 1
                w = oracle.block_backtrace(qd["q"], qd["comp_bias"], ts[int(t_id)], mat, 11, 1, int(r["score"]), int(r["q_end"]), int(r["t_end"]))
                if o["status"] == 0:
                    assert w["ok"], k
                    assert (int(o["q_start"]), int(o["t_start"]), int(o["ident"]), strs[k]) == (w["q_start"], w["t_start"], w["ident"], w["bt"]), k
                    n_ok += 1
                elif o["status"] == 1:
                    assert not w["ok"], k
                else:
                    n_large += 1
            k += 1
    # round 4: blocks grow to the crate's 4096 rows on the device, so nothing is left undecided
    assert n_word > 250 and n_large == 0 and n_ok >= n_word - max(3, n_word // 20), (n_word, n_ok, n_large)


def _long_gap_pairs(matrices, oracle, n, seed):
    """pairs whose alignment carries one long gap between two strongly similar flanks: the block aligner has to grow its block
    well beyond 512 rows to follow it (the x-drop at small sizes ends in the first flank and the score is not reached)"""
    rng = np.random.default_rng(seed)
    bg = matrices["blosum62_pback"].astype(np.float64)[:20]
    bg = bg / bg.sum()
    qs, ts = [], []
    for k in range(n):
        a = rng.choice(20, size=int(rng.integers(170, 260)), p=bg).astype(np.uint8)
        bfl = rng.choice(20, size=int(rng.integers(170, 260)), p=bg).astype(np.uint8)
        z = rng.choice(20, size=int(rng.integers(300, 1500)), p=bg).astype(np.uint8)
        q = np.concatenate([a, bfl])
        t = np.concatenate([a, z, bfl]) if k % 2 == 0 else np.concatenate([rng.choice(20, size=40, p=bg).astype(np.uint8), a, z, bfl])
        if k % 3 == 2:      # gap on the other side
            q, t = t, q
        # a few substitutions so that ties and mismatches occur inside the flanks
        q = q.copy()
        for p in rng.integers(0, len(q), size=len(q) // 25):
            q[p] = rng.integers(0, 20)
        qs.append(q)
        ts.append(t)
    return qs, ts


def test_block_aligner_growth_sequences_equal_the_restatement_step_by_step(gpu, matrices, oracle):
    """a15, beyond the final CIGAR: the block LIST of every pair - where each block was placed, its height and width, whether it
    was a shift right or down (a grow step adds one block of each kind) - as it stands when the crate's align_core returns, i.e.
    the whole sequence of grow / shift decisions with the block sizes 32 -> ... -> 4096 after every x-drop restore, must equal the
    restatement's list entry by entry.  Long-gap pairs (blocks grow to 512 - 4096 rows) and family pairs (32 / 64-row blocks)."""
    from mmseqs2_amd import workloads as wl
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    qs, ts = _long_gap_pairs(matrices, oracle, 30, seed=17)
    (qres, qoff), (tres2, toff2), fam_t, fam_q = wl.config3_prefilter(40, 4, 40, seed=43)
    fq, ft = wl.split(qres, qoff), wl.split(tres2, toff2)
    for qi, q in enumerate(fq):
        for ti in np.nonzero(fam_t == fam_q[qi])[0][:2]:
            qs.append(q)
            ts.append(ft[ti])
    toff = np.zeros(len(ts) + 1, np.uint64)
    toff[1:] = np.cumsum([len(t) for t in ts])
    gpu.load_targets(np.concatenate(ts), toff, 21)
    queries = []
    for qi, q in enumerate(qs):
        cb = oracle.round_comp_bias(oracle.comp_bias(sub16, matrices["blosum62_pback"], q, 1.0))
        queries.append(dict(q=q, comp_bias=cb, targets=np.array([qi], np.uint32), min_start_score=0))
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    res = b.fetch()
    blk, lists = b.block_growth(np.arange(len(res), dtype=np.uint32))
    b.free()
    n = n_grown = 0
    sizes_seen = set()
    for k, qd in enumerate(queries):
        r, o = res[k], blk[k]
        if r["word"] != 1 or r["score"] <= 0:
            continue
        w, want = oracle.block_growth(lambda: oracle.block_backtrace(qd["q"], qd["comp_bias"], ts[k], mat, 11, 1, int(r["score"]), int(r["q_end"]),
                                                                    int(r["t_end"])))
        assert int(o["status"]) == (0 if w["ok"] else 1), k
        got = lists[k]
        assert got.shape == want.shape and np.array_equal(got, want), \
            (k, got.shape, want.shape, next((i, got[i].tolist(), want[i].tolist()) for i in range(min(len(got), len(want))) if not np.array_equal(got[i], want[i])))
        n += 1
        big = int(max(want[:, 2].max(), want[:, 3].max())) if len(want) else 0
        sizes_seen.add(big)
        n_grown += big > 512
    assert n >= 60 and n_grown >= 5 and len(sizes_seen) >= 4, (n, n_grown, sorted(sizes_seen))


@pytest.mark.parametrize("first_tier", [0, 1, 2])
def test_block_aligner_grows_to_the_crates_4096_rows(gpu, matrices, oracle, first_tier, monkeypatch):
    """a15, blocks beyond 512 rows: the later launches (sw_block_kernel<2048, LDS> and <4096, borders in HBM>) answer what the
    512-row form leaves, equal to the restatement field by field; with MMGPU_BLOCK_FIRST_TIER = 1 / 2 every pair (also the short
    ones of the family workload) starts in the 2048-row / 4096-row instantiation."""
    full_only = first_tier > 0
    from mmseqs2_amd import workloads as wl
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    qs, ts = _long_gap_pairs(matrices, oracle, 24, seed=5)
    if full_only:
        monkeypatch.setenv("MMGPU_BLOCK_FIRST_TIER", str(first_tier))
        (qres, qoff), (tres2, toff2), fam_t, fam_q = wl.config3_prefilter(30, 4, 30, seed=41)
        fq, ft = wl.split(qres, qoff), wl.split(tres2, toff2)
        for qi, q in enumerate(fq):
            for ti in np.nonzero(fam_t == fam_q[qi])[0][:2]:
                qs.append(q)
                ts.append(ft[ti])
    toff = np.zeros(len(ts) + 1, np.uint64)
    toff[1:] = np.cumsum([len(t) for t in ts])
    gpu.load_targets(np.concatenate(ts), toff, 21)
    queries = []
    for qi, q in enumerate(qs):
        cb = oracle.round_comp_bias(oracle.comp_bias(sub16, matrices["blosum62_pback"], q, 1.0))
        queries.append(dict(q=q, comp_bias=cb, targets=np.array([qi], np.uint32), min_start_score=0))
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    res = b.fetch()
    blk, strs = b.block_backtrace(np.arange(len(res), dtype=np.uint32))
    first, second = b.block_tiers()
    b.free()
    n_ok = n_big = 0
    for k, qd in enumerate(queries):
        r, o = res[k], blk[k]
        if r["word"] != 1 or r["score"] <= 0:
            assert o["status"] == 3, k
            continue
        w = oracle.block_backtrace(qd["q"], qd["comp_bias"], ts[k], mat, 11, 1, int(r["score"]), int(r["q_end"]), int(r["t_end"]))
        assert int(o["status"]) == (0 if w["ok"] else 1), (k, int(o["status"]), w["ok"], w["block_size"])
        if w["ok"]:
            assert (int(o["q_start"]), int(o["t_start"]), int(o["ident"]), strs[k]) == (w["q_start"], w["t_start"], w["ident"], w["bt"]), k
            n_ok += 1
            n_big += w["block_size"] > 512 or ("D" * 300 in w["bt"]) or ("I" * 300 in w["bt"])
    assert n_ok >= 20 and n_big >= 12, (n_ok, n_big)
    if full_only:
        assert first == 0 and second >= n_ok
    else:
        assert second >= 2 and first > 0, (first, second)      # (a 512-row block usually follows these gaps; a few pairs grow further)


def _family_queries(matrices, oracle, seed, n_fam=120, min_start=0):
    from mmseqs2_amd import workloads as wl
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    (qres, qoff), (tres, toff), fam_t, fam_q = wl.config3_prefilter(n_fam, 6, n_fam, seed=seed)
    qs, ts = wl.split(qres, qoff), wl.split(tres, toff)
    queries = []
    for qi, q in enumerate(qs):
        own = np.nonzero(fam_t == fam_q[qi])[0][:4].astype(np.uint32)
        other = np.array([(qi * 7 + 3) % len(ts), (qi * 11 + 5) % len(ts)], np.uint32)
        cb = oracle.round_comp_bias(oracle.comp_bias(sub16, matrices["blosum62_pback"], q, 1.0))
        queries.append(dict(q=q, comp_bias=cb, targets=np.concatenate([own, other]), min_start_score=min_start))
    return mat, queries, tres, toff


def test_start_not_word_mode_and_reverse_pairs(gpu, matrices, oracle):
    """MMGPU_SW_START_NOT_WORD (mode 2): the reverse scan runs for the hits of the reference's uint8 pass only - an int16-range hit takes
    its start from the block aligner (StripedSmithWaterman.cpp:865-882) - and mmgpu_sw_reverse_pairs supplies, after the fact, the
    start positions of exactly the pairs it is given (the fall-back of :873-882): together they equal mode 1 record by record."""
    mat, queries, tres, toff = _family_queries(matrices, oracle, seed=51, min_start=60)
    qs_long, ts_long = _long_gap_pairs(matrices, oracle, 6, seed=9)      # multi-tile queries as well
    ts = wl.split(tres, toff) + ts_long
    base = len(ts) - len(ts_long)
    sub16 = mat.astype(np.int16)
    for k, q in enumerate(qs_long):
        cb = oracle.round_comp_bias(oracle.comp_bias(sub16, matrices["blosum62_pback"], q, 1.0))
        queries.append(dict(q=q, comp_bias=cb, targets=np.array([base + k, k], np.uint32), min_start_score=60))
    tres2, toff2 = wl.seqs_from_list(ts)
    gpu.load_targets(tres2, toff2, 21)
    b1 = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b1.run()
    ref = b1.fetch()
    b1.free()
    b2 = gpu.sw_prepare(mat, 11, 1, queries, mode=2)
    b2.run()
    got = b2.fetch()
    word = (ref["word"] == 1) & (ref["score"] >= 60)
    assert word.sum() > 100 and ((ref["word"] == 0) & (ref["q_start"] >= 0)).sum() > 20
    for f in ("score", "q_end", "t_end", "word"):
        assert np.array_equal(got[f], ref[f]), f
    assert np.array_equal(got["q_start"][~word], ref["q_start"][~word]) and np.array_equal(got["t_start"][~word], ref["t_start"][~word])
    assert (got["q_start"][word] == -1).all() and (got["t_start"][word] == -1).all()
    # every other int16-range pair, after the fact
    pick = np.nonzero(word)[0][::2].astype(np.uint32)
    back = b2.reverse_pairs(pick)
    assert np.array_equal(back, ref[pick])
    after = b2.fetch()
    rest = np.nonzero(word)[0][1::2]
    assert np.array_equal(after[pick], ref[pick]) and (after["q_start"][rest] == -1).all()
    untouched = np.ones(len(ref), bool)
    untouched[pick] = False
    assert np.array_equal(after[untouched], got[untouched])
    b2.free()


def test_block_starts_is_the_search_semantics_of_one_call(gpu, matrices, oracle):
    """mmgpu_sw_block_starts on a mode-2 batch: the device selects the int16-range hits that reach min_start_score, the block aligner
    supplies their start positions, the reverse scan those of the pairs it declines - record for record what mode 1 + an explicit
    mmgpu_sw_block_backtrace call + the host's choice between the two give (MMGpuMatcher.cpp), and what the restatement says."""
    mat, queries, tres, toff = _family_queries(matrices, oracle, seed=52, n_fam=150, min_start=70)
    ts = wl.split(tres, toff)
    gpu.load_targets(tres, toff, 21)
    b1 = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b1.run()
    ref = b1.fetch()
    sel = np.nonzero((ref["word"] == 1) & (ref["score"] >= 70) & (ref["score"] > 0))[0].astype(np.uint32)
    blk, _ = b1.block_backtrace(sel, mode="starts")
    b1.free()
    expect = ref.copy()
    ok = blk["status"] == 0
    assert set(np.unique(blk["status"]).tolist()) <= {0, 1}
    expect["q_start"][sel[ok]] = blk["q_start"][ok]
    expect["t_start"][sel[ok]] = blk["t_start"][ok]
    # pairs below the threshold keep -1 in both modes; int16-range pairs the block aligner declined keep the reverse scan's start
    b2 = gpu.sw_prepare(mat, 11, 1, queries, mode=2)
    b2.run()
    n_sel, n_declined, n_large = b2.block_starts()
    got = b2.fetch()
    b2.free()
    assert n_sel == len(sel) and n_declined == int((~ok).sum()) and n_large == 0
    assert np.array_equal(got, expect)
    # against the restatement, a sample
    pair_q = np.repeat(np.arange(len(queries)), [len(q["targets"]) for q in queries])
    pair_t = np.concatenate([q["targets"] for q in queries])
    n_checked = 0
    for p in sel[::7]:
        qd = queries[int(pair_q[p])]
        w = oracle.block_backtrace(qd["q"], qd["comp_bias"], ts[int(pair_t[p])], mat, 11, 1, int(ref[p]["score"]), int(ref[p]["q_end"]), int(ref[p]["t_end"]))
        if w["ok"]:
            assert (int(got[p]["q_start"]), int(got[p]["t_start"])) == (w["q_start"], w["t_start"]), p
        else:
            r = oracle.sw_align(qd["q"], qd["comp_bias"], ts[int(pair_t[p])], mat, 11, 1, need_start=True)
            assert (int(got[p]["q_start"]), int(got[p]["t_start"])) == (r["q_start"], r["t_start"]), p
        n_checked += 1
    assert n_sel > 300 and n_checked > 40
