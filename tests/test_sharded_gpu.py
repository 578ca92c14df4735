"""GPU tests of the multi-GPU path whose merged result must equal the UNSPLIT run: the N shards
are run one after the other on the one GPU of the box (same context, same code a rank runs), the exchange records are
stacked as the all-gather would deliver them, and the merged lists / alignment results are compared with

  * the unsplit run of the same device path (bit for bit: ids, scores, diagonals, order), which tests/test_prefilter_gpu.py
    pins against the oracle and the reference-recorded vectors, and
  * the unsplit ORACLE directly for a sample of the queries.
"""
import numpy as np
import pytest
import torch

from mmseqs2_amd import capi
from mmseqs2_amd import distributed as D
from mmseqs2_amd import workloads as wl
from tests import pf_common as pc

pytestmark = pytest.mark.gpu


def _tables(gpu, g):
    km16, um8 = g["vtml80_kmer16"], g["blosum62_ungapped"]
    s3, i3 = capi.host_score_matrix(km16, 3, lib=gpu.L)
    return km16, um8, s3, i3


def _unsplit(gpu, g, tres, toff, queries, thr, max_hits, ref_bins):
    km16, um8, s3, i3 = _tables(gpu, g)
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    b = gpu.pf_prepare(queries, thr, max_hits=max_hits, ref_bins=ref_bins)
    b.run()
    out = b.fetch()
    b.free()
    return out


def _sharded(gpu, g, tres, toff, queries, thr, max_hits, ref_bins, world, identity_global=None):
    """-> merged hits (PF_HIT_DTYPE [nq, stride]), counts, flags, per-shard state for the alignment step"""
    km16, um8, s3, i3 = _tables(gpu, g)
    nq = len(queries)
    dev = torch.device("cuda", 0)
    stride = min(max_hits, len(toff) - 1)
    xh_all = torch.zeros((world, nq, stride, 4), dtype=torch.int32, device=dev)
    xc_all = torch.zeros((world, nq), dtype=torch.int32, device=dev)
    last = None
    for r in range(world):
        info = D.setup_shard(gpu, r, world, tres, toff)
        gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        qs = []
        for qi, qd in enumerate(queries):
            ident = None
            if identity_global is not None and identity_global[qi] != 0xFFFFFFFF and info["shard_of"][identity_global[qi]] == r:
                ident = int(info["local_id"][identity_global[qi]])
            qs.append(dict(q=qd["q"], comp_bias=qd["comp_bias"], identity_id=ident))
        b = gpu.pf_prepare(qs, thr, max_hits=max_hits, ref_bins=ref_bins)
        b.run()
        b.fetch_exchange(xh_all[r].data_ptr(), stride, xc_all[r].data_ptr())
        gpu.synchronize()
        if last is not None:
            last.free()
        last = b
    out_h = torch.zeros((nq, stride, 3), dtype=torch.int32, device=dev)
    out_c = torch.zeros((nq,), dtype=torch.int32, device=dev)
    out_f = torch.zeros((nq,), dtype=torch.int32, device=dev)
    last.merge_exchange(xh_all.data_ptr(), xc_all.data_ptr(), world, stride, identity_global, out_h.data_ptr(), stride,
                        out_c.data_ptr(), out_f.data_ptr())
    hits = out_h.cpu().numpy().reshape(nq, stride * 3).view(capi.PF_HIT_DTYPE).reshape(nq, stride)
    # the host mirror of the merge kernel must agree with it (it is what the gloo tests use)
    xh = xh_all.cpu().numpy().reshape(world, nq, stride * 4).view(capi.PF_XHIT_DTYPE).reshape(world, nq, stride)
    xc = xc_all.cpu().numpy().astype(np.int64) & 0x7FFFFFFF      # bit 31: the shard's "depends on the whole database" flag
    last.free()
    return hits, out_c.cpu().numpy(), out_f.cpu().numpy(), (xh, xc)


def _self_score(g, qd):
    """rescoreHits' self score as mmgpu_pf_prepare computes it (UngappedAlignment.cpp:396-400, QueryMatcher.cpp:566)"""
    um8 = g["blosum62_ungapped"].astype(np.int64)
    q = qd["q"].astype(np.int64)
    cb = qd["comp_bias"].astype(np.float32)
    v = np.where(cb < 0.0, cb / np.float32(4) - np.float32(0.5), cb / np.float32(4) + np.float32(0.5))
    corr = np.trunc(v).astype(np.int8).astype(np.int64)
    cur = (um8[q, q] + corr).astype(np.int8).astype(np.int64)
    sc = mx = 0
    for c in cur.tolist():
        sc = max(sc + c, 0)
        mx = max(mx, sc)
    return mx


def _case(seed, n_fam, members, nq):
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=n_fam, members=members, n_queries=nq, seed=seed)
    return wl.split(qres, qoff), tres, toff


@pytest.mark.parametrize("world,max_hits,ref_bins,stage_gb", [(2, 300, 2, None), (3, 40, 4, "1e-7"), (8, 25, 2, None), (4, 300, 0, "2e-4")])
def test_merged_shards_equal_unsplit_run(gpu, monkeypatch, world, max_hits, ref_bins, stage_gb):
    """lists cut at ties (max_hits far below the family size), uneven shard counts, the host's own bin count; two of the
    cases run the shards' stages 2-3 in chunks of queries (one query per chunk / a few), the unsplit run in one chunk"""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(31 + world, 150, 60, 48)
    km16 = g["vtml80_kmer16"]
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    monkeypatch.delenv("MMGPU_PF_STAGE_GB", raising=False)
    hits_u, counts_u, status_u, _ = _unsplit(gpu, g, tres, toff, queries, thr, max_hits, ref_bins)
    if stage_gb:
        monkeypatch.setenv("MMGPU_PF_STAGE_GB", stage_gb)
    hits_s, counts_s, flags, (xh, xc) = _sharded(gpu, g, tres, toff, queries, thr, max_hits, ref_bins, world)
    monkeypatch.delenv("MMGPU_PF_STAGE_GB", raising=False)
    assert np.all(status_u == 0) and np.all(flags == 0)
    truncated = 0
    for qi in range(len(qs)):
        n = int(counts_u[qi])
        assert int(counts_s[qi]) == n, (qi, int(counts_s[qi]), n)
        for f in ("id", "score", "diagonal"):
            assert np.array_equal(hits_s[qi][f][:n], hits_u[qi][f][:n]), (qi, f)
        truncated += n == min(max_hits, len(toff) - 1)
    assert truncated > len(qs) // 2 or max_hits >= 300      # the tie order at the cut was actually exercised
    # host mirror of the merge kernel (what the gloo tests run) on the very same records
    if ref_bins:
        for qi in range(len(qs)):
            rec = np.concatenate([xh[s, qi, :xc[s, qi]] for s in range(world)])
            m = capi.merge_exchange_host(rec, min(max_hits, len(toff) - 1), 15, ref_bins, _self_score(g, queries[qi]))
            n = int(counts_s[qi])
            assert len(m) == n and all(np.array_equal(m[f], hits_s[qi][f][:n]) for f in ("id", "score", "diagonal")), qi
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    rb = ref_bins if ref_bins else None
    for qi in (0, 7, 21):
        if rb is None:
            break
        o = orc.match(queries[qi]["q"], queries[qi]["comp_bias"], rb, max_hits=max_hits)
        n = int(counts_s[qi])
        assert n == len(o["id"]) and np.array_equal(hits_s[qi]["id"][:n], o["id"]) and \
            np.array_equal(hits_s[qi]["score"][:n], o["score"]) and np.array_equal(hits_s[qi]["diagonal"][:n], o["diagonal"]), qi


def test_merged_shards_with_self_hits_and_saturated_scores(gpu):
    """queries that are database members (self hit first, score 65535) with many near-identical family members: the
    truncated-threshold path (more than max_hits saturated elements) and the exact rescoring"""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    rng = np.random.default_rng(5)
    seeds = [rng.choice(20, size=int(rng.integers(150, 500)), p=wl.BACKGROUND).astype(np.uint8) for _ in range(6)]
    tl = []
    for s in seeds:
        for _ in range(80):
            tl.append(wl.mutate(rng, s, float(rng.uniform(0.85, 0.99))))
    for _ in range(400):
        tl.append(rng.choice(20, size=int(rng.integers(50, 600)), p=wl.BACKGROUND).astype(np.uint8))
    perm = rng.permutation(len(tl))
    tl = [tl[i] for i in perm]
    tres, toff = wl.seqs_from_list(tl)
    km16 = g["vtml80_kmer16"]
    qids = np.array([int(np.nonzero(perm == k * 80 + 3)[0][0]) for k in range(6)], np.uint32)
    queries = [dict(q=tl[i], comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], tl[i], lib=gpu.L)[0], identity_id=int(i))
               for i in qids]
    for max_hits in (30, 300):
        hits_u, counts_u, status_u, stats_u = _unsplit(gpu, g, tres, toff, queries, thr, max_hits, 2)
        hits_s, counts_s, flags, (xh, xc) = _sharded(gpu, g, tres, toff, queries, thr, max_hits, 2, 4, identity_global=qids)
        if max_hits == 30:
            assert np.any(stats_u["diag_thr"] >> 31), "the truncated-threshold path was not reached"
        for qi in range(len(queries)):
            n = int(counts_u[qi])
            assert int(counts_s[qi]) == n
            assert hits_s[qi]["id"][0] == qids[qi] and hits_s[qi]["score"][0] == 65535
            rec = np.concatenate([xh[s, qi, :xc[s, qi]] for s in range(4)])
            m = capi.merge_exchange_host(rec, max_hits, 15, 2, _self_score(g, queries[qi]), int(qids[qi]))
            assert len(m) == n and all(np.array_equal(m[f], hits_s[qi][f][:n]) for f in ("id", "score", "diagonal")), (max_hits, qi)
            for f in ("id", "score", "diagonal"):
                assert np.array_equal(hits_s[qi][f][:n], hits_u[qi][f][:n]), (max_hits, qi, f)


def test_alignment_of_owned_pairs_equals_unsplit(gpu, matrices):
    """every (query, target) pair is aligned by the shard that holds the target; the gathered records equal those of the
    unsplit alignment of the same merged lists, slot by slot"""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(77, 100, 40, 32)
    km16, um8, s3, i3 = _tables(gpu, g)
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    swq = [dict(q=q, comp_bias=capi.host_comp_bias(sub16, matrices["blosum62_pback"], q, lib=gpu.L)[1], min_start_score=30) for q in qs]
    world, max_hits = 3, 60
    hits_s, counts_s, _, _ = _sharded(gpu, g, tres, toff, queries, thr, max_hits, 2, world)
    nq, stride = hits_s.shape
    dev = torch.device("cuda", 0)
    mh = torch.from_numpy(np.ascontiguousarray(hits_s).view(np.int32).reshape(nq, stride, 3)).to(dev)
    mc = torch.from_numpy(counts_s.astype(np.int32)).to(dev)
    full = torch.zeros((nq * stride, 6), dtype=torch.int32, device=dev)
    owned = 0
    for r in range(world):
        D.setup_shard(gpu, r, world, tres, toff)
        m = gpu.sw_marshal_queries(mat, 11, 1, swq)
        b, lc, ls = D.align_owned_pairs(gpu, mat, 11, 1, m, mh, mc, nq, stride, mode=1)
        b.run()
        res = torch.zeros((nq, stride, 6), dtype=torch.int32, device=dev)
        b.fetch_device(res.data_ptr())
        gpu.synchronize()
        mask = torch.arange(stride, device=dev, dtype=torch.int32)[None, :] < lc[:, None]
        slots = (torch.arange(nq, device=dev, dtype=torch.int64)[:, None] * stride + ls.to(torch.int64))[mask]
        full.index_copy_(0, slots, res[mask])
        owned += int(lc.sum().item())
        b.free()
    assert owned == int(counts_s.sum())
    got = full.cpu().numpy().reshape(-1).view(capi.SW_HIT_DTYPE).reshape(nq, stride)
    # unsplit alignment of the same lists
    gpu.load_targets(tres, toff, 21)
    host_q = [dict(q=x["q"], comp_bias=x["comp_bias"], targets=hits_s[i]["id"][:counts_s[i]].copy(), min_start_score=30)
              for i, x in enumerate(swq)]
    exp = gpu.sw_batch(mat, 11, 1, host_q, mode=1)
    off = 0
    for i in range(nq):
        n = int(counts_s[i])
        for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
            assert np.array_equal(got[i, :n][f], exp[off:off + n][f]), (i, f)
        off += n


# ---------------------------------------------------------------------------------------------------------------------
# the collectives inside the library (mmgpu_init_multi / mmgpu_pf_exchange_merge / mmgpu_sw_gather_owned)
def _sw_queries(g, matrices, qs, lib):
    sub16 = matrices["blosum62_sw"].astype(np.int16)
    return [dict(q=q, comp_bias=capi.host_comp_bias(sub16, matrices["blosum62_pback"], q, lib=lib)[1], min_start_score=1) for q in qs]


@pytest.mark.parametrize("world,max_hits", [(3, 40), (4, 300)])
def test_multi_context_run_equals_unsplit_run(gpu, matrices, world, max_hits):
    """mmgpu_init_multi with `world` contexts on the one GPU of the box (copy transport: RCCL refuses two ranks on one
    device): the database dealt by length bucket, every shard's index built on the device, ONE mmgpu_multi_pf_run (shard
    prefilters -> all-gather -> merge kernel on every context) and ONE mmgpu_multi_sw_from_pf (owned pairs -> gather).
    Merged lists and alignment records must equal the unsplit fused run of the single context, bit for bit."""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(77 + world, 150, 60, 40)
    km16, um8, s3, i3 = _tables(gpu, g)
    ident = np.full(len(qs), 0xFFFFFFFF, np.uint32)
    ident[::5] = (np.arange(len(qs))[::5] * 97) % (len(toff) - 1)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0],
                    identity_id=None if ident[i] == 0xFFFFFFFF else int(ident[i])) for i, q in enumerate(qs)]
    swq = _sw_queries(g, matrices, qs, gpu.L)
    mat = matrices["blosum62_sw"]
    # unsplit: prefilter + fused alignment on the single context
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    b = gpu.pf_prepare(queries, thr, max_hits=max_hits, ref_bins=2)
    b.run()
    fb = gpu.sw_prepare_from_pf(mat, 11, 1, swq, b, mode=1)
    fb.run()
    res_u = fb.fetch().reshape(len(qs), b.max_hits)
    hits_u, counts_u, status_u, _ = b.fetch()
    cells_u = fb.cells
    fb.free()
    b.free()
    assert np.all(status_u == 0)
    m = capi.MMGpuMulti([0] * world)
    try:
        assert m.transport() == "copy"
        m.load_targets(tres, toff, 21)
        m.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        mb = m.pf_prepare(queries, thr, max_hits=max_hits, ref_bins=2)
        for _ in range(2):          # a second run re-uses every buffer of the step
            m.pf_run(mb)
        hits_s, counts_s, status_s = m.pf_fetch(mb, len(qs))
        res_s, cells_s, ms = m.sw_from_pf(mat, 11, 1, swq, mb, len(qs), mode=1)
        m.pf_free(mb)
    finally:
        m.close()
    assert np.all(status_s == 0)
    assert np.array_equal(counts_s, counts_u)
    assert cells_s == cells_u and ms > 0
    for qi in range(len(qs)):
        n = int(counts_u[qi])
        for f in ("id", "score", "diagonal"):
            assert np.array_equal(hits_s[qi][f][:n], hits_u[qi][f][:n]), (qi, f)
        for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
            assert np.array_equal(res_s[qi][f][:n], res_u[qi][f][:n]), (qi, f)
        assert not res_s[qi]["score"][n:].any()


def test_hits_clustered_in_one_shard_are_gathered_in_a_second_round(gpu, matrices):
    """The owned-pairs gather sizes a rank's send buffer at 1.5 x its even share of the slots (+ 4096).  Here every hit of every
    list lies in shard 0 (family members at the even ids of one length bucket, decoys at the odd ones): shard 0 packs more than
    that, every rank sees it in the gathered counters, and the library repeats the gather with buffers that hold every slot -
    the records still equal the unsplit run's (before: MMGPU_ERR_STATE unless the caller had set MMGPU_SW_GATHER_DENSE)."""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    rng = np.random.default_rng(123)
    L, fam, members, nq = 208, 2, 700, 128
    seeds = [rng.choice(20, size=L, p=wl.BACKGROUND).astype(np.uint8) for _ in range(fam)]
    sres, soff = wl.seqs_from_list(seeds)
    mres, moff = wl.mutate_many(rng, sres, soff, np.repeat(np.arange(fam), members), id_lo=0.6, id_hi=0.95, max_indels=0)
    mem = wl.split(mres, moff)
    assert all(len(x) == L for x in mem)
    targets = []
    for x in mem:      # member, decoy, member, decoy, ...: one length bucket, dealt alternately
        targets.append(x)
        targets.append(rng.choice(20, size=L, p=wl.BACKGROUND).astype(np.uint8))
    tres, toff = wl.seqs_from_list(targets)
    qres, qoff = wl.mutate_many(rng, sres, soff, np.arange(nq) % fam, id_lo=0.7, id_hi=0.95, max_indels=0)
    qs = wl.split(qres, qoff)
    shard_of = capi.partition_targets(toff, 2, lib=gpu.L)[0]
    assert np.all(shard_of[0::2] == 0) and np.all(shard_of[1::2] == 1)
    km16, um8, s3, i3 = _tables(gpu, g)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    swq = _sw_queries(g, matrices, qs, gpu.L)
    mat = matrices["blosum62_sw"]
    max_hits = 300
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    b = gpu.pf_prepare(queries, thr, max_hits=max_hits, ref_bins=2)
    b.run()
    fb = gpu.sw_prepare_from_pf(mat, 11, 1, swq, b, mode=1)
    fb.run()
    res_u = fb.fetch().reshape(len(qs), b.max_hits)
    hits_u, counts_u, status_u, _ = b.fetch()
    fb.free()
    b.free()
    assert np.all(status_u == 0)
    owned0 = int(sum(int((shard_of[hits_u[qi]["id"][:int(counts_u[qi])]] == 0).sum()) for qi in range(len(qs))))
    slots = len(qs) * max_hits
    assert owned0 > slots // 2 * 3 // 2 + 4096, (owned0, slots)      # the first round's send buffer of shard 0 is too small
    m = capi.MMGpuMulti([0, 0])
    try:
        m.load_targets(tres, toff, 21)
        m.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        mb = m.pf_prepare(queries, thr, max_hits=max_hits, ref_bins=2)
        m.pf_run(mb)
        hits_s, counts_s, status_s = m.pf_fetch(mb, len(qs))
        res_s, _, _ = m.sw_from_pf(mat, 11, 1, swq, mb, len(qs), mode=1)
        m.pf_free(mb)
    finally:
        m.close()
    assert np.all(status_s == 0) and np.array_equal(counts_s, counts_u)
    for qi in range(len(qs)):
        n = int(counts_u[qi])
        assert np.array_equal(hits_s[qi]["id"][:n], hits_u[qi]["id"][:n]), qi
        for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
            assert np.array_equal(res_s[qi][f][:n], res_u[qi][f][:n]), (qi, f)


def test_two_query_groups_of_two_target_shards_equal_the_unsplit_run(gpu, matrices):
    """Round 4 layout (bench.py --query-groups): G = 2 groups x S = 2 target shards on four contexts of the one GPU.  Each group is
    an mmgpu_multi of two contexts (its own communicator, here the copy transport) that holds the whole database dealt by length
    bucket and runs HALF of the queries; together they must give the unsplit run's lists and alignment records for all queries."""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(123, 150, 60, 40)
    km16, um8, s3, i3 = _tables(gpu, g)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    swq = _sw_queries(g, matrices, qs, gpu.L)
    mat = matrices["blosum62_sw"]
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    b = gpu.pf_prepare(queries, thr, max_hits=100, ref_bins=2)
    b.run()
    fb = gpu.sw_prepare_from_pf(mat, 11, 1, swq, b, mode=1)
    fb.run()
    res_u = fb.fetch().reshape(len(qs), b.max_hits)
    hits_u, counts_u, status_u, _ = b.fetch()
    fb.free()
    b.free()
    assert np.all(status_u == 0)
    half = len(qs) // 2
    groups = [capi.MMGpuMulti([0, 0]), capi.MMGpuMulti([0, 0])]
    try:
        for gi, m in enumerate(groups):
            lo, hi = (0, half) if gi == 0 else (half, len(qs))
            m.load_targets(tres, toff, 21)
            m.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
            mb = m.pf_prepare(queries[lo:hi], thr, max_hits=100, ref_bins=2)
            m.pf_run(mb)
            hits_s, counts_s, status_s = m.pf_fetch(mb, hi - lo)
            res_s, _, _ = m.sw_from_pf(mat, 11, 1, swq[lo:hi], mb, hi - lo, mode=1)
            m.pf_free(mb)
            assert np.all(status_s == 0) and np.array_equal(counts_s, counts_u[lo:hi])
            for k in range(hi - lo):
                n = int(counts_s[k])
                for f in ("id", "score", "diagonal"):
                    assert np.array_equal(hits_s[k][f][:n], hits_u[lo + k][f][:n]), (gi, k, f)
                for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
                    assert np.array_equal(res_s[k][f][:n], res_u[lo + k][f][:n]), (gi, k, f)
    finally:
        for m in groups:
            m.close()


def test_queries_a_shard_declines_on_the_host_are_reported_not_silently_short(gpu):
    """MMGPU_PF_LONG_SEQ is decided by mmgpu_pf_run on the HOST of each shard (a query of 32768 residues or more) -
    such a query contributes no exchange records, and the merged flag word used to stay clear: the sharded fetch then returned an
    empty list with status OK.  The status of the worst shard must reach the caller (who hands the query to the CPU matcher), and
    the other queries of the batch must still equal the unsplit run."""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(91, 120, 40, 30)
    rng = np.random.default_rng(5)
    long_q = rng.integers(0, 20, size=33000).astype(np.uint8)
    long_q[100:100 + len(qs[3])] = qs[3]
    qs = list(qs[:6]) + [long_q] + list(qs[6:12])
    km16, um8, s3, i3 = _tables(gpu, g)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    b = gpu.pf_prepare(queries, thr, max_hits=50, ref_bins=2)
    b.run()
    hits_u, counts_u, status_u, _ = b.fetch()
    b.free()
    # (the unsplit run scores a query of 32768 residues or more on the device since round 6: pf_longq_kernel; a shard still declines it)
    assert status_u[6] == 0 and counts_u[6] > 0 and qs[3] is not None
    m = capi.MMGpuMulti([0] * 3)
    try:
        m.load_targets(tres, toff, 21)
        m.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        mb = m.pf_prepare(queries, thr, max_hits=50, ref_bins=2)
        m.pf_run(mb)
        hits_s, counts_s, status_s = m.pf_fetch(mb, len(qs))
        has_full = m.has_unsplit()
        m.pf_free(mb)
    finally:
        m.close()
    # with the whole database in a context of its own the long query is re-run there like a query flagged inexact, and answered
    if has_full:
        assert status_s[6] == 0
    else:
        assert status_s[6] == capi.PF_LONG_SEQ and counts_s[6] == 0
    for qi in range(len(qs)):
        if qi == 6 and not has_full:
            continue
        assert status_s[qi] == 0 and counts_s[qi] == counts_u[qi], qi
        n = int(counts_u[qi])
        for f in ("id", "score", "diagonal"):
            assert np.array_equal(hits_s[qi][f][:n], hits_u[qi][f][:n]), (qi, f)


def test_library_communicator_single_rank_rccl(gpu, matrices, monkeypatch):
    """The RCCL transport itself, as far as a 1-GPU box can run it: mmgpu_comm_unique_id + mmgpu_comm_init_rank with one
    rank, the exchange step's all-gathers as real ncclAllGather calls on the context's stream (MMGPU_COMM_SELF_RCCL),
    merged lists and gathered alignment records = the unsplit run."""
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(5, 100, 40, 24)
    km16, um8, s3, i3 = _tables(gpu, g)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    swq = _sw_queries(g, matrices, qs, gpu.L)
    mat = matrices["blosum62_sw"]
    hits_u, counts_u, status_u, _ = _unsplit(gpu, g, tres, toff, queries, thr, 300, 2)
    g2 = capi.MMGpu(0)
    try:
        g2.comm_init_rank(g2.comm_unique_id(), 0, 1)
        assert g2.comm_info() == (0, 1, "rccl")
        monkeypatch.setenv("MMGPU_COMM_SELF_RCCL", "1")
        D.setup_shard(g2, 0, 1, tres, toff)
        g2.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        b = g2.pf_prepare(queries, thr, max_hits=300, ref_bins=2)
        b.run()
        dh, dc, df, stride = b.exchange_merge()
        sb = g2.sw_prepare_owned(mat, 11, 1, swq, b, mode=1)
        sb.run()
        sb.gather_owned()
        res, nrec = sb.fetch_owned()
        nq = len(qs)
        hits = np.zeros((nq, stride), capi.PF_HIT_DTYPE)
        counts = np.zeros(nq, np.uint32)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(hits.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dh), hits.nbytes, 2) == 0
        assert hip.hipMemcpy(counts.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dc), counts.nbytes, 2) == 0
        sb.free()
        b.free()
    finally:
        g2.close()
    assert np.array_equal(counts, counts_u) and nrec == int(counts_u.sum())
    res = res.reshape(nq, stride)
    for qi in range(nq):
        n = int(counts_u[qi])
        for f in ("id", "score", "diagonal"):
            assert np.array_equal(hits[qi][f][:n], hits_u[qi][f][:n]), (qi, f)
        assert np.all(res[qi]["score"][:n] > 0)


@pytest.mark.parametrize("world", [2, 3])
def test_flagged_queries_are_rerun_against_the_unsplit_database(gpu, matrices, monkeypatch, world):
    """Prefiltering::mergeTargetSplits (Prefiltering.cpp:412-526) hands no query back, and neither does a sharded run: a query whose
    shard reaches its share of the reference's databaseHits buffer (flagged MMGPU_PF_SHARD_INEXACT - the tie order at the cut would be
    shard dependent) runs once more against the WHOLE database, held in a context of its own on the first device, and comes back OK
    with the unsplit run's list.  MMGPU_PF_MAX_DB_MATCHES makes a 9 000-target database reach that share."""
    monkeypatch.setenv("MMGPU_PF_MAX_DB_MATCHES", "2500")
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(301 + world, 150, 60, 48)
    km16, um8, s3, i3 = _tables(gpu, g)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    swq = _sw_queries(g, matrices, qs, gpu.L)
    mat = matrices["blosum62_sw"]
    gpu.load_targets(tres, toff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
    b = gpu.pf_prepare(queries, thr, max_hits=300, ref_bins=2)
    b.run()
    fb = gpu.sw_prepare_from_pf(mat, 11, 1, swq, b, mode=1)
    fb.run()
    res_u = fb.fetch().reshape(len(qs), b.max_hits)
    hits_u, counts_u, status_u, stats_u = b.fetch()
    fb.free()
    b.free()
    assert np.all(status_u == 0)
    assert (stats_u["db_matches"] >= 2500).sum() >= 3      # queries that took the overflow path in the unsplit run
    m = capi.MMGpuMulti([0] * world)
    try:
        m.load_targets(tres, toff, 21)
        assert m.has_unsplit()
        m.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        mb = m.pf_prepare(queries, thr, max_hits=300, ref_bins=2)
        m.pf_run(mb)
        res_s, _, _ = m.sw_from_pf(mat, 11, 1, swq, mb, len(qs), mode=1)      # (the first reader of the lists: the re-run happens here)
        hits_s, counts_s, status_s = m.pf_fetch(mb, len(qs))
        redone, left = m.pf_redone(mb)
        m.pf_free(mb)
    finally:
        m.close()
    assert redone >= 3 and left == 0, (redone, left)
    assert np.all(status_s == 0) and np.array_equal(counts_s, counts_u)
    for qi in range(len(qs)):
        n = int(counts_u[qi])
        for f in ("id", "score", "diagonal"):
            assert np.array_equal(hits_s[qi][f][:n], hits_u[qi][f][:n]), (qi, f)
        for f in ("score", "q_end", "t_end", "q_start", "t_start", "word"):
            assert np.array_equal(res_s[qi][f][:n], res_u[qi][f][:n]), (qi, f)


def test_one_rank_reruns_its_flagged_queries_in_a_second_context(gpu, matrices, monkeypatch):
    """The same for one process per GPU (mmgpu_pf_exchange_redo_unsplit): the rank's own exchange batch, a second context on the device
    with the whole database; flags cleared, lists equal to the unsplit run's, the other lists untouched."""
    import mmseqs2_amd
    monkeypatch.setenv("MMGPU_PF_MAX_DB_MATCHES", "2500")
    g = pc.golden()
    thr = int(g["kmer_thr"])
    qs, tres, toff = _case(311, 150, 60, 40)
    km16, um8, s3, i3 = _tables(gpu, g)
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, g["vtml80_pback"], q, lib=gpu.L)[0], identity_id=None) for q in qs]
    full = mmseqs2_amd.MMGpu(0)
    try:
        full.load_targets(tres, toff, 21)
        full.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        bu = full.pf_prepare(queries, thr, max_hits=300, ref_bins=2)
        bu.run()
        hits_u, counts_u, status_u, _ = bu.fetch()
        bu.free()
        assert np.all(status_u == 0)
        D.setup_shard(gpu, 0, 1, tres, toff)      # a context without a communicator is its own single rank
        gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, um8)
        b = gpu.pf_prepare(queries, thr, max_hits=300, ref_bins=2)
        b.run()
        dh, dc, df, stride = b.exchange_merge()
        gpu.synchronize()
        nq = len(qs)

        def merged():
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            mh, mc, mf = np.zeros((nq, stride), capi.PF_HIT_DTYPE), np.zeros(nq, np.uint32), np.zeros(nq, np.uint32)
            for dst, src in ((mh, dh), (mc, dc), (mf, df)):
                assert hip.hipMemcpy(dst.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(src), dst.nbytes, 2) == 0
            return mh, mc, mf
        mh0, mc0, mf0 = merged()
        assert (mf0 & 1).sum() >= 3
        redone, left = b.redo_unsplit(full)
        mh1, mc1, mf1 = merged()
        b.free()
    finally:
        full.close()
        gpu.pf_clear_shard()
    assert redone == int((mf0 & 1).sum()) and left == 0 and not mf1.any()
    assert np.array_equal(mc1, counts_u)
    for qi in range(nq):
        n = int(counts_u[qi])
        for f in ("id", "score", "diagonal"):
            assert np.array_equal(mh1[qi][f][:n], hits_u[qi][f][:n]), (qi, f)
        if not mf0[qi] & 1:
            assert np.array_equal(mh1[qi][:n], mh0[qi][:n]), qi
