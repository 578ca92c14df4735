"""CPU test of the N>1 path (gloo, world_size 2): the target DB is sharded over two ranks, every rank produces its
shard's hit lists (here with the oracle, since there is no GPU), the lists are all-gathered and merged exactly as the
GPU path does; the result must equal the reference's TARGET_DB_SPLIT semantics (per-split maxResListLen of
Prefiltering.cpp:391-394, concatenate + sort of mergeTargetSplits :412-526) computed without any communication."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shards(g, world):
    from tests import pf_common as pc
    return pc.shards(g, world)


def _local_hits(g, shard, max_hits, nq):
    from mmseqs2_amd import capi
    from tests import pf_common as pc
    orc = pc.pf_oracle()
    orc.build_index(shard[0], shard[1], int(g["kmer_thr"]))
    qs = pc.golden_queries(g)[:nq]
    hits = np.zeros((nq, max_hits), capi.PF_HIT_DTYPE)
    counts = np.zeros(nq, np.uint32)
    for qi, qd in enumerate(qs):
        r = orc.match(qd["q"], qd["comp_bias"], 2, max_hits=max_hits)
        n = len(r["id"])
        hits[qi]["id"][:n] = r["id"]
        hits[qi]["score"][:n] = r["score"]
        hits[qi]["diagonal"][:n] = r["diagonal"]
        counts[qi] = n
    return hits, counts


def _worker(rank, world, port, nq, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmseqs2_amd import capi, distributed as D
    from tests import pf_common as pc
    g = pc.golden()
    shards, sizes = _shards(g, world)
    mh = capi.split_max_hits(300, world)
    hits, counts = _local_hits(g, shards[rank], mh, nq)
    merged = D.gather_and_merge_host(hits, counts, sizes)
    # alignment results of the merged lists: each rank fills the slots of the targets it owns, one all-reduce
    off = D.shard_id_offsets(sizes)
    lo, hi = int(off[rank]), int(off[rank]) + sizes[rank]
    stride = world * mh
    slots, vals = [], []
    for qi, m in enumerate(merged):
        for k, gid in enumerate(m["id"].tolist()):
            if lo <= gid < hi:
                slots.append(qi * stride + k)
                vals.append(gid * 7 + qi)                 # stands for the alignment score of (query, target)
    loc = np.zeros(len(slots), capi.SW_HIT_DTYPE)
    loc["score"] = vals
    loc["q_end"] = rank + 1
    full = D.exchange_sw_results(loc, slots, nq * stride)
    # the tensor form the GPU path uses (nothing staged through numpy) must give the same array
    full_t = D.exchange_sw_results_tensor(torch.from_numpy(loc.view(np.int32).reshape(-1, 6).copy()),
                                          torch.tensor(slots, dtype=torch.int64), nq * stride)
    assert np.array_equal(full_t.numpy().reshape(-1).view(capi.SW_HIT_DTYPE), full)
    if rank == 0:
        q.put(([(m["id"].tolist(), m["score"].tolist(), m["diagonal"].tolist()) for m in merged],
               full["score"].tolist(), full["q_end"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_merge_matches_split_semantics():
    from mmseqs2_amd import capi, distributed as D
    from tests import pf_common as pc
    world, nq = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, nq, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, sw_score, sw_owner = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # expectation without communication
    g = pc.golden()
    shards, sizes = _shards(g, world)
    mh = capi.split_max_hits(300, world)
    assert mh == 300 // 2 + int(4 * (150.0 ** 0.5))
    per = [_local_hits(g, shards[r], mh, nq) for r in range(world)]
    off = D.shard_id_offsets(sizes)
    for qi in range(nq):
        exp = capi.merge_hit_lists_host([per[r][0][qi, :per[r][1][qi]] for r in range(world)], off)
        assert got[qi][0] == exp["id"].tolist() and got[qi][1] == exp["score"].tolist() and got[qi][2] == exp["diagonal"].tolist()
        # sortedness and id ranges of the merged list
        sc = np.abs(np.array(got[qi][1]))
        assert np.all(np.diff(sc) <= 0)
        assert all(0 <= i < sum(sizes) for i in got[qi][0])
    # exchanged alignment results: every merged-list slot carries the value its owner computed
    stride = world * mh
    for qi in range(nq):
        for k, gid in enumerate(got[qi][0]):
            assert sw_score[qi * stride + k] == gid * 7 + qi
            assert sw_owner[qi * stride + k] == (1 if gid < sizes[0] else 2)
        assert all(v == 0 for v in sw_score[qi * stride + len(got[qi][0]):(qi + 1) * stride])
    # with one split the merged list is the plain list
    one = capi.merge_hit_lists_host([per[0][0][0, :per[0][1][0]]], [0])
    assert one["id"].tolist() == per[0][0][0]["id"][:per[0][1][0]].tolist()


# ---------------------------------------------------------------------------------------------------------------------
# The path whose merged result equals the UNSPLIT run: length-bucket partition, all-gather of exchange records, merge over
# the union (host mirror of pf_xmerge_kernel), all-gather of the owned alignment records.  The records here are synthetic
# (what produces them needs a GPU: tests/test_sharded_gpu.py); the expectation is computed without communication.
def _query_truth(n_global, seed, qi):
    """all surviving elements of one query over the whole database (the same draw on every rank)"""
    from mmseqs2_amd import capi
    qr = np.random.default_rng(seed * 1000 + qi)
    rec = np.zeros(n_global, capi.PF_XHIT_DTYPE)
    rec["id"] = np.arange(n_global)
    # even queries: few saturated scores; odd queries: more saturated elements than max_hits (truncated-threshold path)
    if qi % 2 == 0:
        rec["score"] = qr.choice([16, 17, 18, 40, 200, 254, 255, 300], size=n_global, p=[.3, .3, .2, .1, .05, .03, .01, .01])
    else:
        rec["score"] = qr.choice([16, 17, 255, 300, 400, 900, 1200], size=n_global)
    rec["diagonal"] = np.arange(n_global) % 7
    rec["order"] = qr.integers(0, 50, size=n_global)
    return rec


def _synthetic_records(nq, n_global, shard_of, rank, seed, max_hits=25):
    """what a shard hands over: its top max_hits by the unsplit run's order (host mirror of pf_select_kernel<true>)"""
    from mmseqs2_amd import capi
    rec = np.zeros((nq, max_hits), capi.PF_XHIT_DTYPE)
    counts = np.zeros(nq, np.uint32)
    for qi in range(nq):
        mine = _query_truth(n_global, seed, qi)[shard_of == rank]
        sel = capi.select_exchange_host(mine, max_hits, 15, 4, 700 + qi)
        rec[qi][:len(sel)] = sel
        counts[qi] = len(sel)
    return rec, counts


def _worker_unsplit(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmseqs2_amd import capi, distributed as D
    lens = np.random.default_rng(1).integers(30, 4000, 500)
    toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    shard_of, local_id, sizes, residues = capi.partition_targets(toff, world)
    nq = 6
    rec, counts = _synthetic_records(nq, 500, shard_of, rank, 9)
    merged = D.exchange_and_merge_host(rec, counts, 25, 15, 4, [700 + i for i in range(nq)])
    # owned alignment records: value = f(query, global id), gathered into merged-list order
    stride = 25
    lres = torch.zeros((nq, stride, 6), dtype=torch.int32)
    lcnt = torch.zeros((nq,), dtype=torch.int32)
    lslot = torch.zeros((nq, stride), dtype=torch.int32)
    for qi, m in enumerate(merged):
        k = 0
        for pos, gid in enumerate(m["id"].tolist()):
            if shard_of[gid] == rank:
                lres[qi, k, 0] = gid * 3 + qi
                lres[qi, k, 1] = rank + 1
                lslot[qi, k] = pos
                k += 1
        lcnt[qi] = k
    full = D.gather_owned_results(lres, lcnt, lslot, nq, stride)
    if rank == 0:
        q.put(([(m["id"].tolist(), m["score"].tolist()) for m in merged], full.numpy().tolist(), sizes.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_unsplit_equal_path_plumbing():
    from mmseqs2_amd import capi
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_unsplit, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, full, sizes = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lens = np.random.default_rng(1).integers(30, 4000, 500)
    toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    shard_of, local_id, sz, residues = capi.partition_targets(toff, world)
    assert sizes == sz.tolist() and abs(int(sz[0]) - int(sz[1])) <= 1
    assert abs(int(residues[0]) - int(residues[1])) < 0.02 * int(residues.sum())
    # local ids ascend with the global ids inside a shard
    for r in range(world):
        g = np.nonzero(shard_of == r)[0]
        assert np.array_equal(local_id[g], np.arange(len(g)))
    # expectation: the selection over the WHOLE database's elements at once (what the unsplit run sees) - the shards'
    # top lists must be sufficient, in the plain and in the truncated-threshold mode
    nq = 6
    modes = set()
    for qi in range(nq):
        truth = _query_truth(500, 9, qi)
        whole = capi.merge_exchange_host(truth, 25, 15, 4, 700 + qi)
        modes.add(int((np.minimum(truth["score"], 255) == 255).sum() >= 25))
        assert got[qi][0] == whole["id"].tolist() and got[qi][1] == whole["score"].tolist(), qi
        assert len(whole) == 25
        for pos, gid in enumerate(got[qi][0]):
            assert full[qi][pos][0] == gid * 3 + qi and full[qi][pos][1] == shard_of[gid] + 1
    assert modes == {0, 1}


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: G query groups x S target shards (bench.py --query-groups; search_headline's layout).  Four ranks = 2 groups of 2
# shards: every group merges ITS queries over ITS communicator (a torch.distributed sub-group here), the groups never talk to each
# other, and what comes out must be the unsplit run's lists for all queries.
def _worker_groups(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmseqs2_amd import capi, distributed as D
    n_groups, shards = 2, 2
    groups = [dist.new_group(list(range(g * shards, (g + 1) * shards))) for g in range(n_groups)]      # every rank creates every group
    gi, gr = rank // shards, rank % shards
    lens = np.random.default_rng(1).integers(30, 4000, 500)
    toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    shard_of, local_id, sizes, residues = capi.partition_targets(toff, shards)
    nq_all = 8
    mine = list(range(gi * nq_all // n_groups, (gi + 1) * nq_all // n_groups))      # this group's slice of the queries
    rec = np.zeros((len(mine), 25), capi.PF_XHIT_DTYPE)
    counts = np.zeros(len(mine), np.uint32)
    for k, qi in enumerate(mine):
        own = _query_truth(500, 9, qi)[shard_of == gr]
        sel = capi.select_exchange_host(own, 25, 15, 4, 700 + qi)
        rec[k][:len(sel)] = sel
        counts[k] = len(sel)
    merged = D.exchange_and_merge_host(rec, counts, 25, 15, 4, [700 + qi for qi in mine], group=groups[gi])
    stride = 25
    lres = torch.zeros((len(mine), stride, 6), dtype=torch.int32)
    lcnt = torch.zeros((len(mine),), dtype=torch.int32)
    lslot = torch.zeros((len(mine), stride), dtype=torch.int32)
    for k, m in enumerate(merged):
        n = 0
        for pos, gid in enumerate(m["id"].tolist()):
            if shard_of[gid] == gr:
                lres[k, n, 0] = gid * 3 + mine[k]
                lres[k, n, 1] = rank + 1
                lslot[k, n] = pos
                n += 1
        lcnt[k] = n
    full = D.gather_owned_results(lres, lcnt, lslot, len(mine), stride, group=groups[gi])
    if gr == 0:      # one report per group
        q.put((gi, mine, [(m["id"].tolist(), m["score"].tolist()) for m in merged], full.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_as_two_query_groups_of_two_target_shards():
    from mmseqs2_amd import capi
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    reports = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lens = np.random.default_rng(1).integers(30, 4000, 500)
    toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    shard_of, _, _, _ = capi.partition_targets(toff, 2)
    seen = []
    for gi, mine, got, full in reports:
        for k, qi in enumerate(mine):
            whole = capi.merge_exchange_host(_query_truth(500, 9, qi), 25, 15, 4, 700 + qi)      # the unsplit run's list
            assert got[k][0] == whole["id"].tolist() and got[k][1] == whole["score"].tolist(), (gi, qi)
            for pos, gid in enumerate(got[k][0]):      # every slot filled by the rank of THIS group that owns the target
                assert full[k][pos][0] == gid * 3 + qi and full[k][pos][1] == gi * 2 + shard_of[gid] + 1
            seen.append(qi)
    assert sorted(seen) == list(range(8))


def test_query_group_chooser_keeps_target_shards_and_follows_the_stage_model():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert [b.choose_query_groups(n) for n in (1, 2, 4, 8)] == [1, 1, 2, 4]
    assert b.choose_query_groups(8, 2) == 2 and b.choose_query_groups(8, 8) == 8
