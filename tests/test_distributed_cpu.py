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
