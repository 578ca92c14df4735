"""Multi-GPU plumbing of the prefilter: one process per GPU, the target DB sharded across ranks (shard-local ids,
the reference's dbFrom convention, Prefiltering.cpp:879-881), ONE exchange step per query batch - an all-gather of the
per-query hit lists (RCCL over xGMI when the backend is "nccl") - followed by the device analogue of
Prefiltering::mergeTargetSplits (mmgpu_pf_merge_splits).  torch is used for device memory and torch.distributed only.

The same code runs on CPU tensors with the gloo backend (tests): there the merge uses the host mirror
capi.merge_hit_lists_host, which is test plumbing, never a fallback of the GPU path."""
import numpy as np
import torch
import torch.distributed as dist

from . import capi


def allgather_hit_lists(hits_t, counts_t, group=None):
    """hits_t: int32 [nq, stride, 3] (mmgpu_pf_hit viewed as 3 int32), counts_t: int32 [nq]; same shapes on every
    rank.  Returns ([world, nq, stride, 3], [world, nq]) on the tensors' device."""
    world = dist.get_world_size(group)
    out_h = torch.empty((world,) + tuple(hits_t.shape), dtype=hits_t.dtype, device=hits_t.device)
    out_c = torch.empty((world,) + tuple(counts_t.shape), dtype=counts_t.dtype, device=counts_t.device)
    if hits_t.is_cuda and dist.get_backend(group) != "gloo":
        dist.all_gather_into_tensor(out_h, hits_t.contiguous(), group=group)
        dist.all_gather_into_tensor(out_c, counts_t.contiguous(), group=group)
    elif hits_t.is_cuda:
        # debugging aid only (several ranks on one GPU, MMGPU_BENCH_BACKEND=gloo): stage through the host
        gh, gc = allgather_hit_lists(hits_t.cpu(), counts_t.cpu(), group)
        return gh.to(hits_t.device), gc.to(hits_t.device)
    else:
        hl = [torch.empty_like(hits_t) for _ in range(world)]
        cl = [torch.empty_like(counts_t) for _ in range(world)]
        dist.all_gather(hl, hits_t.contiguous(), group=group)
        dist.all_gather(cl, counts_t.contiguous(), group=group)
        out_h = torch.stack(hl)
        out_c = torch.stack(cl)
    return out_h, out_c


def shard_id_offsets(shard_sizes):
    off = np.zeros(len(shard_sizes), np.uint64)
    off[1:] = np.cumsum(np.asarray(shard_sizes, np.uint64))[:-1]
    if int(off[-1]) + int(shard_sizes[-1]) > 0xFFFFFFFF:
        raise ValueError("more than 2^32 targets in total")
    return off.astype(np.uint32)


def gather_and_merge_device(gpu, pf_batch, nq, stride, shard_sizes, group=None):
    """GPU path: batch results -> torch tensors (D2D) -> all-gather (RCCL) -> merge kernel.
    Returns (merged int32 [nq, world*stride, 3], counts int32 [nq]) on the GPU."""
    dev = torch.device("cuda", torch.cuda.current_device())
    hits_t = torch.zeros((nq, stride, 3), dtype=torch.int32, device=dev)
    counts_t = torch.zeros((nq,), dtype=torch.int32, device=dev)
    pf_batch.fetch_device(hits_t.data_ptr(), stride, counts_t.data_ptr())
    gh, gc = allgather_hit_lists(hits_t, counts_t, group)
    world = gh.shape[0]
    out_h = torch.zeros((nq, world * stride, 3), dtype=torch.int32, device=dev)
    out_c = torch.zeros((nq,), dtype=torch.int32, device=dev)
    gpu.pf_merge_splits(gh.data_ptr(), gc.data_ptr(), world, nq, stride, shard_id_offsets(shard_sizes),
                        out_h.data_ptr(), out_c.data_ptr())
    return out_h, out_c


def gather_and_merge_host(hits, counts, shard_sizes, group=None):
    """CPU/gloo path used by the tests: hits PF_HIT_DTYPE [nq, stride], counts uint32 [nq] (numpy)."""
    nq, stride = hits.shape
    h32 = torch.from_numpy(np.ascontiguousarray(hits).view(np.int32).reshape(nq, stride, 3))
    c32 = torch.from_numpy(np.ascontiguousarray(counts).astype(np.int32))
    gh, gc = allgather_hit_lists(h32, c32, group)
    gh = gh.numpy().reshape(gh.shape[0], nq, stride * 3).view(capi.PF_HIT_DTYPE).reshape(gh.shape[0], nq, stride)
    gc = gc.numpy()
    off = shard_id_offsets(shard_sizes)
    return [capi.merge_hit_lists_host([gh[s, q, :gc[s, q]] for s in range(gh.shape[0])], off) for q in range(nq)]


def exchange_sw_results(local_res, slot_index, n_slots, device=None, group=None):
    """Alignment results of the merged lists: every rank aligned the pairs whose target it owns (local_res, structured
    capi.SW_HIT_DTYPE, one per owned pair) and knows the slot of each pair in the flattened [nq * n_splits * stride]
    merged-list array (slot_index).  Slots are owned by exactly one rank, so ONE all-reduce (sum) of the int32 view
    gives every rank the complete result array.  Returns a numpy array of n_slots SW_HIT_DTYPE records (zeros in
    slots no rank owns, i.e. padding of the merged lists)."""
    full = np.zeros(n_slots, capi.SW_HIT_DTYPE)
    full[np.asarray(slot_index, np.int64)] = local_res
    t = torch.from_numpy(full.view(np.int32).reshape(n_slots, 6))
    if device is not None and dist.get_backend(group) != "gloo":
        t = t.to(device)
    dist.all_reduce(t, group=group)
    return t.cpu().numpy().reshape(-1).view(capi.SW_HIT_DTYPE)


def exchange_sw_results_tensor(local_res, slot_index, n_slots, group=None):
    """exchange_sw_results on tensors, nothing staged through the host: local_res is an int32 tensor [n_pairs, 6] (the
    24-byte mmgpu_sw_hit records of the pairs this rank aligned, e.g. filled by SwBatch.fetch_device), slot_index an
    int64 tensor on the same device.  Returns the int32 tensor [n_slots, 6] every rank ends up with (one all-reduce)."""
    full = torch.zeros((n_slots, 6), dtype=torch.int32, device=local_res.device)
    if slot_index.numel():
        full.index_copy_(0, slot_index, local_res)
    dist.all_reduce(full, group=group)
    return full


# ---------------------------------------------------------------------------------------------------------------------
# Sharded search whose merged result equals the UNSPLIT run (include/mmgpu.h, "multi-GPU runs whose merged result equals
# the unsplit run"): the target database is dealt to the ranks by length bucket (capi.partition_targets), every rank
# searches its shard, ONE all-gather of 16-byte exchange records per query batch, the merge kernel redoes the reference's
# threshold / truncation / final sort over the union, every rank aligns the pairs whose target it holds and the
# alignment records of the owned pairs are all-gathered (variable counts, padded to the largest rank).


def setup_shard(gpu, rank, world, tres, toff):
    """Partition the database, make this rank's shard resident and describe it to the library.
    -> dict(shard_of, local_id, global_ids, sizes, residues, n_global)"""
    shard_of, local_id, sizes, residues = capi.partition_targets(toff, world, lib=gpu.L)
    sres, soff, gids = capi.shard_sequences(tres, toff, shard_of, rank)
    gpu.load_targets(sres, soff, 21)
    gpu.pf_set_shard(world, rank, len(toff) - 1, gids, shard_of, local_id)
    return dict(shard_of=shard_of, local_id=local_id, global_ids=gids, sizes=sizes, residues=residues, n_global=len(toff) - 1)


def _all_gather_rows(t, group=None):
    """[...] tensor of identical shape on every rank -> [world, ...]"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not dist.is_initialized():
        return t.unsqueeze(0)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if t.is_cuda and dist.get_backend(group) == "gloo":       # debugging aid: several ranks on one GPU
        return _all_gather_rows(t.cpu(), group).to(t.device)
    if t.is_cuda:
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    else:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        out = torch.stack(parts)
    return out


def exchange_and_merge_device(gpu, pf_batch, nq, stride, identity_global=None, group=None):
    """exchange batch that has been run -> (merged hits int32 [nq, stride, 3] with GLOBAL ids, counts int32 [nq],
    flags int32 [nq]) on the GPU, identical on every rank"""
    dev = torch.device("cuda", torch.cuda.current_device())
    xh = torch.zeros((nq, stride, 4), dtype=torch.int32, device=dev)
    xc = torch.zeros((nq,), dtype=torch.int32, device=dev)
    pf_batch.fetch_exchange(xh.data_ptr(), stride, xc.data_ptr())
    gh, gc = _all_gather_rows(xh, group), _all_gather_rows(xc, group)
    out_h = torch.zeros((nq, stride, 3), dtype=torch.int32, device=dev)
    out_c = torch.zeros((nq,), dtype=torch.int32, device=dev)
    out_f = torch.zeros((nq,), dtype=torch.int32, device=dev)
    pf_batch.merge_exchange(gh.data_ptr(), gc.data_ptr(), gh.shape[0], stride, identity_global, out_h.data_ptr(), stride,
                            out_c.data_ptr(), out_f.data_ptr())
    return out_h, out_c, out_f


def rerun_flagged_unsplit(gpu_full, queries, kmer_thr, max_hits, min_diag_score, ref_bins, out_h, out_c, out_f):
    """exchange_and_merge_device's lists with the flagged ones (bit 0 of out_f) replaced by the unsplit run's: the flagged queries run
    once more on `gpu_full`, a context that holds the whole database (what mmgpu_pf_exchange_redo_unsplit does on a batch's own merged
    lists when the library runs the exchange).  -> (queries re-run, of those left to the host)"""
    flags = out_f.cpu().numpy()
    flagged = np.nonzero(flags & 1)[0]
    if len(flagged) == 0:
        return 0, 0
    b = gpu_full.pf_prepare([queries[int(q)] for q in flagged], kmer_thr, max_hits=max_hits, min_diag_score=min_diag_score, ref_bins=ref_bins)
    b.run()
    hits, counts, status, _ = b.fetch()
    b.free()
    left = 0
    stride = out_h.shape[1]
    for k, q in enumerate(flagged.tolist()):
        if status[k] != 0:
            left += 1
            continue
        row = np.zeros((stride, 3), np.int32)
        n = int(counts[k])
        row[:n] = np.ascontiguousarray(hits[k][:n]).view(np.int32).reshape(n, 3)
        out_h[q].copy_(torch.from_numpy(row))
        out_c[q] = n
        out_f[q] = 0
    return len(flagged), left


def exchange_and_merge_host(xhits, counts, max_hits, min_diag_score, ref_bins, self_scores, identity_global=None, group=None):
    """CPU / gloo mirror used by the tests: xhits PF_XHIT_DTYPE [nq, stride], counts [nq] of THIS rank -> list of merged
    PF_HIT_DTYPE lists (capi.merge_exchange_host per query)"""
    nq, stride = xhits.shape
    h32 = torch.from_numpy(np.ascontiguousarray(xhits).view(np.int32).reshape(nq, stride, 4))
    c32 = torch.from_numpy(np.ascontiguousarray(counts).astype(np.int32))
    gh, gc = _all_gather_rows(h32, group), _all_gather_rows(c32, group)
    gh = gh.numpy().reshape(gh.shape[0], nq, stride * 4).view(capi.PF_XHIT_DTYPE).reshape(gh.shape[0], nq, stride)
    gc = gc.numpy().astype(np.int64) & 0x7FFFFFFF      # bit 31 of an exchanged count is the shard's "whole database" flag
    out = []
    for q in range(nq):
        rec = np.concatenate([gh[s, q, :gc[s, q]] for s in range(gh.shape[0])])
        out.append(capi.merge_exchange_host(rec, max_hits, min_diag_score, ref_bins, self_scores[q],
                                            None if identity_global is None else identity_global[q]))
    return out


def align_owned_pairs(gpu, mat, gap_open, gap_extend, marshalled, merged_hits, merged_counts, nq, stride, mode=1):
    """merged lists (global ids, on the GPU) -> alignment batch of the pairs whose target this rank holds.
    -> (SwBatch, local_counts int32 [nq], local_slot int32 [nq, stride]); the batch's result slot q*stride+k belongs
    to position local_slot[q, k] of query q's merged list"""
    dev = merged_hits.device
    lh = torch.zeros((nq, stride, 3), dtype=torch.int32, device=dev)
    lc = torch.zeros((nq,), dtype=torch.int32, device=dev)
    ls = torch.zeros((nq, stride), dtype=torch.int32, device=dev)
    gpu.pf_localize_lists(merged_hits.data_ptr(), merged_counts.data_ptr(), nq, stride, lh.data_ptr(), lc.data_ptr(), ls.data_ptr())
    b = gpu.sw_prepare_from_lists(mat, gap_open, gap_extend, None, lh.data_ptr(), lc.data_ptr(), stride, mode=mode, marshalled=marshalled)
    return b, lc, ls


def gather_owned_results(local_res, local_counts, local_slot, nq, stride, group=None):
    """local_res int32 [nq, stride, 6] (mmgpu_sw_hit records of this rank's batch), local_counts [nq], local_slot
    [nq, stride] -> int32 [nq, stride, 6]: the records of every rank's owned pairs in merged-list order, on every rank.
    Every rank contributes only its owned pairs: an all-gather of (record, slot) rows padded to the largest rank, instead
    of an all-reduce of the dense array."""
    dev = local_res.device
    mask = torch.arange(stride, device=dev, dtype=torch.int32)[None, :] < local_counts[:, None]
    rows = local_res[mask]                                                     # [P, 6]
    slots = (torch.arange(nq, device=dev, dtype=torch.int64)[:, None] * stride + local_slot.to(torch.int64))[mask]
    packed = torch.cat([rows, slots.to(torch.int32)[:, None]], dim=1)          # [P, 7]
    n_mine = torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev)
    n_all = _all_gather_rows(n_mine, group).reshape(-1)
    cap = int(n_all.max().item())
    pad = torch.zeros((cap, 7), dtype=torch.int32, device=dev)
    pad[:packed.shape[0]] = packed
    allp = _all_gather_rows(pad, group)                                        # [world, cap, 7]
    full = torch.zeros((nq * stride, 6), dtype=torch.int32, device=dev)
    for r in range(allp.shape[0]):
        n = int(n_all[r].item())
        if n:
            full.index_copy_(0, allp[r, :n, 6].to(torch.int64), allp[r, :n, :6])
    return full.reshape(nq, stride, 6)
