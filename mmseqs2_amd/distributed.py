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
