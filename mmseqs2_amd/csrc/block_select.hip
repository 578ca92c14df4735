// Search semantics on the device (mmgpu_sw_block_starts): which pairs of an alignment batch take their start position from the
// block aligner, and where its answers go.
//
// ssw_align_private (StripedSmithWaterman.cpp:846-890): a pair that passes the E-value gate (:857-863; here: score >= the query's
// min_start_score) and whose score left the uint8 pass (word == 1) asks the block aligner first (:865-882); its start position is
// the block aligner's unless that declines, in which case alignStartPosBacktrace's reverse scan runs (:873-882).
//   block_select_kernel   compacts those pairs into the block aligner's job list (BlockJob) without the records leaving the device
//   block_scatter_kernel  writes the answers into the batch's result records; a declined pair is flagged for the reverse scan
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

__global__ __launch_bounds__(256) void block_select_kernel(BlockSelectArgs A) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    bool take = false;
    mmgpu_sw_hit h;
    uint32_t q = 0;
    if (p < A.pairs) {
        h = A.res[p];
        if (h.score > 0 && h.word == 1 && h.t_end >= 0) {
            // the pair's query: the last q with qout_off[q] <= p
            uint32_t lo = 0, hi = A.n_queries;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (A.qout_off[mid] <= p) lo = mid; else hi = mid;
            }
            q = lo;
            take = h.score >= A.q_minstart[q];
        }
    }
    const unsigned long long bal = __ballot(take);
    if (bal == 0ull) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(A.count, (uint32_t)__popcll(bal));
    base = __builtin_amdgcn_readfirstlane(base);
    if (!take) return;
    const uint32_t k = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (k >= A.cap) return;
    BlockJob j;
    j.query = q;
    j.target = A.slot_target[p];
    j.score = h.score;
    j.q_end = h.q_end;
    j.t_end = h.t_end;
    j.slot = k;
    A.jobs[k] = j;
    A.pair_of_slot[k] = p;
    mmgpu_sw_block o;
    o.q_start = -1; o.t_start = -1; o.ident = 0; o.bt_len = 0; o.bt_off = 0; o.status = MMGPU_BLOCK_TOO_LARGE; o.reserved = 0;
    A.blk[k] = o;
}

__global__ __launch_bounds__(256) void block_scatter_kernel(BlockScatterArgs A) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    int st = -1;
    if (k < A.n) {
        const mmgpu_sw_block o = A.blk[k];
        const uint32_t p = A.pair_of_slot[k];
        st = o.status;
        if (st == MMGPU_BLOCK_OK) {
            A.res[p].q_start = o.q_start;
            A.res[p].t_start = o.t_start;
        } else if (st == MMGPU_BLOCK_DECLINED) {
            A.rev_force[p] = 1;
        }
    }
    for (int kind = 0; kind < 3; kind++) {      // counts[0 / 1 / 2] = OK / DECLINED / TOO_LARGE
        const unsigned long long bal = __ballot(st == kind);
        if (bal != 0ull && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(bal)) atomicAdd(A.counts + kind, (uint32_t)__popcll(bal));
    }
}

}  // namespace

hipError_t launch_block_select(const BlockSelectArgs &A, hipStream_t stream) {
    if (A.pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(block_select_kernel, dim3((A.pairs + 255u) / 256u), dim3(256), 0, stream, A);
    return hipGetLastError();
}

hipError_t launch_block_scatter(const BlockScatterArgs &A, hipStream_t stream) {
    if (A.n == 0) return hipSuccess;
    hipLaunchKernelGGL(block_scatter_kernel, dim3((A.n + 255u) / 256u), dim3(256), 0, stream, A);
    return hipGetLastError();
}

}  // namespace mmgpu
