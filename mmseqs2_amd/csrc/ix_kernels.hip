// k-mer index construction in HBM: IndexBuilder::fillDatabase (src/prefiltering/IndexBuilder.cpp:118-166,226-270) with
// masking off, i.e. IndexTable::addKmerCount / addSequence / sortDBSeqLists (IndexTable.h:135-191,350-403), over the
// target set that mmgpu_load_targets made resident.  Same content, bit for bit, as the reference's index (and as
// mmgpu_host_index_build): per target ONE entry (seqId, first position) for every distinct k-mer whose window holds
// no X and whose self score reaches the k-mer threshold; every list sorted by seqId.
//
//   ix_target_kernel<false/true>  one workgroup per target: k-mer of every window into LDS (targets up to 4096
//                                 windows; longer ones use a global scratch line), first-occurrence test through a hash
//                                 set in LDS (k-mer -> smallest window), count / scatter with one atomic per entry
//   (offsets = exclusive scan of the counts, pf_scan_kernel over 64 K chunks)
//   ix_sort_short_kernel          one thread per k-mer list: lists of up to 16 entries are insertion-sorted into the
//                                 final array, longer ones are queued
//   ix_sort_long_kernel           one workgroup per queued list: rank by counting (ids are unique inside a list)
// Setup work, not on the per-query path; bound by atomics / random 8-byte scatters.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

constexpr uint32_t IX_INVALID = 0xFFFFFFFFu;
constexpr int IX_LDS_WINDOWS = 4096;

// First occurrences through a hash set in LDS (round 5; before: every window scanned all earlier windows, O(L^2) per target -
// 2.4 + 2.7 s for the 141 k pieces of <= 10 kb of a nucleotide database).  Slot = k-mer << 32 | smallest window holding it:
// a 64-bit compare-and-swap claims an empty slot, a 64-bit minimum keeps the first window of a k-mer that is already there.
// Targets with more distinct k-mers than half the table go through it in passes, each pass taking the k-mers of one hash
// class; a pass whose probe sequence ever runs around the whole table (cannot happen below load 1) falls back to the scan.
constexpr int IX_HASH_BITS = 12;
constexpr uint32_t IX_HASH_SLOTS = 1u << IX_HASH_BITS;
constexpr unsigned long long IX_EMPTY = ~0ull;

__device__ __forceinline__ uint32_t ix_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <bool FILL>
__global__ __launch_bounds__(256) void ix_target_kernel(IxArgs A) {
    __shared__ uint32_t km[IX_LDS_WINDOWS];
    __shared__ unsigned long long tab[IX_HASH_SLOTS];
    __shared__ uint32_t s_full;
    const uint32_t t = blockIdx.x;
    const uint32_t len = A.t_len[t];
    const uint8_t *res = A.t_res + (size_t)A.t_off4[t] * 4;
    const uint32_t nwin = len >= (uint32_t)A.pattern_len ? len - (uint32_t)A.pattern_len + 1u : 0u;
    if (nwin == 0) return;
    uint32_t *K = nwin <= (uint32_t)IX_LDS_WINDOWS ? km : A.scratch + (size_t)A.t_off4[t] * 4;
    for (uint32_t i = threadIdx.x; i < nwin; i += 256) {
        uint32_t idx = 0, pw = 1;
        int self = 0;
        bool bad = false;
        for (int p = 0; p < A.k; p++) {
            const uint32_t r = res[i + A.pat[p]];
            bad |= r >= A.kalph;
            idx += r * pw;
            pw *= A.kalph;
            self += A.self_score[r & 31u];
        }
        if (bad || (A.kmer_thr > 0 && self < A.kmer_thr)) idx = IX_INVALID;
        K[i] = idx;
    }
    __threadfence_block();
    const uint32_t n_pass = (nwin + IX_HASH_SLOTS / 2 - 1) / (IX_HASH_SLOTS / 2);
    for (uint32_t pass = 0; pass < n_pass; pass++) {
        __syncthreads();
        for (uint32_t h = threadIdx.x; h < IX_HASH_SLOTS; h += 256) tab[h] = IX_EMPTY;
        if (threadIdx.x == 0) s_full = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nwin; i += 256) {
            const uint32_t k = K[i];
            if (k == IX_INVALID) continue;
            const uint32_t hv = ix_mix(k);
            if (n_pass > 1 && (hv >> IX_HASH_BITS) % n_pass != pass) continue;
            const unsigned long long mine = ((unsigned long long)k << 32) | i;
            uint32_t h = hv & (IX_HASH_SLOTS - 1), probes = 0;
            for (;;) {
                const unsigned long long cur = atomicCAS(&tab[h], IX_EMPTY, mine);      // (an atomic: no torn read of a slot in the making)
                if (cur == IX_EMPTY) break;                       // claimed
                if ((uint32_t)(cur >> 32) == k) {                 // the k-mer is there: keep its first window
                    atomicMin(&tab[h], mine);
                    break;
                }
                h = (h + 1) & (IX_HASH_SLOTS - 1);
                if (++probes >= IX_HASH_SLOTS) { s_full = 1; break; }
            }
        }
        __syncthreads();
        const bool full = s_full != 0;      // workgroup-uniform
        for (uint32_t i = threadIdx.x; i < nwin; i += 256) {
            const uint32_t k = K[i];
            if (k == IX_INVALID) continue;
            const uint32_t hv = ix_mix(k);
            if (n_pass > 1 && (hv >> IX_HASH_BITS) % n_pass != pass) continue;
            bool first = true;
            if (!full) {
                uint32_t h = hv & (IX_HASH_SLOTS - 1);
                while ((uint32_t)(tab[h] >> 32) != k) h = (h + 1) & (IX_HASH_SLOTS - 1);
                first = (uint32_t)tab[h] == i;
            } else {
                for (uint32_t j = 0; j < i; j++)
                    if (K[j] == k) { first = false; break; }
            }
            if (!first) continue;
            if (FILL) {
                const uint32_t slot = atomicAdd(&A.counts[k], 1u);
                A.entries[(size_t)A.offsets[k] + slot] = (uint64_t)t | ((uint64_t)i << 32);
            } else {
                atomicAdd(&A.counts[k], 1u);
            }
        }
    }
}

__global__ __launch_bounds__(256) void ix_sort_short_kernel(IxSortArgs A) {
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= A.table) return;
    const uint32_t o0 = A.offsets[k], n = A.offsets[k + 1] - o0;
    if (n == 0) return;
    if (n > 16) {
        const uint32_t w = atomicAdd(A.n_long, 1u);
        if (w < A.long_cap) A.long_lists[w] = (uint32_t)k;
        return;
    }
    uint64_t v[16];
#pragma unroll
    for (int z = 0; z < 16; z++) v[z] = (uint32_t)z < n ? A.src[(size_t)o0 + z] : ~0ull;
    // ids are unique inside a list and sit in the low 32 bits: order by them (insertion network on 16 registers)
#pragma unroll
    for (int a = 1; a < 16; a++) {
#pragma unroll
        for (int b = a; b > 0; b--) {
            const bool sw = (uint32_t)v[b] < (uint32_t)v[b - 1] && (uint32_t)b < n;
            const uint64_t lo = sw ? v[b] : v[b - 1], hi = sw ? v[b - 1] : v[b];
            v[b - 1] = lo;
            v[b] = hi;
        }
    }
#pragma unroll
    for (int z = 0; z < 16; z++)
        if ((uint32_t)z < n) A.dst[(size_t)o0 + z] = v[z];
}

__global__ __launch_bounds__(256) void ix_sort_long_kernel(IxSortArgs A) {
    __shared__ uint32_t tile[256];
    const uint32_t k = A.long_lists[blockIdx.x];
    const uint32_t o0 = A.offsets[k], n = A.offsets[k + 1] - o0;
    for (uint32_t b0 = 0; b0 < n; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint64_t mine = i < n ? A.src[(size_t)o0 + i] : 0ull;
        const uint32_t id = (uint32_t)mine;
        uint32_t rank = 0;
        for (uint32_t c0 = 0; c0 < n; c0 += 256) {
            __syncthreads();
            tile[threadIdx.x] = c0 + threadIdx.x < n ? (uint32_t)A.src[(size_t)o0 + c0 + threadIdx.x] : 0xFFFFFFFFu;
            __syncthreads();
            const uint32_t m = min(256u, n - c0);
            for (uint32_t z = 0; z < m; z++) rank += tile[z] < id ? 1u : 0u;
        }
        if (i < n) A.dst[(size_t)o0 + rank] = mine;
    }
}

}  // namespace

hipError_t launch_ix_target(const IxArgs &A, bool fill, hipStream_t s) {
    if (A.n_targets == 0) return hipSuccess;
    if (fill) hipLaunchKernelGGL(ix_target_kernel<true>, dim3(A.n_targets), dim3(256), 0, s, A);
    else hipLaunchKernelGGL(ix_target_kernel<false>, dim3(A.n_targets), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_ix_sort_short(const IxSortArgs &A, hipStream_t s) {
    hipLaunchKernelGGL(ix_sort_short_kernel, dim3((unsigned)((A.table + 255) / 256)), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_ix_sort_long(const IxSortArgs &A, uint32_t n_long, hipStream_t s) {
    if (n_long == 0) return hipSuccess;
    hipLaunchKernelGGL(ix_sort_long_kernel, dim3(n_long), dim3(256), 0, s, A);
    return hipGetLastError();
}

// first use of any kernel of this file loads its code object (tens of milliseconds): mmgpu_warmup does it ahead of time
void warm_ix() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&ix_target_kernel<true>));
}

}  // namespace mmgpu
