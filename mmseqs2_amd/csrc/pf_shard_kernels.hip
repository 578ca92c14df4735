// Multi-GPU prefilter: the step after the all-gather.  Every device holds one shard of the target
// database and has selected, per query, its top max_hits elements by the UNSPLIT run's total order (pf_select_kernel<true>).
// pf_xmerge_kernel redoes, over the union of the shards' records, what QueryMatcher::matchQuery does after
// keepMaxScoreElementOnly (QueryMatcher.cpp:161-241): score histogram -> computeScoreThreshold (QueryMatcher.h:211-221)
// -> the truncated-threshold case (rescoreHits, :563-586) -> the first maxHitsPerQuery elements of the score-sorted array
// (getResult, :401-458; the array is ordered by count, then by CacheFriendlyOperations bin, then by arrival) -> exact
// scores of saturated elements -> self hit first -> hit_t::compareHitsByScoreAndId.  Why the union suffices: the order is
// total and the same on every shard, so the global top-N is contained in the union of the shards' top-N; counts above the
// global cut are complete in the union, which is all computeScoreThreshold looks at.
//
// pf_localize_kernel then gives every device the part of the merged lists whose targets it holds (list order kept), as
// input of the alignment (mmgpu_sw_prepare_from_lists).
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

__device__ __forceinline__ uint32_t x_rescaled_count(uint32_t score, float fms) {
    // rescoreHits, QueryMatcher.cpp:578-581: (score - 255) / maxSelfScore * 255 + 0.5, float arithmetic, cut to a byte
    uint32_t ns = score - 255u;
    ns = ns < 65535u ? ns : 65535u;
    const float sc = (float)ns;
    const float r = __fmul_rn(__fdiv_rn(sc, fms), 255.0f);
    const double dd = (double)r + 0.5;
    return (uint32_t)(int)dd & 0xFFu;
}

struct XKey {
    uint32_t a, b, c;   // (255 - count) << 11 | bin ; list ordinal ; global id
};
__device__ __forceinline__ bool xkey_less(const XKey &x, const XKey &y) {
    if (x.a != y.a) return x.a < y.a;
    if (x.b != y.b) return x.b < y.b;
    return x.c < y.c;
}

__global__ __launch_bounds__(256) void pf_xmerge_kernel(PfXMergeArgs A) {
    // phase 1 (selection): per element global id / exact score / list ordinal + a permutation
    // phase 2 (final order): 64-bit keys of the selected elements, laid over the same memory
    __shared__ uint32_t s_gid[PF_XMERGE_CAP];
    __shared__ uint32_t s_score[PF_XMERGE_CAP];
    __shared__ uint32_t s_order[PF_XMERGE_CAP];
    __shared__ uint16_t s_perm[PF_XMERGE_CAP];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sbase[65];
    __shared__ uint32_t sh_thr, sh_trunc, sh_nelig, sh_inexact;
    const uint32_t q = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        uint32_t any_flag = 0;
        for (uint32_t sp = 0; sp < A.n_shards; sp++) {
            sbase[sp] = run;
            const uint32_t c = A.counts[(size_t)sp * A.nq + q];      // bit 31: the shard's "depends on the whole database" flag
            any_flag |= c >> 31;
            run += min(c & 0x7FFFFFFFu, A.stride);
        }
        sbase[A.n_shards] = run;
        sh_nelig = 0;
        sh_inexact = any_flag;
    }
    for (int k = (int)threadIdx.x; k < 256; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t total = sbase[A.n_shards];
    const uint32_t ident = A.q_identity ? A.q_identity[q] : 0xFFFFFFFFu;
    for (uint32_t sp = 0; sp < A.n_shards; sp++) {
        const uint32_t n = sbase[sp + 1] - sbase[sp];
        const mmgpu_pf_xhit *src = A.xhits + ((size_t)sp * A.nq + q) * A.stride;
        for (uint32_t k = threadIdx.x; k < n; k += 256) {
            const mmgpu_pf_xhit x = src[k];
            const uint32_t e = sbase[sp] + k;
            s_gid[e] = x.id;
            s_score[e] = x.score;
            s_order[e] = x.order;
            atomicAdd(&hist[min(255u, x.score)], 1u);
            if (x.flags & MMGPU_PF_X_INEXACT_ORDER) sh_inexact = 1;
        }
    }
    __syncthreads();
    const uint32_t max_hits = A.max_hits;
    if (threadIdx.x == 0) {
        uint32_t found = 0, thr = 0;   // computeScoreThreshold, QueryMatcher.h:211-221
        for (thr = 255; thr > 0; thr--) {
            found += hist[thr];
            if (found >= max_hits) break;
        }
        const uint32_t dthr = max(A.min_diag_score, thr);
        sh_thr = dthr;
        sh_trunc = dthr >= 255u ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t dthr = sh_thr;
    const bool trunc = sh_trunc != 0;
    int ms = A.q_self_score[q] - 255;
    ms = ms > 1 ? ms : 1;
    ms = ms < 65535 ? ms : 65535;
    const float fms = (float)ms;
    const uint32_t refmask = A.ref_bins - 1;
    auto key_of = [&](uint32_t e) -> XKey {
        XKey k;
        const uint32_t sc = s_score[e], gid = s_gid[e];
        const uint32_t cnt = min(255u, sc);
        bool el;
        uint32_t kc;
        if (trunc) {
            el = cnt >= 255u && gid != ident;
            kc = x_rescaled_count(sc, fms);
        } else {
            el = cnt >= dthr && gid != ident;
            kc = cnt;
        }
        k.a = el ? (((255u - kc) << 11) | (gid & refmask)) : 0xFFFFFFFFu;
        k.b = s_order[e];
        k.c = gid;
        return k;
    };
    uint32_t np2 = 1;
    while (np2 < total) np2 <<= 1;
    uint32_t mine = 0;
    for (uint32_t e = threadIdx.x; e < np2; e += 256) {
        s_perm[e] = (uint16_t)e;
        if (e < total) mine += key_of(e).a != 0xFFFFFFFFu ? 1u : 0u;
    }
    if (mine) atomicAdd(&sh_nelig, mine);
    __syncthreads();
    // bitonic sort of the permutation by the unsplit run's array order; slots >= total sort to the end
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < np2 / 2; k += 256) {
                const uint32_t i = 2 * k - (k & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint32_t pi = s_perm[i], pj = s_perm[j];
                XKey ki, kj;
                if (pi < total) ki = key_of(pi); else { ki.a = 0xFFFFFFFFu; ki.b = 0xFFFFFFFFu; ki.c = 0xFFFFFFFFu; }
                if (pj < total) kj = key_of(pj); else { kj.a = 0xFFFFFFFFu; kj.b = 0xFFFFFFFFu; kj.c = 0xFFFFFFFFu; }
                if (xkey_less(kj, ki) == up) {
                    s_perm[i] = (uint16_t)pj;
                    s_perm[j] = (uint16_t)pi;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t has_ident = ident != 0xFFFFFFFFu ? 1u : 0u;
    const uint32_t want = max_hits > has_ident ? max_hits - has_ident : 0u;
    const uint32_t nsel = min(sh_nelig, want);
    // final scores of the selected elements (getResult :430-452) into registers, then laid over the phase-1 arrays
    constexpr int PER = PF_XMERGE_CAP / 256;
    uint64_t rkey[PER];
    uint16_t rdiag[PER];
#pragma unroll
    for (int z = 0; z < PER; z++) {
        const uint32_t k = threadIdx.x + (uint32_t)z * 256u;
        rkey[z] = ~0ull;
        rdiag[z] = 0;
        if (k < nsel) {
            const uint32_t e = s_perm[k];
            const uint32_t sc = s_score[e];
            const uint32_t cnt = min(255u, sc);
            uint32_t pref;
            if (trunc) pref = 255u + (x_rescaled_count(sc, fms) * (uint32_t)ms / 255u);
            else pref = cnt >= 255u ? sc : cnt;
            rkey[z] = ((uint64_t)(0xFFFFFFFFu - pref) << 32) | (uint64_t)s_gid[e];
            // diagonal: re-read from the record (not kept in LDS)
            uint32_t sp = 0;
            while (sp + 1 < A.n_shards && sbase[sp + 1] <= e) sp++;
            rdiag[z] = A.xhits[((size_t)sp * A.nq + q) * A.stride + (e - sbase[sp])].diagonal;
        }
    }
    __syncthreads();
    uint32_t np2b = 1;
    while (np2b < nsel) np2b <<= 1;
#pragma unroll
    for (int z = 0; z < PER; z++) {
        const uint32_t k = threadIdx.x + (uint32_t)z * 256u;
        if (k < np2b) {
            s_gid[k] = (uint32_t)rkey[z];            // low word: id
            s_score[k] = (uint32_t)(rkey[z] >> 32);  // high word: ~prefScore
            s_perm[k] = rdiag[z];
        }
    }
    __syncthreads();
    // bitonic sort by (prefScore desc, id asc)   (hit_t::compareHitsByScoreAndId, QueryMatcher.h:38-49)
    for (uint32_t size = 2; size <= np2b; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < np2b / 2; k += 256) {
                const uint32_t i = 2 * k - (k & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = ((uint64_t)s_score[i] << 32) | s_gid[i], b = ((uint64_t)s_score[j] << 32) | s_gid[j];
                if ((a > b) == up) {
                    const uint32_t g = s_gid[i], sc = s_score[i];
                    const uint16_t d = s_perm[i];
                    s_gid[i] = s_gid[j]; s_score[i] = s_score[j]; s_perm[i] = s_perm[j];
                    s_gid[j] = g; s_score[j] = sc; s_perm[j] = d;
                }
            }
            __syncthreads();
        }
    }
    mmgpu_pf_hit *out = A.out_hits + (size_t)q * A.out_stride;
    for (uint32_t k = threadIdx.x; k < nsel; k += 256) {
        mmgpu_pf_hit h;
        h.id = s_gid[k];
        h.score = (int32_t)(0xFFFFFFFFu - s_score[k]);
        h.diagonal = s_perm[k];
        h.reserved = 0;
        out[has_ident + k] = h;
    }
    if (threadIdx.x == 0) {
        if (has_ident && max_hits > 0) {   // self hit first, score USHRT_MAX (getResult :408-424)
            mmgpu_pf_hit h;
            h.id = ident;
            h.score = 65535;
            h.diagonal = 0;
            h.reserved = 0;
            out[0] = h;
        }
        A.out_counts[q] = (max_hits > 0 ? has_ident : 0u) + nsel;
        if (A.out_flags) A.out_flags[q] = sh_inexact;
    }
}

// one workgroup per query: keep the hits whose target this shard holds, in list order (wave ballots + a running count)
__global__ __launch_bounds__(256) void pf_localize_kernel(PfLocalizeArgs A) {
    __shared__ uint32_t wave_count[4];
    __shared__ uint32_t run;
    const uint32_t q = blockIdx.x;
    const uint32_t n = min(A.counts[q], A.stride);
    const mmgpu_pf_hit *src = A.hits + (size_t)q * A.stride;
    mmgpu_pf_hit *dst = A.local_hits + (size_t)q * A.stride;
    uint32_t *slot = A.local_slot + (size_t)q * A.stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (uint32_t k0 = 0; k0 < n; k0 += 256) {
        const uint32_t k = k0 + threadIdx.x;
        mmgpu_pf_hit h;
        bool own = false;
        if (k < n) {
            h = src[k];
            own = A.shard_of[h.id] == A.shard;
        }
        const uint64_t m = __ballot(own);
        if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = run;
        for (uint32_t w = 0; w < wave; w++) before += wave_count[w];
        if (own) {
            const uint32_t p = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            h.id = A.local_id[h.id];
            dst[p] = h;
            slot[p] = k;
        }
        __syncthreads();
        if (threadIdx.x == 0) run += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
    const uint32_t kept = run;
    for (uint32_t k = kept + threadIdx.x; k < A.stride; k += 256) {
        mmgpu_pf_hit z;
        z.id = 0; z.score = 0; z.diagonal = 0; z.reserved = 0;
        dst[k] = z;
        slot[k] = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) A.local_counts[q] = kept;
}

// Alignment records of the pairs this rank owns -> a dense send buffer (order is free: every record carries its slot).
// One atomic per wavefront; records beyond the buffer are counted, not written.
__global__ __launch_bounds__(256) void sw_owned_pack_kernel(SwOwnedPackArgs A) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t q = (uint32_t)(idx / A.stride), k = (uint32_t)(idx % A.stride);
    const bool valid = q < A.nq && k < min(A.local_counts[q < A.nq ? q : 0], A.stride);
    const uint64_t m = __ballot(valid);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&A.counter[0], (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (!valid) return;
    const uint32_t p = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (p >= A.cap) {
        atomicAdd(&A.counter[1], 1u);
        return;
    }
    SwOwnedRec r;
    r.hit = A.res[idx];
    r.slot = q * A.stride + A.local_slot[idx];
    r.pad = 0;
    A.send[p] = r;
}

// every rank's records -> the dense array in merged-list order (each slot is owned by exactly one rank)
__global__ __launch_bounds__(256) void sw_owned_scatter_kernel(SwOwnedScatterArgs A) {
    const uint32_t r = blockIdx.y;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t packed = A.counters[2 * r], lost = A.counters[2 * r + 1];
    const uint32_t n = min(packed, A.cap);
    if (i == 0) {
        atomicAdd(&A.status[0], n);
        if (lost || packed > A.cap) atomicAdd(&A.status[1], 1u);
    }
    if (i >= n) return;
    const SwOwnedRec rec = A.recv[(size_t)r * A.cap + i];
    if (rec.slot < A.n_slots) A.full[rec.slot] = rec.hit;
}

}  // namespace

hipError_t launch_sw_owned_pack(const SwOwnedPackArgs &A, hipStream_t s) {
    const uint64_t n = (uint64_t)A.nq * A.stride;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_owned_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_sw_owned_scatter(const SwOwnedScatterArgs &A, hipStream_t s) {
    if (A.cap == 0 || A.n_ranks == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_owned_scatter_kernel, dim3((A.cap + 255) / 256, A.n_ranks), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_xmerge(const PfXMergeArgs &A, hipStream_t s) {
    if (A.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_xmerge_kernel, dim3(A.nq), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_localize(const PfLocalizeArgs &A, hipStream_t s) {
    if (A.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_localize_kernel, dim3(A.nq), dim3(256), 0, s, A);
    return hipGetLastError();
}

}  // namespace mmgpu
