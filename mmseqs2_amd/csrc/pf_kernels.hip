// k-mer / double-diagonal / ungapped prefilter for gfx950 (CDNA4): device side.
//
// What it computes: QueryMatcher::matchQuery (src/prefiltering/QueryMatcher.cpp:103-241) for amino-acid
// queries with diagonal scoring, bit for bit (hit set, scores, diagonals, order):
//   a5  KmerGenerator::generateKmerList (KmerGenerator.cpp:108-184)            -> pf_kmers_kernel
//   a6  QueryMatcher::match gather of index lists (QueryMatcher.cpp:243-376)    -> pf_kmers_kernel<true> + pf_split_kernel
//   a7  CacheFriendlyOperations::findDuplicates (.cpp:38-49,185-278)            -> pf_replay_kernel
//   a8  UngappedAlignment::align (UngappedAlignment.cpp:36-57,423-437)          -> pf_ungapped_kernel
//   a9  keepMaxElement (.cpp:354-384), computeScoreThreshold (QueryMatcher.h:211-221),
//       radixSortByScoreSize/rescoreHits/getResult (QueryMatcher.cpp:401-458,536-586), final sort -> pf_keepmax_kernel
//       + pf_select_kernel
//
// The CPU algorithm is order dependent: the "double hit" test compares the low 8 bits
// of the diagonal of CONSECUTIVE index entries of one target in arrival order (query position, then similar
// k-mer rank, then index order), and ties at the --max-seqs cut are broken by the CPU's bin order.  The mapping
// below keeps that order without ever sorting the ~4e5 entries a query touches:
//   * similar k-mers and their index lists are enumerated by one wavefront per query position, in the reference's
//     order, and laid out by prefix sums, so every index entry has a well defined ARRIVAL INDEX;
//   * the arrival stream of a query is cut into tiles of PF_T entries; one workgroup gathers a tile (random 6-byte
//     index reads, the HBM-bound part), splits it STABLY into B bins by target id (id & (B-1), like the CPU's
//     cache bins but sized for LDS: <= 4096 targets per bin) in LDS and writes it out grouped by bin;
//   * one wavefront per (query, bin) then replays its bin in arrival order with the CPU's own state machine held
//     in a 16 KB LDS table (previous diagonal / last emitted diagonal per target); equal targets inside one
//     64-entry round are resolved with ballot matching, so the sequential semantics are exact;
//   * the same wavefront scores its candidates (ungapped Kadane on the diagonal) and keeps one best element per
//     target; a workgroup per query selects the top max_hits by (score, CPU bin order, arrival index) with a
//     radix select and sorts them.
// Integer/byte work throughout: no MFMA; the rooflines are HBM (gather, split) and LDS/VALU issue (replay).
#include <type_traits>

#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// __ballot() takes an int: a bool argument is materialised (v_cndmask 0 / 1) and compared with zero again - two VALU
// instructions per ballot in kernels that are bound by instruction issue.  The builtin takes the condition as it is.
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }

// Inclusive scans over the wavefront with DPP moves (row shifts inside the 16-lane rows, then the row ends handed to the
// following rows): six VALU operations instead of six ds_bpermute round trips.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return (uint32_t)x;
}
__device__ __forceinline__ int wave_incl_max_scan(int v) {      // v >= 0
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));
    return v;
}

// Largest lane m with start[m] <= x, for per-lane non-decreasing `start` with start[0] <= x.  Every lane of
// the wave must call this (it shuffles).
__device__ __forceinline__ int seg_find(uint32_t start_mine, uint32_t x) {
    int lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
        const uint32_t s = __shfl(start_mine, lo + step);
        if (s <= x) lo += step;
    }
    return lo;
}

// Lanes of the wave whose `key` (low nbits) equals mine, among lanes with active == true.
__device__ __forceinline__ uint64_t match_lanes(uint32_t key, int nbits, bool active) {
    const uint64_t act = ballot(active);
    uint32_t lo = (uint32_t)act, hi = (uint32_t)(act >> 32);
    for (int b = 0; b < nbits; b++) {
        // per bit: m &= (my bit set ? lanes with the bit : lanes without) = m & ~(ballot ^ my bit spread over the word):
        // one v_bfe_i32, one compare, one v_bitop3 per half
        const int ext = __builtin_amdgcn_sbfe((int)key, (unsigned)b, 1u);      // 0 or -1
        const uint64_t bal = ballot(ext != 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bal, (uint32_t)ext, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bal >> 32), (uint32_t)ext, 0x90);
    }
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ int highest_lane(uint64_t m) { return 63 - __clzll((long long)m); }

// Largest index in [lo, hi] whose value (base[idx * stride], non-decreasing, base[lo * stride] <= key) is <= key.
// The whole wavefront probes 64 evenly spaced elements per round: log64 instead of log2 dependent memory round trips.
__device__ __forceinline__ uint32_t wave_search_le(const uint32_t *base, uint32_t stride, uint32_t lo, uint32_t hi, uint32_t key) {
    const uint32_t lane = (uint32_t)lane_id();
    while (hi > lo) {
        const uint32_t span = hi - lo + 1;
        const uint32_t step = (span + 63u) / 64u;
        const uint32_t idx = lo + lane * step;
        const bool in = idx <= hi;
        const uint32_t v = in ? base[(size_t)idx * stride] : 0xFFFFFFFFu;
        const uint64_t le = ballot(in && v <= key);   // a prefix of the lanes
        const uint32_t k = (uint32_t)__popcll(le) - 1u;
        lo += k * step;
        hi = min(hi, lo + step - 1u);
    }
    return lo;
}

// Candidate k of a (query, bin) bucket: the first PF_CAND0 live in a dense [bucket][PF_CAND0] array (a bucket holds ~12
// candidates at configs[2]; the dense array keeps the replay/score/keepmax kernels inside a few hundred MB instead of
// a region as large as all index entries), later ones in the bucket's slice of the entry-sized overflow array.
__device__ __forceinline__ PfCand *cand_slot(const PfDedupArgs &A, uint64_t bucket, uint32_t k) {
    return k < (uint32_t)PF_CAND0 ? A.cand_small + bucket * PF_CAND0 + k : A.cand + (A.cand_base[bucket] - A.cand_origin) + k;
}

struct __attribute__((packed, aligned(4))) U32Pair {   // two adjacent uint32 at 4-byte alignment: global_load_dwordx2
    uint32_t a, b;
};
struct __attribute__((packed, aligned(4))) U32Quad {
    uint32_t a, b, c, d;
};

// XCD-aware order of a grid's workgroups.  The dispatcher places workgroup b on XCD b % 8 (observed, not a contract: a wrong guess
// only costs speed); each XCD has its own 4 MB L2.  swz gives XCD x the x-th CONTIGUOUS eighth of the work items, in order, so
// that items that are neighbours in the work order meet in one L2 (bijective for any grid size).
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t bid, uint32_t nwg) {
    const uint32_t xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (bid >> 3);
}

// (the work order of the similar-k-mer kernels - PfKmerArgs::order - is built in pf_order.hip)

// ---------------------------------------------------------------------------------------------------------
// Compact offset table (round 5): blocks of { uint32 base = offsets[first k-mer of the block], one uint8 list length per k-mer }.
// One aligned request answers (start, length); what matters is the footprint of a work group's look-ups (pf_order.hip): a stretch
// of 8000 k-mers is ~10 KB instead of 32 KB, ~90 stretches per 3-mer group fit an XCD's L2 several groups over.  A block with a
// list of 255 entries or more has bit 31 of its base set: its k-mers are answered by the full table (databases of 2^31 index
// entries or more keep to the full table altogether, pf_api.hip).  Block size: 16 bytes = base + 12 lengths - ONE dwordx4 load
// per look-up (a wavefront's 64 look-ups are 64 different lines: the address unit takes a clock per lane and instruction, so the
// 32-byte form - base + 28 lengths, two loads, MMGPU_PF_COFS32 - pays that twice for 15 % less footprint;
// profiles/r05_exp_pf_cofs_block.txt).
constexpr uint32_t PF_COFS_KMERS = 12, PF_COFS_DWORDS = 4;

__global__ __launch_bounds__(256) void pf_cofs_kernel(const uint32_t *offsets, uint64_t table, uint32_t *cofs) {
    const uint64_t blk = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t k0 = blk * PF_COFS_KMERS;
    if (k0 >= table) return;
    uint32_t w[PF_COFS_DWORDS];
    const uint32_t base = offsets[k0];
    bool big = false;
    uint32_t prev = base;
#pragma unroll
    for (uint32_t d = 1; d < PF_COFS_DWORDS; d++) w[d] = 0;
#pragma unroll
    for (uint32_t r = 0; r < PF_COFS_KMERS; r++) {
        uint32_t n = 0;
        if (k0 + r < table) {
            const uint32_t next = offsets[k0 + r + 1];
            n = next - prev;
            prev = next;
        }
        big |= n >= 255u;
        w[1 + (r >> 2)] |= (n & 0xFFu) << ((r & 3u) * 8u);
    }
    w[0] = base | (big ? 0x80000000u : 0u);
#pragma unroll
    for (uint32_t d = 0; d < PF_COFS_DWORDS; d++) cofs[blk * PF_COFS_DWORDS + d] = w[d];
}

// A list record is written once and read once, by the split kernel, after every record of the batch has been written: stored
// with the non-temporal hint it does not push the offset-table blocks (the lines this kernel lives on) out of the L2.
__device__ __forceinline__ void pf_store_list(PfList *dst, uint32_t start, uint32_t len, uint32_t lprefix, uint32_t pos) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {start, len, lprefix, pos};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
}

// (start, length) of the index list of `kmer`
__device__ __forceinline__ void pf_lookup(const PfKmerArgs &A, uint32_t kmer, uint32_t &start, uint32_t &len) {
    if (A.cofs) {
        const uint32_t blk = kmer / PF_COFS_KMERS, r = kmer - blk * PF_COFS_KMERS;
        uint32_t base, w[PF_COFS_DWORDS - 1];
        {
            const uint4 lo = A.cofs[(size_t)blk * (PF_COFS_DWORDS / 4)];
            base = lo.x; w[0] = lo.y; w[1] = lo.z; w[2] = lo.w;
        }
        if (!(base >> 31)) {
            uint32_t sum = 0, word = 0;
#pragma unroll
            for (int d = 0; d < (int)PF_COFS_DWORDS - 1; d++) {
                const int rel = (int)r - 4 * d;      // bytes of this dword that lie before the k-mer's own
                const uint32_t mask = rel >= 4 ? 0xFFFFFFFFu : (rel <= 0 ? 0u : (1u << (8 * rel)) - 1u);
                sum = __builtin_amdgcn_sad_u8(w[d] & mask, 0u, sum);
                word = (rel >= 0 && rel < 4) ? w[d] : word;
            }
            start = base + sum;
            len = (word >> ((r & 3u) * 8u)) & 0xFFu;
            return;
        }
    }
    const U32Pair o = *reinterpret_cast<const U32Pair *>(A.offsets + kmer);   // one 8-byte request
    start = o.a;
    len = o.b - o.a;
}

// ---------------------------------------------------------------------------------------------------------
// a5 + first half of a6: one wavefront per query position.
//   EMIT = false: nsim[gp] = number of similar k-mers of the window starting at gp
//   EMIT = true : lists[list_base[gp] + r] = index list of the r-th similar k-mer, pos_entries[gp] = sum of lengths
// k = 6: the k-mer splits into two 3-mers (KmerGenerator::setDivideStrategy, KmerGenerator.cpp:42-87); row A / row B
// are the score-sorted 3-mer rows of the first / last three window residues.  k = 5 (round 6): the split is (2, 3) - case
// kmerSize % 3 == 2, reversed (:73-86) - i.e. row A is the 2-mer row of the first two residues and row B's index counts ka^2.  Order of the output list:
// i over row A while sA[i] >= thr - sB[0], j over row B while sB[j] >= thr - sA[i]  (calculateArrayProduct :187-216).
template <bool EMIT>
__global__ __launch_bounds__(256) void pf_kmers_kernel(PfKmerArgs A) {
    const int lane = lane_id();
    uint32_t gp = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (A.order) {      // grouped by the window's last 3-mer, a contiguous share of the sorted positions per XCD (pf_order.hip)
        gp = xcd_contiguous(blockIdx.x, gridDim.x) * 4u + (threadIdx.x >> 6);
        if (gp >= A.n_pos) return;
        gp = A.order[gp];
    }
    if (gp >= A.n_pos) return;
    if (A.q_kind && A.q_kind[gp]) return;   // position of a profile query: pf_kmers_prof_kernel
    const int thr = A.q_thr[gp];
    if (thr < 0) {   // no window here, or the window contains X (QueryMatcher.cpp:264-268)
        if (lane == 0) {
            if (EMIT) A.pos_entries[gp] = 0; else A.nsim[gp] = 0;
        }
        return;
    }
    const uint8_t *q = A.q_res + gp;
    const uint32_t ka = A.kalph;
    const bool k5 = A.k == 5;      // (uniform over the launch)
    const uint32_t rowA = k5 ? q[A.pat[0]] + ka * q[A.pat[1]] : q[A.pat[0]] + ka * (q[A.pat[1]] + ka * q[A.pat[2]]);
    const uint32_t rowB = k5 ? q[A.pat[2]] + ka * (q[A.pat[3]] + ka * q[A.pat[4]]) : q[A.pat[3]] + ka * (q[A.pat[4]] + ka * q[A.pat[5]]);
    const uint32_t n3 = A.n3;
    const uint32_t nrowA = k5 ? ka * ka : n3;      // elements of row A = what row B's index is multiplied with (Indexer::powers)
    const int16_t *sA = k5 ? A.s2 + (size_t)rowA * nrowA : A.s3 + (size_t)rowA * n3;
    const uint32_t *iA = k5 ? A.i2 + (size_t)rowA * nrowA : A.i3 + (size_t)rowA * n3;
    const int16_t *sB = A.s3 + (size_t)rowB * n3;
    const uint32_t *iB = A.i3 + (size_t)rowB * n3;
    const int cutoff1 = (int)(short)(thr - (int)sB[0]);
    // number of entries of a score-sorted row with score >= c: one lookup in the per-row cumulative table
    const uint16_t *cumA = k5 ? A.cum2 + (size_t)rowA * A.cum2_w : A.cum3 + (size_t)rowA * A.cum_w;
    const uint16_t *cumB = A.cum3 + (size_t)rowB * A.cum_w;
    const int smin = A.score_min, smax = A.score_min + (int)A.cum_w - 2;
    const int sminA = k5 ? A.score2_min : smin, smaxA = k5 ? A.score2_min + (int)A.cum2_w - 2 : smax;
    const uint32_t nA = cutoff1 <= sminA ? nrowA : (cutoff1 > smaxA ? 0u : (uint32_t)cumA[cutoff1 - sminA]);
    uint32_t nlists = 0, running = 0;
    uint32_t lbase = 0;
    if (EMIT) lbase = A.list_base[gp];
    for (uint32_t c0 = 0; c0 < nA; c0 += 64) {
        const uint32_t ia = c0 + (uint32_t)lane;
        const bool inA = ia < nA;
        uint32_t ni = 0;
        if (inA) {
            const int cutoff2 = (int)(short)(thr - (int)sA[ia]);
            ni = cutoff2 <= smin ? n3 : (cutoff2 > smax ? 0u : (uint32_t)cumB[cutoff2 - smin]);
        }
        const uint32_t incl = wave_incl_scan(ni);
        const uint32_t total = __shfl(incl, 63);
        if (EMIT) {
            const uint32_t excl = incl - ni;
            const uint32_t my_idx = inA ? iA[ia] : 0u;
            for (uint32_t r0 = 0; r0 < total; r0 += 64) {
                const uint32_t x = r0 + (uint32_t)lane;
                const bool act = x < total;
                const int m = seg_find(excl, x);
                const uint32_t ex_m = __shfl(excl, m);
                const uint32_t k_a = __shfl(my_idx, m);
                uint32_t start = 0, len = 0;
                if (act) {
                    const uint32_t kmer = k_a + iB[x - ex_m] * nrowA;
                    // (sparse index: the bit table says whether the list is empty before a sector of the offset table is touched)
                    if (!A.nonempty || ((A.nonempty[kmer >> 5] >> (kmer & 31u)) & 1u)) pf_lookup(A, kmer, start, len);
                }
                const uint32_t li = wave_incl_scan(len);
                if (act) pf_store_list(&A.lists[(size_t)lbase + nlists + x], start, len, running + li - len, gp);
                running += __shfl(li, 63);
            }
        }
        nlists += total;
    }
    if (lane == 0) {
        if (EMIT) A.pos_entries[gp] = running; else A.nsim[gp] = nlists;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k = 7: the k-mer splits into (2, 2, 3) residues (setDivideStrategy case kmerSize % 3 == 1, reversed,
// KmerGenerator.cpp:57-86).  generateKmerList runs two products: row A (2-mer of residues 0,1) x row B (2-mer of 2,3)
// with the branch-and-bound cutoffs, then every element of that intermediate list, in order, x row C (3-mer of 4..6)
// with cutoff thr - score (the second product has cutoff1 = -1000, i.e. no early exit, :165).  Same output contract as
// pf_kmers_kernel.
template <bool EMIT>
__global__ __launch_bounds__(256) void pf_kmers7_kernel(PfKmerArgs A) {
    const int lane = lane_id();
    uint32_t gp = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (A.order) {      // grouped by the window's last 3-mer, a contiguous share of the sorted positions per XCD (pf_order.hip)
        gp = xcd_contiguous(blockIdx.x, gridDim.x) * 4u + (threadIdx.x >> 6);
        if (gp >= A.n_pos) return;
        gp = A.order[gp];
    }
    if (gp >= A.n_pos) return;
    if (A.q_kind && A.q_kind[gp]) return;
    const int thr = A.q_thr[gp];
    if (thr < 0) {
        if (lane == 0) {
            if (EMIT) A.pos_entries[gp] = 0; else A.nsim[gp] = 0;
        }
        return;
    }
    const uint8_t *q = A.q_res + gp;
    const uint32_t ka = A.kalph;
    const uint32_t n2 = ka * ka, n3 = A.n3;
    const uint32_t rowA = q[A.pat[0]] + ka * q[A.pat[1]];
    const uint32_t rowB = q[A.pat[2]] + ka * q[A.pat[3]];
    const uint32_t rowC = q[A.pat[4]] + ka * (q[A.pat[5]] + ka * q[A.pat[6]]);
    const int16_t *sA = A.s2 + (size_t)rowA * n2;
    const uint32_t *iA = A.i2 + (size_t)rowA * n2;
    const int16_t *sB = A.s2 + (size_t)rowB * n2;
    const uint32_t *iB = A.i2 + (size_t)rowB * n2;
    const uint32_t *iC = A.i3 + (size_t)rowC * n3;
    const uint16_t *cumA = A.cum2 + (size_t)rowA * A.cum2_w;
    const uint16_t *cumB = A.cum2 + (size_t)rowB * A.cum2_w;
    const uint16_t *cumC = A.cum3 + (size_t)rowC * A.cum_w;
    const int min2 = A.score2_min, max2 = A.score2_min + (int)A.cum2_w - 2;
    const int min3 = A.score_min, max3 = A.score_min + (int)A.cum_w - 2;
    auto count2 = [&](const uint16_t *cum, int c) -> uint32_t { return c <= min2 ? n2 : (c > max2 ? 0u : (uint32_t)cum[c - min2]); };
    auto count3 = [&](const uint16_t *cum, int c) -> uint32_t { return c <= min3 ? n3 : (c > max3 ? 0u : (uint32_t)cum[c - min3]); };
    const int hB = (int)sB[0], hC = (int)A.s3[(size_t)rowC * n3];
    const int rest0 = (int)(short)(hB + (int)(short)hC), rest1 = (int)(short)hC;
    const int cutoff1 = (int)(short)(thr - rest0);
    const uint32_t nA = count2(cumA, cutoff1);
    const uint32_t mult2 = n2, mult3 = n2 * n2;   // Indexer::powers[2], powers[4]
    uint32_t nlists = 0, running = 0;
    uint32_t lbase = 0;
    if (EMIT) lbase = A.list_base[gp];
    for (uint32_t c0 = 0; c0 < nA; c0 += 64) {
        const uint32_t ia = c0 + (uint32_t)lane;
        const bool inA = ia < nA;
        const int scA = inA ? (int)sA[ia] : 0;
        const uint32_t idxA = inA ? iA[ia] : 0u;
        uint32_t n1 = 0;
        if (inA) n1 = count2(cumB, (int)(short)(thr - scA - rest1));
        const uint32_t incl1 = wave_incl_scan(n1);
        const uint32_t total1 = __shfl(incl1, 63);
        const uint32_t excl1 = incl1 - n1;
        for (uint32_t e0 = 0; e0 < total1; e0 += 64) {          // intermediate elements (i, j), 64 at a time
            const uint32_t x = e0 + (uint32_t)lane;
            const bool actE = x < total1;
            const int m = seg_find(excl1, x);
            const uint32_t ex_m = __shfl(excl1, m);
            const int sc_i = __shfl(scA, m);
            const uint32_t idx_i = __shfl(idxA, m);
            int sc_e = 0;
            uint32_t idx_e = 0, n2e = 0;
            if (actE) {
                const uint32_t j = x - ex_m;
                sc_e = (int)(short)(sc_i + (int)sB[j]);
                idx_e = idx_i + iB[j] * mult2;
                n2e = count3(cumC, (int)(short)(thr - sc_e));
            }
            const uint32_t incl2 = wave_incl_scan(n2e);
            const uint32_t total2 = __shfl(incl2, 63);
            if (EMIT) {
                const uint32_t excl2 = incl2 - n2e;
                for (uint32_t r0 = 0; r0 < total2; r0 += 64) {
                    const uint32_t y = r0 + (uint32_t)lane;
                    const bool act = y < total2;
                    const int me = seg_find(excl2, y);
                    const uint32_t ex_e = __shfl(excl2, me);
                    const uint32_t k_e = __shfl(idx_e, me);
                    uint32_t start = 0, len = 0;
                    if (act) {
                        const uint32_t kmer = k_e + iC[y - ex_e] * mult3;
                        pf_lookup(A, kmer, start, len);
                    }
                    const uint32_t li = wave_incl_scan(len);
                    if (act) pf_store_list(&A.lists[(size_t)lbase + nlists + y], start, len, running + li - len, gp);
                    running += __shfl(li, 63);
                }
            }
            nlists += total2;
        }
    }
    if (lane == 0) {
        if (EMIT) A.pos_entries[gp] = running; else A.nsim[gp] = nlists;
    }
}

// ---------------------------------------------------------------------------------------------------------
// takeOnlyBestKmer (--exact-kmer-matching; every nucleotide search, Search.cpp:186): the window matches its own k-mer only
// (QueryMatcher.cpp:279-282) - one index list per window, no similar k-mers, no threshold.  One thread per position.
template <bool EMIT>
__global__ __launch_bounds__(256) void pf_kmers_exact_kernel(PfKmerArgs A) {
    const uint32_t gp = blockIdx.x * 256u + threadIdx.x;
    if (gp >= A.n_pos) return;
    if (A.q_thr[gp] < 0) {     // no window here, or the window contains X / N
        if (EMIT) A.pos_entries[gp] = 0; else A.nsim[gp] = 0;
        return;
    }
    if (!EMIT) {
        A.nsim[gp] = 1;
        return;
    }
    const uint8_t *q = A.q_res + gp;
    uint32_t idx = 0, pw = 1;
    for (int i = 0; i < A.k; i++) {
        idx += (uint32_t)q[A.pat[i]] * pw;
        pw *= A.kbase;      // (the index's own base: kalph, or the full alphabet for profile targets)
    }
    const uint32_t o0 = A.offsets[idx], o1 = A.offsets[idx + 1];
    PfList rec;
    rec.start = o0;
    rec.len = o1 - o0;
    rec.lprefix = 0;
    rec.pos = gp;
    A.lists[A.list_base[gp]] = rec;
    A.pos_entries[gp] = o1 - o0;
}

// ---------------------------------------------------------------------------------------------------------
// Profile queries: KmerGenerator::setDivideStrategy(ScoreMatrix **one) (KmerGenerator.cpp:32-41) - k steps of ONE residue
// each; step i multiplies the running list with the 20 score-sorted entries of query position gp + pattern[i]
// (Sequence::nextProfileKmer, Sequence.cpp:354-365), cutoff thr - score - (sum of the best scores of the later steps),
// first step with the early exit at thr - rest[0] (generateKmerList :126-166, calculateArrayProduct :187-216).  The output
// order is the nested-loop order, so the enumeration is a depth-first expansion: each lane holds one element of the
// current level, counts its children (entries of the next row above its cutoff), the children are laid out by a
// prefix sum and expanded 64 at a time; the last level emits the index lists exactly like the sequence kernels.
struct ProfGen {
    const int16_t *sc[8];       // the k rows of this window (scores, descending)
    const uint8_t *le[8];       // their letters
    int rest[8];                // possibleRest
    uint32_t mult[8];           // Indexer::powers
    int thr;
    uint32_t nlists, running, lbase, gp;
};

__device__ __forceinline__ uint32_t prof_count_ge(const int16_t *row, int c) {   // entries of a descending row with score >= c
    uint32_t lo = 0, hi = PF_PROF_LETTERS;      // first index with row[idx] < c
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((int)row[mid] >= c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int LVL, int K, bool EMIT>
__device__ __forceinline__ void prof_expand(const PfKmerArgs &A, ProfGen &G, bool act, int sc, uint32_t idx) {
    const int lane = lane_id();
    uint32_t n = 0;
    if (act) n = prof_count_ge(G.sc[LVL], (int)(short)(G.thr - sc - G.rest[LVL]));
    const uint32_t incl = wave_incl_scan(n);
    const uint32_t total = __shfl(incl, 63);
    const uint32_t excl = incl - n;
    for (uint32_t e0 = 0; e0 < total; e0 += 64) {
        const uint32_t x = e0 + (uint32_t)lane;
        const bool actE = x < total;
        const int m = seg_find(excl, x);
        const uint32_t ex_m = __shfl(excl, m);
        const int sc_p = __shfl(sc, m);
        const uint32_t idx_p = __shfl(idx, m);
        int sc_c = 0;
        uint32_t idx_c = 0;
        if (actE) {
            const uint32_t j = x - ex_m;
            sc_c = (int)(short)(sc_p + (int)G.sc[LVL][j]);
            idx_c = idx_p + (uint32_t)G.le[LVL][j] * G.mult[LVL];
        }
        if constexpr (LVL + 1 < K) {
            prof_expand<LVL + 1, K, EMIT>(A, G, actE, sc_c, idx_c);
        } else {
            const uint32_t cnt = min(64u, total - e0);
            if (EMIT) {
                uint32_t start = 0, len = 0;
                if (actE) {
                    const U32Pair o = *reinterpret_cast<const U32Pair *>(A.offsets + idx_c);
                    start = o.a;
                    len = o.b - o.a;
                }
                const uint32_t li = wave_incl_scan(len);
                if (actE) {
                    PfList rec;
                    rec.start = start;
                    rec.len = len;
                    rec.lprefix = G.running + li - len;
                    rec.pos = G.gp;
                    A.lists[(size_t)G.lbase + G.nlists + (uint32_t)lane] = rec;
                }
                G.running += __shfl(li, 63);
            }
            G.nlists += cnt;
        }
    }
}

template <int K, bool EMIT>
__global__ __launch_bounds__(256) void pf_kmers_prof_kernel(PfKmerArgs A) {
    const int lane = lane_id();
    const uint32_t gp = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (gp >= A.n_pos) return;
    if (!A.q_kind[gp]) return;             // position of a sequence query
    const int thr = A.q_thr[gp];
    if (thr < 0) {
        if (lane == 0) {
            if (EMIT) A.pos_entries[gp] = 0; else A.nsim[gp] = 0;
        }
        return;
    }
    ProfGen G;
    uint32_t pw = 1;
#pragma unroll
    for (int i = 0; i < K; i++) {
        G.sc[i] = A.prof_score + (size_t)(gp + A.pat[i]) * PF_PROF_LETTERS;
        G.le[i] = A.prof_letter + (size_t)(gp + A.pat[i]) * PF_PROF_LETTERS;
        G.mult[i] = pw;
        pw *= A.kalph;
    }
    G.rest[K - 1] = 0;
#pragma unroll
    for (int i = K - 1; i >= 1; i--) G.rest[i - 1] = (int)(short)((int)G.sc[i][0] + G.rest[i]);
    G.thr = thr;
    G.nlists = 0;
    G.running = 0;
    G.gp = gp;
    G.lbase = EMIT ? A.list_base[gp] : 0u;
    // first step: entries of row 0 with score >= thr - rest[0]
    const uint32_t n0 = prof_count_ge(G.sc[0], (int)(short)(thr - G.rest[0]));
    const bool act = (uint32_t)lane < n0;
    const int sc0 = act ? (int)G.sc[0][lane] : 0;
    const uint32_t idx0 = act ? (uint32_t)G.le[0][lane] : 0u;
    prof_expand<1, K, EMIT>(A, G, act, sc0, idx0);
    if (lane == 0) {
        if (EMIT) A.pos_entries[gp] = G.running; else A.nsim[gp] = G.nlists;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Exclusive scan of in[q_off[q] .. q_off[q+1]) per query (one workgroup per query), written relative to the query
// (+ base[q] when base != nullptr); totals[q] = the query's sum (64-bit).  out has one extra element per batch:
// the last workgroup also writes out[n_pos] = base[nq-1] + total so that out[gp+1] is valid for every gp.
__global__ __launch_bounds__(256) void pf_scan_kernel(const uint32_t *in, const uint32_t *q_off, uint32_t nq,
                                                      const uint64_t *base, uint32_t *out, uint64_t *totals) {
    __shared__ uint32_t wsum[4];
    __shared__ uint64_t carry_s;
    const uint32_t q = blockIdx.x;
    const uint32_t p0 = q_off[q], p1 = q_off[q + 1];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const uint64_t b = base ? base[q] : 0ull;
    for (uint32_t c = p0; c < p1; c += 256) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < p1 ? in[i] : 0u;
        const uint32_t incl = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const uint64_t carry = carry_s;
        if (i < p1 && out) out[i] = (uint32_t)(b + carry + woff + incl - v);
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (totals) totals[q] = carry_s;
        if (out && q == nq - 1) out[p1] = (uint32_t)(b + carry_s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// second half of a6 + hashIndexEntry (CacheFriendlyOperations.cpp:341-351): one workgroup per tile of PF_T
// arrival-ordered index entries of one query.  Gathers (seqId, position_j) -> (id, diagonal = i - j), splits the
// tile stably by bin = id & (B-1), writes it grouped by bin plus the B+1 bin offsets of the tile.
// Staged entry word (LDS): id | diagonal << 32 | slot-in-tile << 48  (slot = arrival index - tile start).
// Written out (round 5): 4 bytes per entry - (id >> log2 B) | low diagonal byte << 12 | slot << 20: the bin is where the entry
// stands, the 12 id bits above it number the <= 4096 targets of a bin, the replay's state machine needs the low diagonal byte only -
// plus the high diagonal byte in a byte array of the same layout, which only the ~1 % of entries that become candidates are
// looked up in.  Every request to memory costs the same whatever it carries (~55 G/s chip wide, pf_order.hip): the tiles were
// 29 GB written and 29 GB read back per 10 000 queries, now 18 and 14.5.
// Wavefronts per tile.  The kernel waits on dependent LDS / cross-lane / gather latencies, so its throughput follows the
// number of resident wavefronts (measured: half the occupancy = 1.65x the time); the tile's 32 KB stage allows four
// workgroups per CU, eight wavefronts each fill the SIMDs' eight slots (the per-(wave, bin) counters are 16 bit for that).
constexpr int SPW = 8;
constexpr int SPLIT_WAVES_PER_EU = 6;      // (profiles/r05_exp_pf_*: 4, 8 and padded-LDS occupancies measured slower)
__global__ __launch_bounds__(SPW * 64) __attribute__((amdgpu_waves_per_eu(SPLIT_WAVES_PER_EU))) void pf_split_kernel(PfSplitArgs A) {
    __shared__ uint64_t stage[PF_T];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    uint16_t *cnt = reinterpret_cast<uint16_t *>(dyn_lds);   // [SPW][B]: counts <= PF_T / SPW, then tile offsets < PF_T
    __shared__ uint32_t wsum[SPW];
    const uint32_t B = A.bins;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t t = blockIdx.x;
    const uint32_t q = A.tile_q[t];
    const uint32_t a0 = A.tile_idx[t] * (uint32_t)PF_T;
    const uint32_t q_entries = A.q_entries[q];
    const uint32_t tile_n = min((uint32_t)PF_T, q_entries - a0);
    const uint32_t qp0 = A.q_off[q], qlen = A.q_off[q + 1] - qp0;

    for (uint32_t k = threadIdx.x; k < (uint32_t)SPW * B; k += (uint32_t)SPW * 64u) cnt[k] = 0;

    // ---- phase A: gather this wave's share of the tile into `stage`, arrival order ----
    const uint32_t wa = a0 + (uint32_t)wave * (PF_T / SPW);
    const uint32_t wb = min(a0 + tile_n, wa + PF_T / SPW);
    if (wa < wb) {
        // position holding arrival index wa (largest p with peb[p] <= wa), then the list inside it
        const uint32_t pidx = wave_search_le(A.pos_entry_base + qp0, 1, 0, qlen - 1, wa);
        const uint32_t gp = qp0 + pidx;
        const uint32_t rel = wa - A.pos_entry_base[gp];
        const uint32_t l0 = wave_search_le(&A.lists[0].lprefix, sizeof(PfList) / 4, A.list_base[gp], A.list_base[gp + 1] - 1, rel);
        uint32_t L = l0;
        const uint32_t Lend = A.list_base[qp0 + qlen];
        uint32_t cur = wa;
        PfList r;
        r.start = 0; r.len = 0; r.lprefix = 0; r.pos = qp0;
        if (L + (uint32_t)lane < Lend) r = A.lists[L + (uint32_t)lane];
        while (cur < wb) {
            const bool valid = L + (uint32_t)lane < Lend;
            // the next 64 list records are requested before this chunk's entries are gathered
            PfList rn;
            rn.start = 0; rn.len = 0; rn.lprefix = 0; rn.pos = qp0;
            if (L + 64u + (uint32_t)lane < Lend) rn = A.lists[L + 64u + (uint32_t)lane];
            const uint32_t es = valid ? A.pos_entry_base[r.pos] + r.lprefix : 0xFFFFFFFFu;
            const uint32_t ee = es + r.len;
            const uint32_t nvalid = min(64u, Lend - L);
            const uint32_t chunk_end = min(wb, (uint32_t)__shfl(ee, (int)nvalid - 1));
            for (uint32_t x0 = cur; x0 < chunk_end; x0 += 256) {
                // four rounds at a time: all index-entry loads are in flight before the first LDS store
                uint32_t eidx[4], ipos[4], slot[4];
                bool act[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t x = x0 + 64u * (uint32_t)k + (uint32_t)lane;
                    act[k] = x < chunk_end;
                    const int m = seg_find(es, x);
                    const uint32_t es_m = __shfl(es, m);
                    const uint32_t st_m = __shfl(r.start, m);
                    const uint32_t gp_m = __shfl(r.pos, m);
                    eidx[k] = st_m + (x - es_m);
                    ipos[k] = gp_m - qp0;
                    slot[k] = x - a0;
                }
                uint64_t ent[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ent[k] = act[k] ? A.idx_entries[eidx[k]] : 0ull;   // seqId | position_j << 32
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (act[k]) {
                        const uint32_t id = (uint32_t)ent[k];
                        const uint32_t diag = (ipos[k] - (uint32_t)(ent[k] >> 32)) & 0xFFFFu;
                        stage[slot[k]] = (uint64_t)id | ((uint64_t)diag << 32) | ((uint64_t)slot[k] << 48);
                    }
                }
            }
            if (chunk_end > cur) cur = chunk_end;
            L += 64;
            r = rn;
            if (L >= Lend) break;
        }
    }
    __syncthreads();

    // ---- phase B: per-(wave, bin) counts and each entry's rank inside its (wave, bin) ----
    int nbits = 0;
    while ((1u << nbits) < B) nbits++;
    constexpr int ROUNDS = PF_T / (64 * SPW);   // entries per thread
    uint64_t ent[ROUNDS];
    uint32_t rk[ROUNDS];
    uint16_t *mycnt = cnt + (uint32_t)wave * B;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t slot = (uint32_t)wave * (PF_T / SPW) + (uint32_t)r * 64u + (uint32_t)lane;
        const bool valid = slot < tile_n;
        ent[r] = valid ? stage[slot] : 0ull;
        const uint32_t bin = (uint32_t)ent[r] & (B - 1);
        const uint64_t m = match_lanes(bin, nbits, valid);
        const uint32_t rank = (uint32_t)__popcll(m & lanes_below(lane));
        uint32_t c = 0;
        if (valid) c = mycnt[bin];
        rk[r] = c + rank;
        if (valid && rank == 0) mycnt[bin] = (uint16_t)(c + (uint32_t)__popcll(m));
    }
    __syncthreads();

    // ---- scan: bin totals -> tile bin offsets; cnt[w][b] becomes the absolute start of (bin b, wave w) ----
    {
        const uint32_t bpt = B >= 256 ? B / 256 : 1;   // bins per thread, contiguous
        const uint32_t b0 = threadIdx.x * bpt;
        uint32_t s = 0;
        if (b0 < B)
            for (uint32_t b = b0; b < b0 + bpt; b++)
                for (int w = 0; w < SPW; w++) s += cnt[(uint32_t)w * B + b];
        const uint32_t incl = wave_incl_scan(s);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - s;
        for (int w = 0; w < wave; w++) run += wsum[w];
        if (b0 < B) {
            uint16_t *bo = A.bin_off + (size_t)t * (B + 1);
            for (uint32_t b = b0; b < b0 + bpt; b++) {
                bo[b] = (uint16_t)run;
                uint32_t tot = 0;
                for (int w = 0; w < SPW; w++) {
                    const uint32_t c = cnt[(uint32_t)w * B + b];
                    cnt[(uint32_t)w * B + b] = (uint16_t)(run + tot);
                    tot += c;
                }
                if (tot) atomicAdd(&A.bucket_count[(size_t)q * B + b], tot);
                run += tot;
            }
            if (b0 + bpt == B) bo[B] = (uint16_t)run;
        }
    }
    __syncthreads();

    // ---- phase C: stable scatter inside LDS (every entry is in a register now: the stage is free), in the output format -
    // the 4-byte words in the first half of the stage, the high diagonal bytes behind them - then coalesced write-out ----
    uint32_t *stage_w = reinterpret_cast<uint32_t *>(stage);
    uint8_t *stage_h = reinterpret_cast<uint8_t *>(stage) + (size_t)PF_T * 4;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t slot = (uint32_t)wave * (PF_T / SPW) + (uint32_t)r * 64u + (uint32_t)lane;
        if (slot < tile_n) {
            const uint32_t id = (uint32_t)ent[r];
            const uint32_t bin = id & (B - 1);
            const uint32_t diag = (uint32_t)(ent[r] >> 32) & 0xFFFFu;
            const uint32_t at = mycnt[bin] + rk[r];
            stage_w[at] = (id >> nbits) | ((diag & 0xFFu) << 12) | ((uint32_t)(ent[r] >> 48) << 20);
            stage_h[at] = (uint8_t)(diag >> 8);
        }
    }
    __syncthreads();
    uint32_t *dst = A.split + (size_t)t * PF_T;
    uint32_t *dst_h = reinterpret_cast<uint32_t *>(A.split_hi + (size_t)t * PF_T);
    const uint32_t *src_h = reinterpret_cast<const uint32_t *>(stage_h);
    for (uint32_t s = threadIdx.x; s < tile_n; s += (uint32_t)SPW * 64u) dst[s] = stage_w[s];
    for (uint32_t s = threadIdx.x; s < (tile_n + 3u) / 4u; s += (uint32_t)SPW * 64u) dst_h[s] = src_h[s];
}

// ---------------------------------------------------------------------------------------------------------
// a8: ungapped score of every candidate: s = max(0, s + P[q_pos][t_res]), best = max s along the diagonal
// (scalarDiagonalScoring, UngappedAlignment.cpp:45-57; overlap as in computeSingelSequenceScores :423-437;
// P = matrix + per-position composition term, createProfile :388-421).
// One wavefront per (query, bin); a 16-lane group per candidate (4 candidates in flight per wavefront); per pass a
// lane owns 16 consecutive diagonal cells (so a group reads 256 contiguous target bytes) and summarises them as the
// function s -> (max(a, s + b), running best max(M, s + P)); a 4-step ordered tree over the group composes the 16
// summaries, lane 0 of the group applies them to the carried score.  Bytes per cell: 1 (target) from HBM, the query
// side stays in L1/L2.
struct Seg {
    int a, b, P, M;
};
__device__ __forceinline__ Seg seg_combine(const Seg &l, const Seg &r) {   // l then r
    Seg o;
    o.b = l.b + r.b;
    o.a = max(r.a, l.a + r.b);
    o.P = max(l.P, l.b + r.P);
    o.M = max(max(l.M, r.M), l.a + r.P);
    return o;
}
// Round 6: a pass of 4 * NW cells per lane (NW = 4: 256 cells per group and pass, NW = 6: 384).  Most diagonals of a protein
// search are 250 - 380 cells long: with 16 cells per lane they took a second pass - a whole pass's instructions for a few dozen
// cells, and a second tree.  Cells past the segment's end count as score 0, which changes no summary (the prefix sums stay at the
// sum of the real cells, a stays a); a lane with no cell at all keeps the identity.
template <int NW>
__device__ __forceinline__ void load_cells(const uint8_t *p, uint32_t (&w)[NW]) {
    static_assert(NW == 4 || NW == 6, "16 or 24 bytes");
    // 4 * NW bytes from an arbitrarily aligned address: a 16-byte request and a 4-byte (12-byte) one on the enclosing aligned dwords
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const uint32_t *a = reinterpret_cast<const uint32_t *>(u & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(u & 3u);
    const U32Quad d = *reinterpret_cast<const U32Quad *>(a);
    w[0] = __builtin_amdgcn_alignbyte(d.b, d.a, sh);
    w[1] = __builtin_amdgcn_alignbyte(d.c, d.b, sh);
    w[2] = __builtin_amdgcn_alignbyte(d.d, d.c, sh);
    if constexpr (NW == 4) {
        w[3] = __builtin_amdgcn_alignbyte(a[4], d.d, sh);
    } else {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 e = *reinterpret_cast<const u32x3 *>(a + 4);      // (dwordx3: 4-byte alignment is enough)
        w[3] = __builtin_amdgcn_alignbyte(e.x, d.d, sh);
        w[4] = __builtin_amdgcn_alignbyte(e.y, e.x, sh);
        w[5] = __builtin_amdgcn_alignbyte(e.z, e.y, sh);
    }
}
template <int NW>
__device__ __forceinline__ Seg seg_cells_n(const uint32_t (&tw)[NW], const uint32_t (&qw)[NW], const uint32_t (&cw)[NW], int nn,
                                           const int8_t *smat) {
    Seg g;
    g.a = 0; g.b = 0; g.P = -(1 << 28); g.M = 0;
#pragma unroll
    for (int k = 0; k < 4 * NW; k++) {
        // (the matrix lies in LDS in rows of 32: the cell's entry is one shift-or away - q * alphabet + t with a run-time alphabet
        // was a 64-bit multiply-add per cell; letters are below 32, what lies past a segment's end is masked below)
        const uint32_t t5 = (tw[k >> 2] >> ((k & 3) * 8)) & 31u;
        const uint32_t q5 = (qw[k >> 2] >> ((k & 3) * 8)) & 31u;
        const int cb = (int)(int8_t)((cw[k >> 2] >> ((k & 3) * 8)) & 0xFFu);
        int x = (int)(int8_t)(smat[(q5 << 5) | t5] + cb);
        x = k < nn ? x : 0;
        g.b += x;
        g.a = max(0, g.a + x);
        g.P = max(g.P, g.b);
        g.M = max(g.M, g.a);
    }
    return g;
}
// profile query: the score of cell k is row (first position + k) of the query's score rows at the target letter
template <int NW>
__device__ __forceinline__ Seg seg_cells_rows_n(const uint32_t (&tw)[NW], const int8_t *rows, int nn) {
    Seg g;
    g.a = 0; g.b = 0; g.P = -(1 << 28); g.M = 0;
#pragma unroll
    for (int k = 0; k < 4 * NW; k++) {
        if (k < nn) {
            const int tb_ = (int)((tw[k >> 2] >> ((k & 3) * 8)) & 0xFFu);
            const int x = (int)rows[k * PF_PROW + (tb_ & (PF_PROW - 1))];
            g.b += x;
            g.a = max(0, g.a + x);
            g.P = max(g.P, g.b);
            g.M = max(g.M, g.a);
        }
    }
    return g;
}
__device__ __forceinline__ Seg seg_tree16(Seg g, int gl) {   // ordered tree over the 16 lanes of a group; result in lane 0
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        Seg r;
        r.a = __shfl_down(g.a, d);
        r.b = __shfl_down(g.b, d);
        r.P = __shfl_down(g.P, d);
        r.M = __shfl_down(g.M, d);
        if ((gl & (2 * d - 1)) == 0) g = seg_combine(g, r);
    }
    return g;
}

// Scores up to 64 candidates of one (query, bin) bucket - lane l holds candidate cb0 + l - and, when the bucket has no
// more than 64 candidates, finishes keepMaxElement for it.  Latency plan: target metadata of the whole chunk in one
// round trip, then the first pass (256 or 384 diagonal cells) of 8 candidates at a time is requested together before any is
// scored.  The query side is read from global memory (L1/L2).
// SLOTS = false (the overflow path's elements, which do not live in candidate slots): the scores only, handed back in *score_out.
template <bool SLOTS = true>
__device__ __forceinline__ uint64_t score_chunk(const PfDedupArgs &A, const int8_t *smat, uint64_t bucket, uint32_t q, uint32_t cb0,
                                                uint32_t nin, uint32_t ncand, PfCand c, int bshift, uint32_t *score_out = nullptr) {
    const int lane = lane_id();
    const int grp = lane >> 4, gl = lane & 15;
    const uint32_t qp0 = A.q_off[q];
    const int qlen = (int)(A.q_off[q + 1] - qp0);
    const uint8_t *qr = A.q_res + qp0;
    const uint8_t *qc = reinterpret_cast<const uint8_t *>(A.q_corr) + qp0;
    const bool has = (uint32_t)lane < nin;
    const int8_t *prows = (A.q_isprof && A.q_isprof[q]) ? A.q_rows + (size_t)qp0 * PF_PROW : nullptr;   // wave-uniform
    int my_len = 0, my_qs = 0;
    unsigned long long my_addr = 0;
    uint64_t cells = 0;
    // A candidate of an ordinary query arrives with the low byte of its diagonal and, in `score`, where its entry stands in the
    // QUERY's split tiles (replay_bucket_impl; the tiles of a batch may hold 2^32 slots and more, a query's do not): the high byte
    // is read here, one round trip for the whole chunk, and written back with the score.  (Overflow-path queries and
    // --diag-score 0 complete their diagonals in the replay / count kernels.)
    if (has && !(A.q_nseg && A.q_nseg[q]))
        c.diag = (uint16_t)((c.diag & 0xFFu) | ((uint32_t)A.split_hi[(size_t)A.q_tile_base[q] * PF_T + c.score] << 8));
    if (has) {
        const int d = (int)(short)c.diag;
        const int tlen = (int)A.t_len[c.id];
        const uint8_t *t = A.t_res + (size_t)A.t_off4[c.id] * 4;
        const int mind = d < 0 ? -d : d;
        int ts = 0;
        if (d >= 0 && mind < qlen) {
            my_len = min(tlen, qlen - mind);
            my_qs = mind;
        } else if (d < 0 && mind < tlen) {
            my_len = min(tlen - mind, qlen);
            ts = mind;
        }
        my_addr = (unsigned long long)reinterpret_cast<uintptr_t>(t + ts);
        cells = (uint64_t)my_len;
        // targets of 32768 residues or more take the reference's computeLongScore (every 65536-shift of the 16-bit
        // diagonal, and a batch quirk of scoreDiagonalAndUpdateHits): the query is handed back to the host
        if (tlen >= 32768 && A.q_flags) atomicOr(&A.q_flags[q], 1u);
    }
    int my_score = 0;
    constexpr int UN = 2;   // candidates per group in flight: 4 groups x UN = 8 per round trip
    // one trip = 8 candidates; NW dwords (4 * NW cells) per lane and pass
    auto trip = [&](auto nw_tag, const uint32_t k0) {
        constexpr int NW = decltype(nw_tag)::value;
        constexpr int CPL = 4 * NW, PASS = 16 * CPL;
        uint32_t tw[UN][NW], qw[UN][NW], cw[UN][NW];
        int len_u[UN], qs_u[UN];
        unsigned long long addr_u[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int src = (int)k0 + u * 4 + grp;          // < 64
            len_u[u] = __shfl(my_len, src);
            qs_u[u] = __shfl(my_qs, src);
            addr_u[u] = __shfl(my_addr, src);
            if ((uint32_t)src >= nin) len_u[u] = 0;
#pragma unroll
            for (int z = 0; z < NW; z++) { tw[u][z] = 0; qw[u][z] = 0; cw[u][z] = 0; }
            if (gl * CPL < len_u[u]) {
                load_cells<NW>(reinterpret_cast<const uint8_t *>((uintptr_t)addr_u[u]) + gl * CPL, tw[u]);
                if (!prows) {
                    load_cells<NW>(qr + qs_u[u] + gl * CPL, qw[u]);
                    load_cells<NW>(qc + qs_u[u] + gl * CPL, cw[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int len = len_u[u];
            Seg g;
            g.a = 0; g.b = 0; g.P = -(1 << 28); g.M = 0;
            if (gl * CPL < len) {
                if (prows) g = seg_cells_rows_n<NW>(tw[u], prows + (size_t)(qs_u[u] + gl * CPL) * PF_PROW, min(CPL, len - gl * CPL));
                else g = seg_cells_n<NW>(tw[u], qw[u], cw[u], min(CPL, len - gl * CPL), smat);
            }
            g = seg_tree16(g, gl);
            int sc = 0, best = 0;
            if (gl == 0 && len > 0) {
                best = max(g.M, g.P);
                sc = max(g.a, g.b);
            }
            int maxlen = len;
            maxlen = max(maxlen, __shfl_xor(maxlen, 16));
            maxlen = max(maxlen, __shfl_xor(maxlen, 32));
            for (int p0 = PASS; p0 < maxlen; p0 += PASS) {      // diagonals longer than one pass
                const int o = p0 + gl * CPL;
                Seg h;
                h.a = 0; h.b = 0; h.P = -(1 << 28); h.M = 0;
                if (o < len) {
                    uint32_t t2[NW], q2[NW], c2[NW];
                    load_cells<NW>(reinterpret_cast<const uint8_t *>((uintptr_t)addr_u[u]) + o, t2);
                    if (prows) h = seg_cells_rows_n<NW>(t2, prows + (size_t)(qs_u[u] + o) * PF_PROW, min(CPL, len - o));
                    else {
                        load_cells<NW>(qr + qs_u[u] + o, q2);
                        load_cells<NW>(qc + qs_u[u] + o, c2);
                        h = seg_cells_n<NW>(t2, q2, c2, min(CPL, len - o), smat);
                    }
                }
                h = seg_tree16(h, gl);
                if (gl == 0 && p0 < len) {
                    best = max(best, max(h.M, sc + h.P));
                    sc = max(h.a, sc + h.b);
                }
            }
            // hand the four group results to the candidates' own lanes
            const int r0 = __shfl(best, 0), r1 = __shfl(best, 16), r2 = __shfl(best, 32), r3 = __shfl(best, 48);
            const int rel = lane - ((int)k0 + u * 4);
            if (rel >= 0 && rel < 4) my_score = rel == 0 ? r0 : (rel == 1 ? r1 : (rel == 2 ? r2 : r3));
        }
    };
    for (uint32_t k0 = 0; k0 < nin; k0 += 4 * UN) {
        // the trip's longest diagonal decides the pass width (wave-uniform: the trip's eight candidates sit in lanes k0 .. k0 + 7)
        const bool mine = (uint32_t)lane >= k0 && (uint32_t)lane < k0 + 4u * UN && has;
        if (ballot(mine && my_len > 256)) trip(std::integral_constant<int, 6>{}, k0);
        else trip(std::integral_constant<int, 4>{}, k0);
    }
    c.score = (uint32_t)my_score;
    if (score_out) *score_out = c.score;
    if constexpr (!SLOTS) return cells;
    if (has) {
        PfCand *slot = cand_slot(A, bucket, cb0 + (uint32_t)lane);
        slot->score = c.score;
        slot->diag = c.diag;
    }
    if (ncand <= 64 && !A.nucl) {
        // keepMaxElement for the whole bucket right here (the common case; larger buckets go to pf_keepmax_kernel):
        // per target the first candidate holding the target's maximum count
        const uint32_t cnt = min(255u, c.score);
        const uint64_t same = match_lanes(c.id >> bshift, 12, has);
        bool win = has;
        uint64_t m = same & ~(1ull << lane);
        while (ballot(m != 0)) {
            const int o = m ? __ffsll((long long)m) - 1 : lane;
            const uint32_t oc = __shfl(cnt, o);
            if (m) {
                if (oc > cnt || (oc == cnt && o < lane)) win = false;
                m &= m - 1;
            }
        }
        win = win && cnt >= A.min_diag_score;
        const uint64_t wb = ballot(win);
        if (wb) {
            PfCand *surv = A.surv + (A.cand_base[(uint64_t)q * A.bins] - A.cand_origin);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.surv_count[q], (uint32_t)__popcll(wb));
            base = __shfl(base, 0);
            if (win) surv[base + (uint32_t)__popcll(wb & lanes_below(lane))] = c;
        }
    }
    return cells;
}

// ---------------------------------------------------------------------------------------------------------
// a7: one wavefront per (query, bin), four per workgroup: replay of the bin's entries in arrival order.
// Per target the CPU keeps `prev` = low byte of the previous entry's diagonal (zero-initialised,
// CacheFriendlyOperations.cpp:186-208) and emits an entry whose byte equals it; the emitted list is then run-length
// de-duplicated per target on that byte (:240-265).  Output: candidates (id, diagonal, arrival index) in arrival
// order at cand[cand_base[bucket] ..), their number in cand_count[bucket].
//
// Round 6: the sequential semantics come from the LDS itself.  A round is 64 consecutive entries of the bucket, one per lane, and
// the CPU's `prev = tmp[id]; tmp[id] = diagonal` is ONE instruction for all of them: ds_mskor_rtn_b32, a byte-granular atomic
// exchange (the word's other three targets keep their bytes).  When several lanes of the instruction name the same target, the
// LDS applies them one after the other IN ASCENDING LANE ORDER - arrival order - so that every lane receives exactly the byte its
// predecessor left, within the round or before it (scripts/probes/lds_atomic_order.hip: 50 M lanes under every degree of conflict,
// no deviation; tests/test_prefilter_gpu.py runs the same probe, and every hit list of the bench run is compared with the
// reference's).  58 % of the rounds of the 10 000 x 1 M batch hold a target more than once (homologs hit on every position);
// rounds 3 - 5 found those lanes with an LDS bit per target, matched their keys and walked predecessor masks - ~70 of the 222
// VALU instructions of a round.  The second pass works the same way on the flagged lanes only: an atomic OR on the target's "has
// emitted" bit (first emission?) and a second byte exchange on the byte last emitted.  No emitter table, no redo list.
// Per wavefront: 4 KB previous bytes + 4 KB emitted bytes + 512 B bits + the first 64 candidates: 9.5 KB, four workgroups per CU.
constexpr int REPLAY_WAVES_PER_EU = 4;
// rounds whose entries are in flight while a round is processed: with ~70 instructions per round the kernel waits for memory, and a
// wavefront's 256 bytes per round are few - the rounds ahead are what keeps the memory system busy
constexpr int REPLAY_PD = 4;      // (1, 2 and 8 measured: gpurun r06h - 25.1 / 25.3 / 27.0 ms against 26.1 before the slots kept their registers)


struct ReplayLds {
    uint32_t prev[4][PF_IDS_PER_BIN / 4];     // a byte per target: low diagonal byte of the target's previous entry
    uint32_t last[4][PF_IDS_PER_BIN / 4];     // a byte per target: diagonal byte of its last emitted entry (read only where `emit` is set)
    uint32_t emit[4][PF_IDS_PER_BIN / 32];    // a bit per target: has emitted
    uint32_t cand[4][3][64];   // first 64 candidates of the bucket: key | diagonal byte << 12, arrival index, index of the entry in the split tiles
    uint8_t mark[4][64];       // segment starts of a round (request)
    int8_t smat[32 * 32];
};

// byte-granular atomic exchange in LDS: the byte `shift / 8` of *word becomes `byte`, the old word comes back.  Lanes of one
// instruction that name the same word are applied in ascending lane order (see above).
__device__ __forceinline__ uint32_t lds_byte_exchange(uint32_t *word, uint32_t shift, uint32_t byte) {
    uint32_t old;
    const uint32_t addr = (uint32_t)(uintptr_t)word;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(0xFFu << shift), "v"(byte << shift) : "memory");
    return old;
}

// SEGS: the query is on the reference's overflow path (databaseHits flushes, nseg > 0).  The ordinary query's round is straight
// line code: no boundary compare, no loop over the pieces of a round.
template <bool SEGS>
__device__ __forceinline__ void replay_bucket_impl(const PfDedupArgs &A, ReplayLds &M, uint64_t bucket) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t B = A.bins;
    const uint32_t q = (uint32_t)(bucket / B), bin = (uint32_t)(bucket % B);
    const uint32_t ntiles = A.q_ntiles[q];
    if (ntiles == 0) {
        if (lane == 0) A.cand_count[bucket] = 0;
        return;
    }
    const uint32_t tb = A.q_tile_base[q];
    uint32_t *P = M.prev[wave];
    uint32_t *L = M.last[wave];
    uint32_t *E = M.emit[wave];
    uint8_t *mark = M.mark[wave];
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    auto clear_state = [&]() {      // (the emitted bytes are only read under a set bit: the bits are what is cleared)
        uint4 *P4 = reinterpret_cast<uint4 *>(P);
        for (int k = lane; k < PF_IDS_PER_BIN / 16; k += 64) P4[k] = make_uint4(0u, 0u, 0u, 0u);
        for (int k = lane; k < PF_IDS_PER_BIN / 32; k += 64) E[k] = 0;
    };
    clear_state();

    uint32_t ncand = 0;
    const uint64_t below = lanes_below(lane);
    // where the bucket's candidates go (cand_slot): read once - a load inside the rounds would make every later wait a full one
    PfCand *const cand_small = A.cand_small + bucket * PF_CAND0;
    PfCand *const cand_big = A.cand + (A.cand_base[bucket] - A.cand_origin);
    const uint8_t *const split_hi_q = A.split_hi + (size_t)tb * PF_T;
    const uint32_t *const split_q = A.split + (size_t)tb * PF_T;
    const uint32_t nseg = SEGS ? A.q_nseg[q] : 0u;
    const uint32_t *segs = SEGS ? A.seg_start + (size_t)q * (PF_MAX_SEG + 2) : nullptr;
    uint32_t cur_seg = 0, next_boundary = SEGS ? segs[1] : 0xFFFFFFFFu;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 64) {
        const uint32_t tl = t0 + (uint32_t)lane;
        uint32_t o0 = 0, n = 0;
        if (tl < ntiles) {
            const uint16_t *bo = A.bin_off + (size_t)(tb + tl) * (B + 1) + bin;
            o0 = bo[0];
            n = (uint32_t)bo[1] - o0;
        }
        const uint32_t incl = wave_incl_scan(n);
        const uint32_t total = __shfl(incl, 63);
        const uint32_t excl = incl - n;
        // Tile whose segment holds the bucket's entry x, for the 64 entries of a round at once (rounds are requested in
        // increasing order): every tile whose non-empty segment starts inside the round marks its slot, a running maximum
        // along the lanes spreads the marks, the segment that reaches into the round from before is the last one of the
        // previous request.  (A search over the segment starts costs six dependent cross-lane round trips per round.)
        int m_carry = 0;
        const uint32_t seg_delta = o0 - excl;      // entry x of the bucket is entry x + seg_delta of its tile
        auto request = [&](uint32_t xr, uint32_t &e_out, uint32_t &tile_out, uint32_t &at_out) {
            const uint32_t x = xr + (uint32_t)lane;
            mark[lane] = 0;
            const uint32_t rel = excl - xr;
            if (n != 0u && rel < 64u) mark[rel] = (uint8_t)lane;
            // (another lane's mark may sit in this lane's slot: the compiler must not forward this lane's own zero to the load)
            asm volatile("" ::: "memory");
            int m = (int)mark[lane];
            m = max(wave_incl_max_scan(m), m_carry);
            m_carry = __builtin_amdgcn_readlane(m, 63);
            const uint32_t d_m = (uint32_t)__shfl((int)seg_delta, m);
            tile_out = t0 + (uint32_t)m;
            at_out = (uint32_t)(x + d_m);      // (the sum wraps in 32 bits)
            // every lane loads (a lane past the end of the group re-reads the query's first slot, one request for all of them): a
            // load under a branch leaves the compiler no lower bound on the loads issued after it, and every wait becomes vmcnt(0)
            e_out = split_q[x < total ? (size_t)tile_out * PF_T + at_out : (size_t)0];
        };
        // One round: the slot's entries are processed, the slot is refilled with the round REPLAY_PD rounds ahead.  The slots are
        // NAMED SCALARS and the round has no condition of its own (a slot past the end of the group has no active lane, its refill
        // loads and marks nothing): as arrays under per-slot branches the compiler kept each array in one register tuple and copied
        // the tuple around every load - s_waitcnt vmcnt(0) before each request, i.e. no round in flight at all.
        auto one_round = [&](const uint32_t x0, uint32_t &e_slot, uint32_t &tile_slot, uint32_t &at_slot) {
            const uint32_t e = e_slot;
            const uint32_t tile_cur = tile_slot;
            const uint32_t where = tile_cur * (uint32_t)PF_T + at_slot;      // the entry's place in the query's tiles of the split arrays
            const bool act = x0 + (uint32_t)lane < total;
            request(x0 + 64u * REPLAY_PD, e_slot, tile_slot, at_slot);
            const uint32_t key = e & 0xFFFu;   // < PF_IDS_PER_BIN
            const uint32_t d8 = (e >> 12) & 0xFFu;
            const uint32_t arr = tile_cur * (uint32_t)PF_T + (e >> 20);
            const uint32_t sh = (key & 3u) * 8u;
            // Overflow path (nseg > 0): the reference flushes databaseHits at segment boundaries and starts the
            // double-diagonal state from scratch (QueryMatcher.cpp:310-346), so a round that straddles a boundary is
            // processed in pieces with the tables cleared in between.
            uint64_t todo = SEGS ? ballot(act) : 1ull;
            while (todo) {
                const bool now = SEGS ? (act && ((todo >> lane) & 1ull) && arr < next_boundary) : act;
                if (!SEGS || ballot(now)) {
                    // first pass (:194-208): the byte the target's previous entry left, mine in its place
                    bool flag = false;
                    if (now) flag = ((lds_byte_exchange(&P[key >> 2], sh, d8) >> sh) & 0xFFu) == d8;
                    if (ballot(flag)) {
                        // second pass (:240-265) over the flagged entries, in their order: the first of a target is kept, a later one
                        // when its byte differs from the one the target emitted last
                        bool keep = false;
                        if (flag) {
                            const uint32_t bit = 1u << (key & 31u);
                            const bool emitted_before = (atomicOr(&E[key >> 5], bit) & bit) != 0u;
                            const uint32_t last_byte = (lds_byte_exchange(&L[key >> 2], sh, d8) >> sh) & 0xFFu;
                            keep = !emitted_before || last_byte != d8;
                        }
                        const uint64_t kb = ballot(keep);
                        if (keep) {
                            PfCand c;
                            c.id = (key << bshift) | bin;
                            c.arr = arr;
                            // ordinary query: the scoring step reads the high diagonal byte (score_chunk); the overflow path's
                            // kernels take the candidates as they are, so their diagonals are completed here
                            c.score = SEGS ? 0u : where;
                            c.diag = (uint16_t)(SEGS ? (d8 | ((uint32_t)split_hi_q[where] << 8)) : d8);
                            c.pad = (uint16_t)cur_seg;
                            const uint32_t ck = ncand + (uint32_t)__popcll(kb & below);
                            *(ck < (uint32_t)PF_CAND0 ? cand_small + ck : cand_big + ck) = c;
                            if (!SEGS && ck < 64) {
                                M.cand[wave][0][ck] = key | (d8 << 12);
                                M.cand[wave][1][ck] = arr;
                                M.cand[wave][2][ck] = where;
                            }
                        }
                        ncand += (uint32_t)__popcll(kb);
                    }
                }
                if (!SEGS) break;
                todo &= ~ballot(now);
                if (todo) {   // the remaining entries belong to the next segment: fresh state
                    clear_state();
                    cur_seg++;
                    next_boundary = cur_seg + 1 <= nseg ? segs[cur_seg + 1] : 0xFFFFFFFFu;
                }
            }
        };
        static_assert(REPLAY_PD == 4, "four named slots below");
        uint32_t e0, e1, e2, e3, tile0, tile1, tile2, tile3, at0, at1, at2, at3;
        request(0u, e0, tile0, at0);
        request(64u, e1, tile1, at1);
        request(128u, e2, tile2, at2);
        request(192u, e3, tile3, at3);
        for (uint32_t xg = 0; xg < total; xg += 64u * REPLAY_PD) {
            one_round(xg, e0, tile0, at0);
            one_round(xg + 64u, e1, tile1, at1);
            one_round(xg + 128u, e2, tile2, at2);
            one_round(xg + 192u, e3, tile3, at3);
        }
    }
    if (lane == 0) {
        A.cand_count[bucket] = ncand;
        // the work list of the scoring / keepMax kernels of larger buckets (a few thousand of a batch's millions)
        if (!SEGS && ncand > 64 && A.big_list) A.big_list[atomicAdd(A.big_count, 1u)] = (uint32_t)(bucket - (uint64_t)A.q_first * B);
    }
    // a8 + keepMaxElement for the common case of at most 64 candidates, straight from LDS (no second kernel's
    // count -> record -> metadata round trips); larger buckets are left to pf_ungapped_kernel / pf_keepmax_kernel
    if (!SEGS && ncand > 0 && ncand <= 64) {
        PfCand c;
        c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
        if ((uint32_t)lane < ncand) {
            const uint32_t kd = M.cand[wave][0][lane];
            c.id = ((kd & 0xFFFu) << bshift) | bin;
            c.arr = M.cand[wave][1][lane];
            c.diag = (uint16_t)(kd >> 12);
            c.score = M.cand[wave][2][lane];
        }
        uint64_t cells = score_chunk(A, M.smat, bucket, q, 0, ncand, ncand, c, bshift);
        if (A.cell_counter) {
            for (int dd = 1; dd < 64; dd <<= 1) cells += __shfl_xor((unsigned long long)cells, dd);
            if (lane == 0 && cells) atomicAdd((unsigned long long *)&A.cell_counter[q], (unsigned long long)cells);
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(REPLAY_WAVES_PER_EU))) void pf_replay_kernel(PfDedupArgs A) {
    __shared__ ReplayLds M;
    const int wave = (int)(threadIdx.x >> 6);
    for (int k = (int)threadIdx.x; k < 32 * 32; k += 256)      // rows of 32 (seg_cells_n)
        M.smat[k] = ((k >> 5) < A.alphabet && (k & 31) < A.alphabet) ? A.mat[(k >> 5) * A.alphabet + (k & 31)] : (int8_t)0;
    __syncthreads();
    // (the buckets of one query read the same tiles; a contiguous run of queries per XCD measured slower than the round-robin
    // deal - an eighth of the batch's queries is an uneven share of the work: profiles/r05_exp_pf_order.txt)
    const uint64_t bucket = (uint64_t)A.q_first * A.bins + (uint64_t)blockIdx.x * 4u + (uint32_t)wave;
    if (bucket >= (uint64_t)(A.q_first + A.n_queries) * A.bins) return;
    if (A.q_nseg && A.q_nseg[(uint32_t)(bucket / A.bins)]) replay_bucket_impl<true>(A, M, bucket);      // wave-uniform
    else replay_bucket_impl<false>(A, M, bucket);
}

// Buckets with more than 64 candidates (the replay kernel scores the others itself): one wavefront per (query, bin).
__device__ __forceinline__ void ungapped_bucket(const PfDedupArgs &A, const int8_t *smat, uint64_t bucket) {
    const int lane = lane_id();
    const uint32_t B = A.bins;
    const uint32_t ncand = A.cand_count[bucket];
    const uint32_t q = (uint32_t)(bucket / B);
    if (ncand <= 64 || (A.q_nseg && A.q_nseg[q])) return;   // small bins are done; overflow queries have their own path
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    uint64_t cells = 0;
    for (uint32_t cb0 = 0; cb0 < ncand; cb0 += 64) {
        const uint32_t nin = min(64u, ncand - cb0);
        PfCand c;
        c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
        if ((uint32_t)lane < nin) c = *cand_slot(A, bucket, cb0 + (uint32_t)lane);
        cells += score_chunk(A, smat, bucket, q, cb0, nin, ncand, c, bshift);
    }
    if (A.cell_counter) {   // statistics: one counter per query (a single global counter serialises 2.6 M atomics)
        for (int dd = 1; dd < 64; dd <<= 1) cells += __shfl_xor((unsigned long long)cells, dd);
        if (lane == 0 && cells) atomicAdd((unsigned long long *)&A.cell_counter[q], (unsigned long long)cells);
    }
}

// Round 6: with a work list (A.big_list, written by the replay kernel) the grid is a fixed number of workgroups whose wavefronts
// walk the list - a grid over every bucket of the chunk is hundreds of thousands of workgroups that find nothing to do.
__global__ __launch_bounds__(256) void pf_ungapped_kernel(PfDedupArgs A) {
    __shared__ int8_t smat[32 * 32];
    const int wave = (int)(threadIdx.x >> 6);
    for (int k = (int)threadIdx.x; k < 32 * 32; k += 256)      // rows of 32 (seg_cells_n)
        smat[k] = ((k >> 5) < A.alphabet && (k & 31) < A.alphabet) ? A.mat[(k >> 5) * A.alphabet + (k & 31)] : (int8_t)0;
    __syncthreads();
    const uint32_t B = A.bins;
    if (A.big_list) {
        const uint32_t n = *A.big_count;
        for (uint32_t i = blockIdx.x * 4u + (uint32_t)wave; i < n; i += gridDim.x * 4u) ungapped_bucket(A, smat, (uint64_t)A.q_first * B + A.big_list[i]);
        return;
    }
    const uint64_t bucket = (uint64_t)A.q_first * B + (uint64_t)blockIdx.x * 4u + (uint32_t)wave;
    if (bucket >= (uint64_t)(A.q_first + A.n_queries) * B) return;
    ungapped_bucket(A, smat, bucket);
}

// ---------------------------------------------------------------------------------------------------------
// --diag-score 0 (mmgpu_pf_params::kmer_score): findDuplicates with computeTotalScore (CacheFriendlyOperations.cpp:
// 194-239).  The first pass is the same as above (an entry is flagged when its diagonal byte equals the previous entry's of
// the same target); then the flagged entries are COUNTED per target (saturating at 255) and the target's element is its
// first flagged entry (id, that entry's diagonal) with count = that number.  One wavefront per (query, bin); state per
// target in LDS: previous diagonal byte | count << 8, plus the "has an element" bit.  Elements are listed in the order
// of their first flagged entry (the reference's output order inside a bin); the counts are attached at the end, elements
// with count >= min_diag_score go to the query's survivor list (the cut of QueryMatcher.cpp:216-219 is never lower).
__global__ __launch_bounds__(256) void pf_count_kernel(PfDedupArgs A) {
    __shared__ uint16_t s_state[4][PF_IDS_PER_BIN];
    __shared__ uint32_t s_emit[4][PF_IDS_PER_BIN / 32];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t B = A.bins;
    const uint64_t bucket = (uint64_t)A.q_first * B + (uint64_t)blockIdx.x * 4u + (uint32_t)wave;
    if (bucket >= (uint64_t)(A.q_first + A.n_queries) * B) return;
    const uint32_t q = (uint32_t)(bucket / B), bin = (uint32_t)(bucket % B);
    const uint32_t ntiles = A.q_ntiles[q];
    if (lane == 0) A.cand_count[bucket] = 0;      // nothing for the scoring kernels of the other mode
    if (ntiles == 0) return;
    const uint32_t tb = A.q_tile_base[q];
    uint16_t *S = s_state[wave];
    uint32_t *E = s_emit[wave];
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    for (int k = lane; k < PF_IDS_PER_BIN; k += 64) S[k] = 0;
    for (int k = lane; k < PF_IDS_PER_BIN / 32; k += 64) E[k] = 0;
    uint32_t ncand = 0;
    const uint64_t below = lanes_below(lane);
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 64) {
        const uint32_t tl = t0 + (uint32_t)lane;
        uint32_t o0 = 0, n = 0;
        if (tl < ntiles) {
            const uint16_t *bo = A.bin_off + (size_t)(tb + tl) * (B + 1) + bin;
            o0 = bo[0];
            n = (uint32_t)bo[1] - o0;
        }
        const uint32_t incl = wave_incl_scan(n);
        const uint32_t total = __shfl(incl, 63);
        const uint32_t excl = incl - n;
        for (uint32_t x0 = 0; x0 < total; x0 += 64) {
            const uint32_t x = x0 + (uint32_t)lane;
            const bool now = x < total;
            const int m = seg_find(excl, now ? x : 0u);
            const uint32_t ex_m = __shfl(excl, m), o_m = __shfl(o0, m);
            const uint32_t tile = t0 + (uint32_t)m;
            uint32_t e = 0;
            const size_t where = (size_t)(tb + tile) * PF_T + o_m + (x - ex_m);
            if (now) e = A.split[where];
            const uint32_t key = e & 0xFFFu;   // < PF_IDS_PER_BIN
            const uint32_t id = (key << bshift) | bin;
            const uint32_t d8 = (e >> 12) & 0xFFu;
            const uint32_t arr = tile * (uint32_t)PF_T + (e >> 20);
            const uint64_t same = match_lanes(key, 12, now);
            uint32_t st = 0, em = 0;
            if (now) {
                st = S[key];
                em = (E[key >> 5] >> (key & 31u)) & 1u;
            }
            // is my diagonal byte the previous entry's of this target?  (:194-208; the table starts at zero)
            const uint64_t pm = same & below;
            const int pl = pm ? highest_lane(pm) : lane;
            const uint32_t d_pl = __shfl(d8, pl);
            const uint32_t prevd = pm ? d_pl : (st & 0xFFu);
            const bool flag = now && d8 == prevd;
            const uint64_t fl = ballot(flag);
            const uint64_t fm = same & fl;
            // the last lane of a target's group writes the state: its byte, the count so far + the group's flagged entries
            if (now && (same & ~below & ~(1ull << lane)) == 0) {
                const uint32_t cnt = min(255u, (st >> 8) + (uint32_t)__popcll(fm));
                S[key] = (uint16_t)(d8 | (cnt << 8));
            }
            // the target's element: its first flagged entry
            const bool keep = flag && (fm & below) == 0 && em == 0u;
            if (keep) atomicOr(&E[key >> 5], 1u << (key & 31u));
            const uint64_t kb = ballot(keep);
            if (keep) {
                PfCand c;
                c.id = id;
                c.arr = arr;
                c.score = 0;
                c.diag = (uint16_t)(d8 | ((uint32_t)A.split_hi[where] << 8));
                c.pad = 0;
                *cand_slot(A, bucket, ncand + (uint32_t)__popcll(kb & below)) = c;
            }
            ncand += (uint32_t)__popcll(kb);
        }
    }
    if (ncand == 0) return;
    __threadfence();      // the elements were written by other lanes of this wavefront
    if (lane == 0 && A.q_ncand) {
        const uint32_t before = atomicAdd(&A.q_ncand[q], ncand);
        // resultSize >= foundDiagonalsSize / 2: the reference sorts with an unstable std::sort (QueryMatcher.cpp:221-231)
        if (before + ncand >= A.sort_cap && A.q_flags) atomicOr(&A.q_flags[q], 2u);
    }
    PfCand *surv = A.surv + (A.cand_base[(uint64_t)q * B] - A.cand_origin);
    unsigned long long total_count = 0;
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        PfCand c;
        c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
        bool win = false;
        if (ci < ncand) {
            c = *cand_slot(A, bucket, ci);
            c.score = (uint32_t)S[c.id >> bshift] >> 8;
            total_count += c.score;
            win = c.score >= A.min_diag_score;
        }
        const uint64_t wb = ballot(win);
        if (wb) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.surv_count[q], (uint32_t)__popcll(wb));
            base = __shfl(base, 0);
            if (win) surv[base + (uint32_t)__popcll(wb & below)] = c;
        }
    }
    if (A.cell_counter) {   // statistics_t::doubleMatches = sum of the counts (QueryMatcher.cpp:366-385)
        for (int dd = 1; dd < 64; dd <<= 1) total_count += __shfl_xor(total_count, dd);
        if (lane == 0 && total_count) atomicAdd((unsigned long long *)&A.cell_counter[q], total_count);
    }
}

// ---------------------------------------------------------------------------------------------------------
// keepMaxElement (CacheFriendlyOperations.cpp:354-384): per target keep the first candidate (bin order = arrival
// order) whose count = min(255, score) is the target's maximum; survivors with count >= min_diag_score are appended
// to the query's list.  One wavefront per (query, bin); LDS table of (count << 24 | ~candidate index) per target.
__device__ __forceinline__ void keepmax_bucket(const PfDedupArgs &A, uint32_t *S, uint64_t bucket) {
    const int lane = lane_id();
    const uint32_t B = A.bins;
    const uint32_t ncand = A.cand_count[bucket];
    const uint32_t q = (uint32_t)(bucket / B);
    if (ncand <= 64 || (A.q_nseg && A.q_nseg[q])) return;   // scored and reduced already / overflow path
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    for (int k = lane; k < PF_IDS_PER_BIN; k += 64) S[k] = 0;
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        if (ci < ncand) {
            const PfCand *cp = cand_slot(A, bucket, ci);
            const uint32_t cnt = min(255u, cp->score);
            const uint32_t k2 = (cnt << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu));
            atomicMax(&S[cp->id >> bshift], k2);
        }
    }
    PfCand *surv = A.surv + (A.cand_base[(uint64_t)q * B] - A.cand_origin);
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        bool win = false;
        PfCand c;
        c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
        if (ci < ncand) {
            c = *cand_slot(A, bucket, ci);
            const uint32_t cnt = min(255u, c.score);
            const uint32_t k2 = (cnt << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu));
            win = S[c.id >> bshift] == k2 && cnt >= A.min_diag_score;
        }
        const uint64_t wb = ballot(win);
        if (wb) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.surv_count[q], (uint32_t)__popcll(wb));
            base = __shfl(base, 0);
            if (win) surv[base + (uint32_t)__popcll(wb & lanes_below(lane))] = c;
        }
    }
}

__global__ __launch_bounds__(256) void pf_keepmax_kernel(PfDedupArgs A) {
    __shared__ uint32_t s_tab[4][PF_IDS_PER_BIN];
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t B = A.bins;
    if (A.nucl) return;                                      // pf_keepmax_nucl_kernel
    const uint64_t bucket = (uint64_t)A.q_first * B + (uint64_t)blockIdx.x * 4u + (uint32_t)wave;
    if (bucket >= (uint64_t)(A.q_first + A.n_queries) * B) return;
    keepmax_bucket(A, s_tab[wave], bucket);
}

// The same over the replay kernel's work list (see pf_ungapped_kernel): a WORKGROUP of 16 wavefronts per listed bucket.  The list is
// short (hundreds of buckets per stage chunk of the headline batch) and its time was that of its largest buckets - repeats that put
// thousands of candidates into one bin -, which one wavefront walked 64 candidates at a time with an atomic round trip to the
// query's survivor count per step (1.4 ms for a chunk's 625 buckets).
__global__ __launch_bounds__(1024) void pf_keepmax_list_kernel(PfDedupArgs A) {
    __shared__ uint32_t S[PF_IDS_PER_BIN];
    const int lane = lane_id();
    const uint32_t B = A.bins;
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    const uint32_t n = *A.big_count;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint64_t bucket = (uint64_t)A.q_first * B + A.big_list[i];
        const uint32_t ncand = A.cand_count[bucket];
        const uint32_t q = (uint32_t)(bucket / B);
        for (uint32_t k = threadIdx.x; k < (uint32_t)PF_IDS_PER_BIN; k += 1024) S[k] = 0;
        __syncthreads();
        for (uint32_t ci = threadIdx.x; ci < ncand; ci += 1024) {
            const PfCand *cp = cand_slot(A, bucket, ci);
            atomicMax(&S[cp->id >> bshift], (min(255u, cp->score) << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu)));
        }
        __syncthreads();
        PfCand *surv = A.surv + (A.cand_base[(uint64_t)q * B] - A.cand_origin);
        for (uint32_t c0 = 0; c0 < ncand; c0 += 1024) {      // (every wavefront takes part in its ballot)
            const uint32_t ci = c0 + threadIdx.x;
            bool win = false;
            PfCand c;
            c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
            if (ci < ncand) {
                c = *cand_slot(A, bucket, ci);
                const uint32_t cnt = min(255u, c.score);
                win = S[c.id >> bshift] == ((cnt << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu))) && cnt >= A.min_diag_score;
            }
            const uint64_t wb = ballot(win);
            if (wb) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&A.surv_count[q], (uint32_t)__popcll(wb));
                base = __shfl(base, 0);
                if (win) surv[base + (uint32_t)__popcll(wb & lanes_below(lane))] = c;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Targets of 32768 residues or more (UngappedAlignment.cpp:187-312).  The 16-bit diagonal of a candidate does not say where in
// such a target the match lies: the reference scores every 65536-shift of it (computeLongScore :295-312) - and which TARGET it
// scores depends on the batches of eight elements of one 16-bit diagonal that computeScores (:315-346) forms over the query's
// result array: in a batch that is not full a long target gets its own long score; in a FULL batch the long targets enter the
// length sort with length 0 (first, in their order) and the write-back loop gives the h-th of them the long score of the batch's
// h-th element IN ARRAY ORDER when that one's target is long, and 0 otherwise (:262-275 reads hits[hitIdx], not hits[seqs[hitIdx].id];
// oracle/prefilter_oracle.c mmo_score_batch restates it and is pinned against the reference).  The replay / scoring kernels give such
// candidates a meaningless score and set bit 0 of the query's flags; this kernel, one workgroup per flagged query, after them:
//   1. marks the 16-bit diagonals of the query's long-target candidates in an LDS bit set;
//   2. pools ALL candidates of the query on those diagonals (key: diagonal, the reference's cache bin, arrival index = the order of
//      the reference's result array), sorts the pool: a diagonal's run is its elements in array order, batches are its eighths;
//   3. decides for every long-target candidate whose target scores it, scores (one wavefront per candidate, every shift), writes
//      the count and - for the rescoring of saturated elements, which reads the element's own target (scoreSingleSequence :453-461) -
//      the own exact score into the candidate's slot (pf_el_count / pf_el_exact);
//   4. redoes keepMaxElement for every bucket of the query and clears the flag.
// Queries it leaves flagged (handed to the host as before): overflow-path and nucleotide queries, more than PF_LONG_POOL (4096) pooled candidates.
constexpr int PF_LONG_POOL = 4096;     // (89 KB of LDS with the tables below: a workgroup of this kernel has a CU's LDS to itself if need be)
constexpr uint32_t PF_LONG_NONE = 0xFFFFu;

// ungapped score of one diagonal (computeSingelSequenceScores :423-437) by a whole wavefront: passes of 64 x 16 cells
__device__ int wave_diag_score(const int8_t *smat, const uint8_t *qr, const uint8_t *qc, const int8_t *prows, int qlen, const uint8_t *t,
                               int tlen, int diag) {
    const int lane = lane_id();
    const int mind = diag < 0 ? -diag : diag;
    int len, qs, ts;
    if (diag >= 0 && mind < qlen) {
        len = min(tlen, qlen - mind);
        qs = mind;
        ts = 0;
    } else if (diag < 0 && mind < tlen) {
        len = min(tlen - mind, qlen);
        qs = 0;
        ts = mind;
    } else {
        return 0;
    }
    int sc = 0, best = 0;
    for (int p0 = 0; p0 < len; p0 += 1024) {
        const int o = p0 + lane * 16;
        Seg h;
        h.a = 0; h.b = 0; h.P = -(1 << 28); h.M = 0;
        if (o < len) {
            uint32_t tw[4], qw[4], cw[4];
            load_cells<4>(t + ts + o, tw);
            if (prows) h = seg_cells_rows_n<4>(tw, prows + (size_t)(qs + o) * PF_PROW, min(16, len - o));
            else {
                load_cells<4>(qr + qs + o, qw);
                load_cells<4>(qc + qs + o, cw);
                h = seg_cells_n<4>(tw, qw, cw, min(16, len - o), smat);
            }
        }
        h = seg_tree16(h, lane & 15);      // lanes 0, 16, 32, 48 hold their group's summary
        Seg g[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            g[k].a = __shfl(h.a, 16 * k);
            g[k].b = __shfl(h.b, 16 * k);
            g[k].P = __shfl(h.P, 16 * k);
            g[k].M = __shfl(h.M, 16 * k);
        }
        const Seg all = seg_combine(seg_combine(g[0], g[1]), seg_combine(g[2], g[3]));
        best = max(best, max(all.M, sc + all.P));
        sc = max(all.a, sc + all.b);
    }
    return best;
}

// computeLongScore (:295-312): the shifts are computed in unsigned int and read back as int
__device__ int wave_long_score(const int8_t *smat, const uint8_t *qr, const uint8_t *qc, const int8_t *prows, int qlen, const uint8_t *t,
                               int tlen, uint32_t diag16) {
    int total = 0;
    for (uint32_t d = 1; d <= 1u + (uint32_t)tlen / 32768u; d++)
        total = max(total, wave_diag_score(smat, qr, qc, prows, qlen, t, tlen, (int)(0u - d * 65536u + diag16)));
    for (uint32_t d = 0; d <= (uint32_t)qlen / 65536u; d++)
        total = max(total, wave_diag_score(smat, qr, qc, prows, qlen, t, tlen, (int)(d * 65536u + diag16)));
    return total;
}

// Queries of 32768 residues or more (:199-208): computeLongScore for every element, whatever its batch.  One wavefront per (query, bin)
// like the scoring kernels; the query's flag is raised so that pf_long_kernel redoes its keepMaxElement (overflow-path queries stay
// flagged and go to the host).
__global__ __launch_bounds__(256) void pf_longq_kernel(PfDedupArgs A) {
    __shared__ int8_t smat[32 * 32];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    for (int k = (int)threadIdx.x; k < 32 * 32; k += 256)
        smat[k] = ((k >> 5) < A.alphabet && (k & 31) < A.alphabet) ? A.mat[(k >> 5) * A.alphabet + (k & 31)] : (int8_t)0;
    __syncthreads();
    const uint32_t B = A.bins;
    const uint64_t bucket = (uint64_t)A.q_first * B + (uint64_t)blockIdx.x * 4u + (uint32_t)wave;
    if (bucket >= (uint64_t)(A.q_first + A.n_queries) * B) return;
    const uint32_t q = (uint32_t)(bucket / B);
    const uint32_t qp0 = A.q_off[q];
    const int qlen = (int)(A.q_off[q + 1] - qp0);
    if (qlen < 32768) return;
    if (lane == 0 && bucket == (uint64_t)q * B) atomicOr(&A.q_flags[q], 1u);
    if (A.nucl || (A.q_nseg && A.q_nseg[q])) return;
    const uint8_t *qr = A.q_res + qp0;
    const uint8_t *qc = reinterpret_cast<const uint8_t *>(A.q_corr) + qp0;
    const int8_t *prows = (A.q_isprof && A.q_isprof[q]) ? A.q_rows + (size_t)qp0 * PF_PROW : nullptr;
    const uint32_t ncand = A.cand_count[bucket];
    for (uint32_t ci = 0; ci < ncand; ci++) {
        PfCand *slot = cand_slot(A, bucket, ci);
        const uint32_t id = slot->id;
        const int own = wave_long_score(smat, qr, qc, prows, qlen, A.t_res + (size_t)A.t_off4[id] * 4, (int)A.t_len[id], slot->diag);
        if (lane == 0) slot->score = (uint32_t)own;
    }
}

__global__ __launch_bounds__(256) void pf_long_kernel(PfDedupArgs A) {
    __shared__ uint32_t s_tab[PF_IDS_PER_BIN];
    __shared__ uint32_t s_dbits[65536 / 32];
    __shared__ uint64_t s_key[PF_LONG_POOL];
    __shared__ uint32_t s_ci[PF_LONG_POOL];        // candidate index inside its bucket | long target << 31
    __shared__ uint16_t s_bucket[PF_LONG_POOL];
    __shared__ uint16_t s_job[PF_LONG_POOL];       // long-target entries: pool position of the element whose target scores it
    __shared__ int8_t smat[32 * 32];
    __shared__ uint32_t sh_npool, sh_nlong;
    const uint32_t q = A.q_first + blockIdx.x;
    if (!(A.q_flags[q] & 1u)) return;
    if (A.nucl || (A.q_nseg && A.q_nseg[q])) return;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t B = A.bins;
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    for (int k = (int)threadIdx.x; k < 32 * 32; k += 256)
        smat[k] = ((k >> 5) < A.alphabet && (k & 31) < A.alphabet) ? A.mat[(k >> 5) * A.alphabet + (k & 31)] : (int8_t)0;
    for (int k = (int)threadIdx.x; k < 65536 / 32; k += 256) s_dbits[k] = 0;
    if (threadIdx.x == 0) {
        sh_npool = 0;
        sh_nlong = 0;
    }
    __syncthreads();
    const uint32_t qp0 = A.q_off[q];
    const int qlen = (int)(A.q_off[q + 1] - qp0);
    // (a query of 32768 residues or more: pf_longq_kernel has given every candidate its own long score - no batches, :199-208)
    if (qlen < 32768) {
    // 1. diagonals of the long-target candidates
    for (uint32_t bk = (uint32_t)wave; bk < B; bk += 4) {
        const uint64_t bucket = (uint64_t)q * B + bk;
        const uint32_t ncand = A.cand_count[bucket];
        for (uint32_t ci = (uint32_t)lane; ci < ncand; ci += 64) {
            const PfCand *cp = cand_slot(A, bucket, ci);
            if (A.t_len[cp->id] >= 32768u) atomicOr(&s_dbits[cp->diag >> 5], 1u << (cp->diag & 31u));
        }
    }
    __syncthreads();
    // 2. every candidate of the query on one of those diagonals
    const uint32_t refmask = A.ref_bins - 1;
    for (uint32_t bk = (uint32_t)wave; bk < B; bk += 4) {
        const uint64_t bucket = (uint64_t)q * B + bk;
        const uint32_t ncand = A.cand_count[bucket];
        for (uint32_t ci = (uint32_t)lane; ci < ncand; ci += 64) {
            const PfCand *cp = cand_slot(A, bucket, ci);
            const uint32_t d = cp->diag;
            if (s_dbits[d >> 5] & (1u << (d & 31u))) {
                const uint32_t at = atomicAdd(&sh_npool, 1u);
                if (at < (uint32_t)PF_LONG_POOL) {
                    const uint32_t is_long = A.t_len[cp->id] >= 32768u ? 1u : 0u;
                    s_key[at] = ((uint64_t)d << 43) | ((uint64_t)(cp->id & refmask) << 32) | (uint64_t)cp->arr;
                    s_ci[at] = ci | (is_long << 31);
                    s_bucket[at] = (uint16_t)bk;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t npool = sh_npool;
    if (npool > (uint32_t)PF_LONG_POOL) return;      // stays flagged: the host runs the query
    uint32_t np2 = 1;
    while (np2 < npool) np2 <<= 1;
    for (uint32_t k = npool + threadIdx.x; k < np2; k += 256) {
        s_key[k] = ~0ull;
        s_ci[k] = 0;
        s_bucket[k] = 0;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < np2 / 2; k += 256) {
                const uint32_t i = 2 * k - (k & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = s_key[i], b = s_key[j];
                if ((a > b) == up) {
                    s_key[i] = b;
                    s_key[j] = a;
                    const uint32_t tc = s_ci[i];
                    s_ci[i] = s_ci[j];
                    s_ci[j] = tc;
                    const uint16_t tb = s_bucket[i];
                    s_bucket[i] = s_bucket[j];
                    s_bucket[j] = tb;
                }
            }
            __syncthreads();
        }
    }
    // 3a. whose target scores a long-target element: its own (batch not full), the batch's h-th element's (full batch, when that
    //     one's target is long), nobody's (score 0)
    for (uint32_t p = threadIdx.x; p < npool; p += 256) {
        uint16_t job = (uint16_t)PF_LONG_NONE;
        if (s_ci[p] >> 31) {
            const uint64_t d = s_key[p] >> 43;
            uint32_t lo = p, hi = p;
            while (lo > 0 && (s_key[lo - 1] >> 43) == d) lo--;
            while (hi + 1 < npool && (s_key[hi + 1] >> 43) == d) hi++;
            const uint32_t r = p - lo, n_d = hi - lo + 1, b0 = (r / 8u) * 8u;
            if (b0 + 8u > n_d) job = (uint16_t)p;
            else {
                uint32_t h = 0;
                for (uint32_t z = lo + b0; z < p; z++) h += s_ci[z] >> 31;
                const uint32_t e = lo + b0 + h;
                job = (s_ci[e] >> 31) ? (uint16_t)e : (uint16_t)PF_LONG_NONE;
            }
            atomicAdd(&sh_nlong, 1u);
        }
        s_job[p] = job;
    }
    __syncthreads();
    // 3b. the scores: one wavefront per long-target element
    const uint8_t *qr = A.q_res + qp0;
    const uint8_t *qc = reinterpret_cast<const uint8_t *>(A.q_corr) + qp0;
    const int8_t *prows = (A.q_isprof && A.q_isprof[q]) ? A.q_rows + (size_t)qp0 * PF_PROW : nullptr;
    for (uint32_t p = (uint32_t)wave; p < npool; p += 4) {
        if (!(s_ci[p] >> 31)) continue;                  // wave-uniform
        PfCand *slot = cand_slot(A, (uint64_t)q * B + s_bucket[p], s_ci[p] & 0x7FFFFFFFu);
        const uint32_t id = slot->id, d16 = slot->diag;
        const int own = wave_long_score(smat, qr, qc, prows, qlen, A.t_res + (size_t)A.t_off4[id] * 4, (int)A.t_len[id], d16);
        int by = 0;
        const uint32_t job = s_job[p];
        if (job == p) by = own;
        else if (job != PF_LONG_NONE) {
            const uint32_t id2 = cand_slot(A, (uint64_t)q * B + s_bucket[job], s_ci[job] & 0x7FFFFFFFu)->id;
            by = wave_long_score(smat, qr, qc, prows, qlen, A.t_res + (size_t)A.t_off4[id2] * 4, (int)A.t_len[id2], d16);
        }
        const uint32_t cnt = (uint32_t)min(255, by), ex = (uint32_t)own;
        if (lane == 0) slot->score = (cnt == min(255u, ex)) ? ex : (0x80000000u | (cnt << 23) | (ex & 0x7FFFFFu));
    }
    }
    __threadfence();
    __syncthreads();
    // 4. keepMaxElement (CacheFriendlyOperations.cpp:354-384) over every bucket of the query, as pf_keepmax_kernel does it
    if (threadIdx.x == 0) A.surv_count[q] = 0;
    __syncthreads();
    PfCand *surv = A.surv + (A.cand_base[(uint64_t)q * B] - A.cand_origin);
    for (uint32_t bk = 0; bk < B; bk++) {
        const uint64_t bucket = (uint64_t)q * B + bk;
        const uint32_t ncand = A.cand_count[bucket];
        if (ncand == 0) continue;                        // workgroup-uniform
        for (uint32_t ci = threadIdx.x; ci < ncand; ci += 256) s_tab[cand_slot(A, bucket, ci)->id >> bshift] = 0;
        __syncthreads();
        for (uint32_t ci = threadIdx.x; ci < ncand; ci += 256) {
            const PfCand *cp = cand_slot(A, bucket, ci);
            atomicMax(&s_tab[cp->id >> bshift], (pf_el_count(cp->score) << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu)));
        }
        __syncthreads();
        for (uint32_t ci = threadIdx.x; ci < ncand; ci += 256) {
            const PfCand c = *cand_slot(A, bucket, ci);
            const uint32_t cnt = pf_el_count(c.score);
            if (s_tab[c.id >> bshift] == ((cnt << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu))) && cnt >= A.min_diag_score)
                surv[atomicAdd(&A.surv_count[q], 1u)] = c;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) A.q_flags[q] &= ~1u;
}

// ---------------------------------------------------------------------------------------------------------
// Nucleotide searches (QueryMatcher.cpp:147-177): before keepMaxScoreElementOnly the reference brings the SATURATED elements
// (count == 255) of every target together and writes the diagonal with the best exact score
// (scoreSingleSequenceCombined) into the first of them; the others lose the keepMax that follows.  Per target: a saturated
// element with the best exact score wins; two different diagonals with the same best score: the first in the order the
// reference's std::sort by id left.  libstdc++'s std::sort of at most 16 elements is an insertion sort, i.e. stable, and the
// order before it is the arrival order for one target - the earliest candidate holding the best score (F below) is the
// reference's choice as well.  With more than 16 saturated elements in the QUERY the sort partitions and the order of equal
// ids is a property of introsort: bit 2 of q_flags records such a tie, the query's saturated elements are exported (A.sat,
// A.q_nsat) and mmgpu_pf_fetch runs the same std::sort over them on the host and corrects the diagonal of the target's hit.
// Without saturated elements the ordinary rule (highest count, earliest arrival).  One wavefront per (query, bin), every bucket (the replay kernel leaves keepMax alone in this
// mode); two LDS tables: best key per target, and the first candidate holding it.
__global__ __launch_bounds__(128) void pf_keepmax_nucl_kernel(PfDedupArgs A) {
    __shared__ uint32_t s_key[2][PF_IDS_PER_BIN];
    __shared__ uint32_t s_first[2][PF_IDS_PER_BIN];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t B = A.bins;
    const uint64_t bucket = (uint64_t)A.q_first * B + (uint64_t)blockIdx.x * 2u + (uint32_t)wave;
    if (bucket >= (uint64_t)(A.q_first + A.n_queries) * B) return;
    const uint32_t ncand = A.cand_count[bucket];
    const uint32_t q = (uint32_t)(bucket / B);
    if (ncand == 0) return;
    if (lane == 0) {
        const uint32_t before = atomicAdd(&A.q_ncand[q], ncand);
        if (before + ncand >= A.sort_cap && A.q_flags) atomicOr(&A.q_flags[q], 2u);    // canBeSorted is false (:146): not restated
    }
    if (A.q_nseg && A.q_nseg[q]) {      // overflow path of a nucleotide query: left to the host
        if (lane == 0 && A.q_flags) atomicOr(&A.q_flags[q], 2u);
        return;
    }
    uint32_t *K = s_key[wave], *F = s_first[wave];
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    for (int k = lane; k < PF_IDS_PER_BIN; k += 64) { K[k] = 0; F[k] = 0xFFFFFFFFu; }
    auto key_of = [&](const PfCand &c, uint32_t ci) -> uint32_t {
        const uint32_t cnt = min(255u, c.score);
        return cnt >= 255u ? (0xFF000000u | min(c.score, 0xFFFFFFu)) : ((cnt << 24) | (0xFFFFFFu - min(ci, 0xFFFFFEu)));
    };
    // the saturated elements also go to the query's export list: the range the reference sorts is all of them (fetch)
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        bool sat = false;
        PfCand c;
        c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
        if (ci < ncand) {
            c = *cand_slot(A, bucket, ci);
            atomicMax(&K[c.id >> bshift], key_of(c, ci));
            sat = c.score >= 255u;
        }
        const uint64_t sb = ballot(sat);
        if (sb && A.q_nsat) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.q_nsat[q], (uint32_t)__popcll(sb));
            base = __shfl(base, 0);
            const uint32_t slot = base + (uint32_t)__popcll(sb & lanes_below(lane));
            if (sat && slot < A.sat_cap) A.sat[(size_t)q * A.sat_cap + slot] = c;
        }
    }
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        if (ci < ncand) {
            const PfCand c = *cand_slot(A, bucket, ci);
            if (K[c.id >> bshift] == key_of(c, ci)) atomicMin(&F[c.id >> bshift], ci);
        }
    }
    PfCand *surv = A.surv + (A.cand_base[(uint64_t)q * B] - A.cand_origin);
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        bool win = false;
        PfCand c;
        c.id = 0; c.arr = 0; c.score = 0; c.diag = 0; c.pad = 0;
        if (ci < ncand) {
            c = *cand_slot(A, bucket, ci);
            const uint32_t t = c.id >> bshift;
            const uint32_t cnt = min(255u, c.score);
            const bool top = K[t] == key_of(c, ci);
            win = top && F[t] == ci && cnt >= A.min_diag_score;
            if (top && F[t] != ci && cnt >= 255u) {      // another saturated element with the same exact score
                const PfCand w = *cand_slot(A, bucket, F[t]);
                if (w.diag != c.diag && A.q_flags) atomicOr(&A.q_flags[q], 4u);      // decided with the query's saturated total (fetch)
            }
        }
        const uint64_t wb = ballot(win);
        if (wb) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.surv_count[q], (uint32_t)__popcll(wb));
            base = __shfl(base, 0);
            if (win) surv[base + (uint32_t)__popcll(wb & lanes_below(lane))] = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Overflow path of QueryMatcher::match (QueryMatcher.cpp:310-346): a query whose index lists hold >= maxDbMatches
// entries is processed by the reference in SEGMENTS - whenever the next list would not fit, everything gathered so
// far goes through findDuplicates on its own, the buffer is emptied and gathering continues.
// pf_segments_kernel finds the segment boundaries (arrival indices) of such a query: one wavefront per query,
// 64-ary search over its list records for the first list with start + len >= segment start + cap.
__global__ __launch_bounds__(64) void pf_segments_kernel(PfSegArgs A) {
    const uint32_t q = A.ovf_queries[blockIdx.x];
    const int lane = lane_id();
    const uint32_t qp0 = A.q_off[q], qp1 = A.q_off[q + 1];
    const uint32_t l0 = A.list_base[qp0], l1 = A.list_base[qp1];   // this query's list records
    const uint32_t total = A.q_entries[q];
    uint32_t *seg = A.seg_start + (size_t)q * (PF_MAX_SEG + 2);
    uint32_t nseg = 0, seg_begin = 0, lcur = l0;
    if (lane == 0) seg[0] = 0;
    while (true) {
        const uint64_t limit = (uint64_t)seg_begin + A.cap;   // first list with start + len >= limit opens a new segment
        if ((uint64_t)total < limit) break;                   // (start + len <= total for every list)
        // lists [lcur, l1): value(l) = start(l) + len(l) is non-decreasing; find the first l with value >= limit
        uint32_t lo = lcur, hi = l1;                          // answer in [lo, hi]; hi == l1 means none
        while (lo < hi) {
            const uint32_t span = hi - lo;
            const uint32_t step = (span + 63u) / 64u;
            const uint32_t idx = lo + (uint32_t)lane * step;
            bool ge = true;                                   // beyond the range counts as "satisfies"
            if (idx < hi) {
                const PfList r = A.lists[idx];
                ge = (uint64_t)A.pos_entry_base[r.pos] + r.lprefix + r.len >= limit;
            }
            const uint64_t m = ballot(ge);                  // a suffix of the lanes
            const uint32_t first = m ? (uint32_t)(__ffsll((long long)m) - 1) : 64u;
            // the answer lies in (lo + (first-1)*step, lo + first*step]
            const uint32_t nhi = min(hi, lo + first * step);
            const uint32_t nlo = first == 0 ? lo : min(hi, lo + (first - 1u) * step + 1u);
            lo = nlo;
            hi = nhi;
            if (step == 1) { lo = hi = nhi; }
        }
        if (lo >= l1) break;
        const PfList r = A.lists[lo];
        const uint32_t start = A.pos_entry_base[r.pos] + r.lprefix;
        if (nseg >= (uint32_t)PF_MAX_SEG) { nseg = PF_MAX_SEG + 1; break; }
        nseg++;
        if (lane == 0) seg[nseg] = start;
        seg_begin = start;
        lcur = lo;
    }
    if (lane == 0) {
        A.q_nseg[q] = nseg;
        A.q_final[q] = total - seg_begin;
    }
}

// ---------------------------------------------------------------------------------------------------------
// One flush / the final merge of the overflow path for every (overflow query, bin): a single wavefront keeps the
// reference's foundDiagonals semantics for its bin.  All operations of the reference (mergeDiagonalKeepScoredHits-
// Duplicates, mergeDiagonalDuplicates, keepMaxElement, CacheFriendlyOperations.cpp:83-148,354-384) act per target on
// the elements in array order, and a target lives in exactly one bin, so a bin can be processed on its own; the
// position of an element inside the CPU's array (needed only to break score ties at the --max-seqs cut) is carried as
// a key `ord` for which appending = larger key and the reversal done by the keep-scored merge = negation.
__device__ __forceinline__ PfOvfElem ovf_from_cand(const PfCand &c, uint32_t step) {
    PfOvfElem e;
    e.id = c.id;
    e.score = 0;
    e.ord = ((long long)step << 33) + (long long)c.arr;   // |ord| < 2^39 for PF_MAX_SEG + 1 <= 63 steps
    e.diag = c.diag;
    e.pad0 = 0;
    e.pad1 = 0;
    return e;
}

__global__ __launch_bounds__(256) void pf_overflow_kernel(PfOvfArgs A) {
    __shared__ uint32_t s_tab[4][PF_IDS_PER_BIN];
    __shared__ int8_t smat[32 * 32];
    const PfDedupArgs &D = A.D;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    for (int k = (int)threadIdx.x; k < 32 * 32; k += 256)      // rows of 32 (seg_cells_n)
        smat[k] = ((k >> 5) < D.alphabet && (k & 31) < D.alphabet) ? D.mat[(k >> 5) * D.alphabet + (k & 31)] : (int8_t)0;
    __syncthreads();
    const uint32_t B = D.bins;
    const uint64_t w = (uint64_t)blockIdx.x * 4u + (uint32_t)wave;
    if (w >= (uint64_t)A.n_ovf * B) return;
    const uint32_t qi = (uint32_t)(w / B), bin = (uint32_t)(w % B);
    const uint32_t q = A.ovf_queries[qi];
    const uint32_t ns = D.q_nseg[q];
    const uint32_t step = A.step;
    if (ns > (uint32_t)PF_MAX_SEG || step > ns + 1) return;
    const bool final_step = step == ns + 1;
    if (final_step && A.q_final[q] == 0) return;      // numMatches == 0 after the last flush: hitCount stays 0 (:353-362)
    const uint64_t bucket = (uint64_t)q * B + bin;
    const uint32_t ncand = D.cand_count[bucket];
    uint32_t *tab = s_tab[wave];
    int bshift = 0;
    while ((1u << bshift) < B) bshift++;
    const uint64_t below = lanes_below(lane);
    const uint64_t base = A.ovf_base[qi] + (uint64_t)(D.cand_base[bucket] - D.cand_base[(uint64_t)q * B]);
    PfOvfElem *O = A.buf_a + base, *S = A.buf_b + base;
    uint32_t nO = A.o_count[(size_t)qi * B + bin];
    uint32_t *totals = A.totals + (size_t)qi * (PF_MAX_SEG + 2);

    // candidates of segment step-1 of this bin: a contiguous run of the bin's candidate list (tags are sorted)
    uint32_t c_lo = 0, c_hi = 0;
    for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t ci = c0 + (uint32_t)lane;
        uint32_t tag = 0xFFFFu;
        if (ci < ncand) tag = cand_slot(D, bucket, ci)->pad;
        c_lo += (uint32_t)__popcll(ballot(ci < ncand && tag < step - 1));
        c_hi += (uint32_t)__popcll(ballot(ci < ncand && tag <= step - 1));
    }
    const uint32_t nC = c_hi - c_lo;
    const uint32_t prev_total = step > 1 ? totals[step - 1] : 0u;
    // append the segment's candidates behind O (unscored)
    for (uint32_t c0 = 0; c0 < nC; c0 += 64) {
        const uint32_t k = c0 + (uint32_t)lane;
        if (k < nC) O[nO + k] = ovf_from_cand(*cand_slot(D, bucket, c_lo + k), step);
    }
    uint32_t n = nO + nC;
    __threadfence();
    const bool do_merge = final_step ? prev_total != 0 : (step > 1 && prev_total != 0);
    PfOvfElem *cur = O;
    if (do_merge && !final_step) {
        // mergeDiagonalKeepScoredHitsDuplicates: diag + 1 written forwards, then walked backwards
        for (uint32_t r0 = 0; r0 < n; r0 += 64) {
            const uint32_t idx = r0 + (uint32_t)lane;
            const bool act = idx < n;
            uint32_t id = 0, d8 = 0;
            if (act) { id = O[idx].id; d8 = O[idx].diag & 0xFFu; }
            const uint32_t key = id >> bshift;
            const uint64_t same = match_lanes(key, 12, act);
            if (act && (same & ~below & ~(1ull << lane)) == 0) tab[key] = (d8 + 1u) & 0xFFu;
        }
        uint32_t m = 0;
        for (uint32_t r0 = 0; r0 < n; r0 += 64) {
            const bool act = r0 + (uint32_t)lane < n;
            const uint32_t idx = act ? n - 1u - (r0 + (uint32_t)lane) : 0u;
            PfOvfElem e;
            e.id = 0; e.score = 0; e.ord = 0; e.diag = 0; e.pad0 = 0; e.pad1 = 0;
            if (act) e = O[idx];
            const uint32_t d8 = e.diag & 0xFFu, key = e.id >> bshift;
            const uint64_t same = match_lanes(key, 12, act);
            const uint64_t pm = same & below;
            const int pl = pm ? highest_lane(pm) : lane;
            const uint32_t d_pl = __shfl(d8, pl);
            uint32_t prev = 0;
            if (act) prev = pm ? d_pl : tab[key];
            const bool keep = act && (e.score != 0 || prev != d8);
            if (act && (same & ~below & ~(1ull << lane)) == 0) tab[key] = d8;
            const uint64_t kb = ballot(keep);
            if (keep) {
                e.ord = -e.ord;
                S[m + (uint32_t)__popcll(kb & below)] = e;
            }
            m += (uint32_t)__popcll(kb);
        }
        n = m;
        cur = S;
    } else if (do_merge && final_step) {
        // mergeDiagonalDuplicates: diag + 1 written backwards (the first element of a target wins), walked forwards
        for (uint32_t r0 = 0; r0 < n; r0 += 64) {
            const bool act = r0 + (uint32_t)lane < n;
            const uint32_t idx = act ? n - 1u - (r0 + (uint32_t)lane) : 0u;
            uint32_t id = 0, d8 = 0;
            if (act) { id = O[idx].id; d8 = O[idx].diag & 0xFFu; }
            const uint32_t key = id >> bshift;
            const uint64_t same = match_lanes(key, 12, act);
            if (act && (same & ~below & ~(1ull << lane)) == 0) tab[key] = (d8 + 1u) & 0xFFu;
        }
        uint32_t m = 0;
        for (uint32_t r0 = 0; r0 < n; r0 += 64) {
            const uint32_t idx = r0 + (uint32_t)lane;
            const bool act = idx < n;
            PfOvfElem e;
            e.id = 0; e.score = 0; e.ord = 0; e.diag = 0; e.pad0 = 0; e.pad1 = 0;
            if (act) e = O[idx];
            const uint32_t d8 = e.diag & 0xFFu, key = e.id >> bshift;
            const uint64_t same = match_lanes(key, 12, act);
            const uint64_t pm = same & below;
            const int pl = pm ? highest_lane(pm) : lane;
            const uint32_t d_pl = __shfl(d8, pl);
            uint32_t prev = 0;
            if (act) prev = pm ? d_pl : tab[key];
            const bool keep = act && prev != d8;
            if (act && (same & ~below & ~(1ull << lane)) == 0) tab[key] = d8;
            const uint64_t kb = ballot(keep);
            if (keep) S[m + (uint32_t)__popcll(kb & below)] = e;
            m += (uint32_t)__popcll(kb);
        }
        n = m;
        cur = S;
    }
    __threadfence();
    const bool do_score = final_step || (step > 1 && prev_total != 0);
    if (!do_score) {   // first flush (or nothing kept so far): the candidates simply become foundDiagonals (:329-331)
        if (lane == 0) {
            A.o_count[(size_t)qi * B + bin] = n;
            if (n) atomicAdd(&totals[step], n);
        }
        return;
    }
    // UngappedAlignment::align: elements without a score only (computeScores, UngappedAlignment.cpp:322-324).  Round 6: with the
    // scoring kernels' own scorer (score_chunk: a 16-lane group per element, dword loads, passes of 256 or 384 cells) - the queries on
    // this path are the LONGEST of a batch (their index entries exceed the reference's buffer), and a lane walking one element's
    // thousands of diagonal cells byte by byte made the final launch of nine such queries 1.27 ms long.  The unscored elements of a
    // round of 64 are brought together in the low lanes (the bin's table is free between the merge and keepMax: lane numbers and
    // scores pass through its first 128 words).
    for (uint32_t r0 = 0; r0 < n; r0 += 64) {
        const uint32_t idx = r0 + (uint32_t)lane;
        const bool need = idx < n && cur[idx].score == 0;
        const uint64_t nb = ballot(need);
        if (!nb) continue;                                   // wave-uniform
        const uint32_t cnt = (uint32_t)__popcll(nb), mine = (uint32_t)__popcll(nb & below);
        if (need) tab[mine] = (uint32_t)lane;
        __builtin_amdgcn_wave_barrier();
        const int src = (uint32_t)lane < cnt ? (int)tab[lane] : 0;
        const uint32_t my_id = idx < n ? cur[idx].id : 0u, my_diag = idx < n ? (uint32_t)cur[idx].diag : 0u;
        PfCand c;
        c.id = (uint32_t)__shfl((int)my_id, src);
        c.arr = 0;
        c.score = 0;
        c.diag = (uint16_t)__shfl((int)my_diag, src);
        c.pad = 0;
        uint32_t sc = 0;
        (void)score_chunk<false>(D, smat, bucket, q, 0, cnt, 65u, c, bshift, &sc);
        tab[64 + lane] = sc;
        __builtin_amdgcn_wave_barrier();
        if (need) cur[idx].score = tab[64 + mine];
    }
    __threadfence();
    // keepMaxElement: per target the first element holding the maximum count, plus every zero-count element after it
    for (int k = lane; k < PF_IDS_PER_BIN; k += 64) tab[k] = 0;
    for (uint32_t r0 = 0; r0 < n; r0 += 64) {
        const uint32_t idx = r0 + (uint32_t)lane;
        if (idx < n) {
            const uint32_t cnt = min(255u, cur[idx].score);
            atomicMax(&tab[cur[idx].id >> bshift], (cnt << 24) | (0xFFFFFFu - min(idx, 0xFFFFFEu)));
        }
    }
    PfOvfElem *dst = cur == O ? S : O;
    uint32_t m = 0;
    PfCand *surv = D.surv + (D.cand_base[(uint64_t)q * B] - D.cand_origin);
    for (uint32_t r0 = 0; r0 < n; r0 += 64) {
        const uint32_t idx = r0 + (uint32_t)lane;
        bool keep = false;
        PfOvfElem e;
        e.id = 0; e.score = 0; e.ord = 0; e.diag = 0; e.pad0 = 0; e.pad1 = 0;
        if (idx < n) {
            e = cur[idx];
            const uint32_t cnt = min(255u, e.score);
            const uint32_t first = 0xFFFFFFu - (tab[e.id >> bshift] & 0xFFFFFFu);
            keep = idx == first || (cnt == 0 && idx > first);
            if (final_step) keep = keep && cnt >= D.min_diag_score;
        }
        const uint64_t kb = ballot(keep);
        if (final_step) {
            if (kb) {
                uint32_t sb = 0;
                if (lane == 0) sb = atomicAdd(&D.surv_count[q], (uint32_t)__popcll(kb));
                sb = __shfl(sb, 0);
                if (keep) {
                    const unsigned long long o45 = (unsigned long long)(e.ord + (1ll << 40));   // in (0, 2^41)
                    PfCand c;
                    c.id = e.id;
                    c.arr = (uint32_t)o45;
                    c.score = e.score;
                    c.diag = e.diag;
                    c.pad = (uint16_t)(o45 >> 32);
                    surv[sb + (uint32_t)__popcll(kb & below)] = c;
                }
            }
        } else if (keep) {
            dst[m + (uint32_t)__popcll(kb & below)] = e;
        }
        m += (uint32_t)__popcll(kb);
    }
    if (!final_step) {
        __threadfence();
        if (dst != O) {   // foundDiagonals lives in buf_a between launches
            for (uint32_t r0 = 0; r0 < m; r0 += 64) {
                const uint32_t idx = r0 + (uint32_t)lane;
                if (idx < m) O[idx] = dst[idx];
            }
        }
        if (lane == 0) {
            A.o_count[(size_t)qi * B + bin] = m;
            if (m) atomicAdd(&totals[step], m);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// a9: one workgroup per query over the query's surviving elements (one per target).
__device__ __forceinline__ uint32_t rescaled_count(uint32_t score, float fms) {
    // rescoreHits, QueryMatcher.cpp:576-581
    const uint32_t ns = score - 255u;
    const float sc = (float)min(ns, 65535u);
    const float r = __fmul_rn(__fdiv_rn(sc, fms), 255.0f);
    const double dd = (double)r + 0.5;
    return (uint32_t)(int)dd & 0xFFu;
}

// Ordinal, in the query's stream of similar-k-mer lists, of the list that holds arrival index `arr`: the part of the
// CPU's arrival order that does not depend on which targets a shard holds (windows and their similar k-mers are a
// function of the query alone), used to order elements of different shards of a multi-GPU run (pf_shard_kernels.hip).
__device__ uint32_t list_ordinal(const PfSelectArgs &A, uint32_t q, uint32_t arr) {
    const uint32_t p0 = A.q_off[q], p1 = A.q_off[q + 1];
    uint32_t lo = p0, hi = p1 - 1;            // largest window p with peb[p] <= arr
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (A.peb[mid] <= arr) lo = mid; else hi = mid - 1;
    }
    const uint32_t rel = arr - A.peb[lo];
    uint32_t a = A.list_base[lo], b = A.list_base[lo + 1];
    if (b <= a) return a - A.list_base[p0];
    b -= 1;                                    // largest record r of the window with lprefix <= rel
    while (a < b) {
        const uint32_t mid = (a + b + 1) >> 1;
        if (A.lists[mid].lprefix <= rel) a = mid; else b = mid - 1;
    }
    return a - A.list_base[p0];
}

// XCHG = false: the query's final hit list (QueryMatcher::matchQuery's result).
// XCHG = true : this device holds one shard of the database (mmgpu_pf_set_shard): the top max_hits elements by the
//               unsplit run's own order - count, the reference's cache bin of the GLOBAL id, arrival order - as exchange
//               records for pf_merge_exchange_kernel, which redoes threshold / truncation / final scores over all shards.
// BIG = true: --max-seqs above PF_MAX_HITS - the selected elements do not fit the LDS arrays and are sorted in the query's slice of
// a global scratch instead (same network, global loads / stores; such lists are rare and long anyway).
template <bool XCHG, bool BIG>
__global__ __launch_bounds__(256) void pf_select_kernel(PfSelectArgs A) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_thr, sh_trunc, sh_nelig, sh_nsel, sh_ntie;
    __shared__ uint64_t sh_prefix, sh_mask;
    __shared__ uint32_t sh_remaining;
    __shared__ uint64_t skey_lds[BIG ? 1 : PF_MAX_HITS];
    __shared__ uint16_t sdiag_lds[BIG ? 1 : PF_MAX_HITS];
    const uint32_t q = A.q_first + blockIdx.x;
    uint64_t *skey = BIG ? A.big_keys + (size_t)q * A.big_stride : skey_lds;
    uint16_t *sdiag = BIG ? A.big_diags + (size_t)q * A.big_stride : sdiag_lds;
    const uint32_t sort_cap = BIG ? A.big_stride : (uint32_t)PF_MAX_HITS;
    const uint32_t n = A.surv_count[q];
    const PfCand *S = A.surv + (A.cand_base[(uint64_t)q * A.bins] - A.cand_origin);
    const uint32_t ident = A.q_identity[q];
    const uint32_t max_hits = A.max_hits;   // already min(maxHitsPerQuery, dbSize)
    mmgpu_pf_hit *out = A.hits + (size_t)q * A.hit_stride;

    for (int k = (int)threadIdx.x; k < 256; k += 256) hist[k] = 0;
    if (threadIdx.x == 0) sh_remaining = 0;
    __syncthreads();
    if (A.cand_count != nullptr) {      // resultSize >= foundDiagonalsSize / 2 (QueryMatcher.cpp:188): not this kernel's branch
        uint32_t loc = 0;
        for (uint32_t k = threadIdx.x; k < A.bins; k += 256) loc += A.cand_count[(uint64_t)q * A.bins + k];
        for (int d = 1; d < 64; d <<= 1) loc += __shfl_xor(loc, d);
        if ((threadIdx.x & 63u) == 0 && loc) atomicAdd(&sh_remaining, loc);
        __syncthreads();
        if (threadIdx.x == 0 && sh_remaining >= A.cand_cap && A.q_flags) A.q_flags[q] |= 2u;
        __syncthreads();
    }
    for (uint32_t k = threadIdx.x; k < n; k += 256) atomicAdd(&hist[pf_el_count(S[k].score)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        // computeScoreThreshold, QueryMatcher.h:211-221
        uint32_t found = 0, thr = 0;
        for (thr = 255; thr > 0; thr--) {
            found += hist[thr];
            if (found >= max_hits) break;
        }
        const uint32_t dthr = max(A.min_diag_score, thr);
        sh_thr = dthr;
        sh_trunc = (dthr >= 255u && !A.kmer_score) ? 1u : 0u;   // getResult<KMER_SCORE> has no truncated-threshold path
        sh_nelig = 0;
        sh_nsel = 0;
    }
    __syncthreads();
    const uint32_t dthr = sh_thr;
    const bool trunc = sh_trunc != 0;
    int ms = A.q_self_score[q] - 255;
    ms = ms > 1 ? ms : 1;
    ms = ms < 65535 ? ms : 65535;
    const float fms = (float)ms;
    const uint32_t refmask = A.ref_bins - 1;

    // sort key of an element: (255 - count) : bin of the reference's CacheFriendlyOperations (11 bits) : order key (45 bits)
    auto key_of = [&](const PfCand &c, bool *elig) -> uint64_t {
        const uint32_t cnt = pf_el_count(c.score);
        uint32_t kc;
        if (trunc) {
            *elig = cnt >= 255u && c.id != ident;
            kc = rescaled_count(pf_el_exact(c.score), fms);
        } else {
            *elig = cnt >= dthr && c.id != ident;
            kc = cnt;
        }
        // order inside the CPU's array: arrival index (ordinary queries) or the 48-bit merge key of the overflow path
        // (nucleotide searches: the saturated elements were sorted by target id before keepMax, QueryMatcher.cpp:154)
        const uint64_t ord = (A.nucl && cnt >= 255u) ? (uint64_t)c.id : (((uint64_t)c.pad << 32) | (uint64_t)c.arr);
        const uint32_t gid = XCHG ? A.global_ids[c.id] : c.id;
        if (XCHG) *elig = trunc ? cnt >= 255u : cnt >= dthr;   // the self hit's element takes part (see the merge kernel)
        return ((uint64_t)(255u - kc) << 56) | ((uint64_t)(gid & refmask) << 45) | (ord & ((1ull << 45) - 1));
    };

    uint32_t mine = 0;
    for (uint32_t k = threadIdx.x; k < n; k += 256) {
        bool el;
        (void)key_of(S[k], &el);
        mine += el ? 1u : 0u;
    }
    if (mine) atomicAdd(&sh_nelig, mine);
    __syncthreads();
    const uint32_t nelig = sh_nelig;
    const uint32_t has_ident = (!XCHG && ident != 0xFFFFFFFFu) ? 1u : 0u;
    const uint32_t want = max_hits > has_ident ? max_hits - has_ident : 0u;

    // radix select of the `want` smallest keys (keys are unique: the arrival index is)
    uint64_t kstar = ~0ull;
    if (nelig > want && want > 0) {
        if (threadIdx.x == 0) {
            sh_prefix = 0;
            sh_mask = 0;
            sh_remaining = want;
        }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int k = (int)threadIdx.x; k < 256; k += 256) hist[k] = 0;
            __syncthreads();
            const uint64_t prefix = sh_prefix, mask = sh_mask;
            for (uint32_t k = threadIdx.x; k < n; k += 256) {
                bool el;
                const uint64_t key = key_of(S[k], &el);
                if (el && (key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 0xFFu], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t rem = sh_remaining, cum = 0, d = 0;
                for (d = 0; d < 256; d++) {
                    if (cum + hist[d] >= rem) break;
                    cum += hist[d];
                }
                sh_remaining = rem - cum;
                sh_prefix = prefix | ((uint64_t)d << shift);
                sh_mask = mask | (0xFFull << shift);
                sh_ntie = hist[d];
            }
            __syncthreads();
            if (!BIG && shift == 56 && sh_ntie <= sort_cap) {
                // Round 6: the cut falls inside ONE count class (the top byte of the key), usually a few dozen elements: bring the class's
                // keys into the sort array (free until the gather below), order them there and read the cut off - instead of seven
                // more passes over all the query's survivors
                const uint32_t m = sh_ntie, rem = sh_remaining;
                const uint64_t prefix1 = sh_prefix;
                __syncthreads();
                if (threadIdx.x == 0) sh_ntie = 0;
                __syncthreads();
                for (uint32_t k = threadIdx.x; k < n; k += 256) {
                    bool el;
                    const uint64_t key = key_of(S[k], &el);
                    if (el && (key & (0xFFull << 56)) == prefix1) skey[atomicAdd(&sh_ntie, 1u)] = key;
                }
                uint32_t mp2 = 1;
                while (mp2 < m) mp2 <<= 1;
                __syncthreads();
                for (uint32_t k = m + threadIdx.x; k < mp2; k += 256) skey[k] = ~0ull;
                __syncthreads();
                for (uint32_t size = 2; size <= mp2; size <<= 1) {
                    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                        for (uint32_t k = threadIdx.x; k < mp2 / 2; k += 256) {
                            const uint32_t i = 2 * k - (k & (stride - 1));
                            const uint32_t j = i + stride;
                            const bool up = (i & size) == 0;
                            const uint64_t a = skey[i], b = skey[j];
                            if ((a > b) == up) {
                                skey[i] = b;
                                skey[j] = a;
                            }
                        }
                        __syncthreads();
                    }
                }
                if (threadIdx.x == 0) sh_prefix = skey[rem - 1];      // the rem-th smallest key of the class: everything up to it is taken
                __syncthreads();
                break;
            }
        }
        kstar = sh_prefix;
    }
    __syncthreads();

    // gather the selected elements, compute prefScore (getResult, QueryMatcher.cpp:430-452)
    if (want > 0) {
        for (uint32_t k = threadIdx.x; k < n; k += 256) {
            bool el;
            const PfCand c = S[k];
            const uint64_t key = key_of(c, &el);
            if (el && key <= kstar) {
                const uint32_t slot = atomicAdd(&sh_nsel, 1u);
                if (XCHG) {
                    if (slot < max_hits) {
                        const bool ovf = (A.q_nseg != nullptr && A.q_nseg[q] != 0) || (A.q_flags != nullptr && A.q_flags[q] != 0);
                        mmgpu_pf_xhit x;
                        x.id = A.global_ids[c.id];
                        x.score = c.score;
                        x.diagonal = c.diag;
                        x.flags = (uint16_t)((ovf ? MMGPU_PF_X_INEXACT_ORDER : 0) | (c.id == ident ? MMGPU_PF_X_IDENTITY : 0));
                        x.order = ovf ? c.arr : list_ordinal(A, q, c.arr);
                        A.xhits[(size_t)q * A.hit_stride + slot] = x;
                    }
                } else if (slot < sort_cap) {
                    uint32_t pref;
                    const uint32_t cnt = pf_el_count(c.score);
                    if (trunc) pref = 255u + (rescaled_count(pf_el_exact(c.score), fms) * (uint32_t)ms / 255u);
                    else pref = (cnt >= 255u && !A.kmer_score) ? pf_el_exact(c.score) : cnt;
                    skey[slot] = ((uint64_t)(0xFFFFFFFFu - pref) << 32) | (uint64_t)c.id;
                    sdiag[slot] = c.diag;
                }
            }
        }
    }
    __syncthreads();
    if (XCHG) {
        if (threadIdx.x == 0) {
            // bit 31 of the exchanged count: this shard saw the query take (or possibly take) a branch of the reference that
            // depends on the WHOLE database - the databaseHits overflow path (its share of the entries reached its share of the
            // limit), an unscored long sequence, the unsorted branch.  The merge flags the query whichever elements survive.
            const bool whole_db = (A.q_nseg != nullptr && A.q_nseg[q] != 0) || (A.q_flags != nullptr && A.q_flags[q] != 0);
            A.hit_count[q] = min(sh_nsel, max_hits) | (whole_db ? 0x80000000u : 0u);
            A.q_diag_thr[q] = dthr | (trunc ? 0x80000000u : 0u);
        }
        return;
    }
    if (BIG) __threadfence();      // the scratch slice is read back by other threads of the workgroup
    const uint32_t nsel = min(sh_nsel, sort_cap);
    // bitonic sort by (prefScore desc, id asc)   (hit_t::compareHitsByScoreAndId, QueryMatcher.h:38-49)
    uint32_t np2 = 1;
    while (np2 < nsel) np2 <<= 1;
    for (uint32_t k = nsel + threadIdx.x; k < np2; k += 256) {
        skey[k] = ~0ull;
        sdiag[k] = 0;
    }
    if (BIG) __threadfence();
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < np2 / 2; k += 256) {
                const uint32_t i = 2 * k - (k & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = skey[i], b = skey[j];
                if ((a > b) == up) {
                    skey[i] = b;
                    skey[j] = a;
                    const uint16_t t = sdiag[i];
                    sdiag[i] = sdiag[j];
                    sdiag[j] = t;
                }
            }
            if (BIG) __threadfence();
            __syncthreads();
        }
    }
    for (uint32_t k = threadIdx.x; k < nsel; k += 256) {
        mmgpu_pf_hit h;
        h.id = (uint32_t)skey[k];
        h.score = (int32_t)(0xFFFFFFFFu - (uint32_t)(skey[k] >> 32));
        h.diagonal = sdiag[k];
        h.reserved = 0;
        out[has_ident + k] = h;
    }
    if (threadIdx.x == 0) {
        if (has_ident && max_hits > 0) {   // self hit first, score USHRT_MAX (getResult :408-424)
            mmgpu_pf_hit h;
            h.id = ident;
            h.score = A.kmer_score ? 255 : 65535;      // UCHAR_MAX for getResult<KMER_SCORE> (:410-413)
            h.diagonal = 0;
            h.reserved = 0;
            out[0] = h;
        }
        A.hit_count[q] = (max_hits > 0 ? has_ident : 0u) + nsel;
        A.q_diag_thr[q] = dthr | (trunc ? 0x80000000u : 0u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Device analogue of Prefiltering::mergeTargetSplits (Prefiltering.cpp:412-526): the hit lists of one query from
// n_splits target shards (gathered with one all-gather over RCCL) are concatenated, shard-local ids become global
// ids (+ id_offset[split], the dbFrom convention of Prefiltering.cpp:879-881) and the result is sorted with
// hit_t::compareHitsByScoreAndId.  One workgroup per query, bitonic sort in LDS.
__global__ __launch_bounds__(256) void pf_merge_kernel(PfMergeArgs A) {
    __shared__ uint64_t skey[PF_MERGE_CAP];
    __shared__ uint16_t sdiag[PF_MERGE_CAP];
    __shared__ uint32_t sbase[65];
    const uint32_t q = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t sp = 0; sp < A.n_splits; sp++) {
            sbase[sp] = run;
            run += min(A.counts[(size_t)sp * A.nq + q], A.stride);
        }
        sbase[A.n_splits] = run;
    }
    __syncthreads();
    const uint32_t total = sbase[A.n_splits];
    for (uint32_t sp = 0; sp < A.n_splits; sp++) {
        const uint32_t n = sbase[sp + 1] - sbase[sp];
        const mmgpu_pf_hit *src = A.hits + ((size_t)sp * A.nq + q) * A.stride;
        for (uint32_t k = threadIdx.x; k < n; k += 256) {
            const mmgpu_pf_hit h = src[k];
            const uint32_t a = (uint32_t)(h.score < 0 ? -h.score : h.score);
            skey[sbase[sp] + k] = ((uint64_t)(0xFFFFFFFFu - a) << 32) | (uint64_t)(h.id + A.id_offset[sp]);
            sdiag[sbase[sp] + k] = h.diagonal;
        }
    }
    uint32_t np2 = 1;
    while (np2 < total) np2 <<= 1;
    __syncthreads();
    for (uint32_t k = total + threadIdx.x; k < np2; k += 256) {
        skey[k] = ~0ull;
        sdiag[k] = 0;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < np2 / 2; k += 256) {
                const uint32_t i = 2 * k - (k & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = skey[i], b = skey[j];
                if ((a > b) == up) {
                    skey[i] = b;
                    skey[j] = a;
                    const uint16_t t = sdiag[i];
                    sdiag[i] = sdiag[j];
                    sdiag[j] = t;
                }
            }
            __syncthreads();
        }
    }
    mmgpu_pf_hit *dst = A.out_hits + (size_t)q * A.n_splits * A.stride;
    for (uint32_t k = threadIdx.x; k < total; k += 256) {
        mmgpu_pf_hit h;
        h.id = (uint32_t)skey[k];
        h.score = (int32_t)(0xFFFFFFFFu - (uint32_t)(skey[k] >> 32));
        h.diagonal = sdiag[k];
        h.reserved = 0;
        dst[k] = h;
    }
    if (threadIdx.x == 0) A.out_counts[q] = total;
}

}  // namespace

// cofs: PF_COFS_DWORDS dwords per block of PF_COFS_KMERS k-mers, ceil(table / PF_COFS_KMERS) blocks
size_t pf_cofs_bytes(uint64_t table) { return (size_t)((table + PF_COFS_KMERS - 1) / PF_COFS_KMERS) * PF_COFS_DWORDS * 4; }
hipError_t launch_pf_cofs(const uint32_t *offsets, uint64_t table, void *cofs, hipStream_t s) {
    if (table == 0) return hipSuccess;
    const uint64_t blocks = (table + PF_COFS_KMERS - 1) / PF_COFS_KMERS;
    hipLaunchKernelGGL(pf_cofs_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, s, offsets, table, reinterpret_cast<uint32_t *>(cofs));
    return hipGetLastError();
}

hipError_t launch_pf_kmers(const PfKmerArgs &A, bool emit, hipStream_t s) {
    if (A.n_pos == 0) return hipSuccess;
    if (A.exact) {
        const dim3 g1((A.n_pos + 255) / 256), b1(256);
        if (emit) hipLaunchKernelGGL(pf_kmers_exact_kernel<true>, g1, b1, 0, s, A);
        else hipLaunchKernelGGL(pf_kmers_exact_kernel<false>, g1, b1, 0, s, A);
        return hipGetLastError();
    }
    const dim3 grid((A.n_pos + 3) / 4), block(256);
    if (A.k == 7) {
        if (emit) hipLaunchKernelGGL(pf_kmers7_kernel<true>, grid, block, 0, s, A);
        else hipLaunchKernelGGL(pf_kmers7_kernel<false>, grid, block, 0, s, A);
    } else {
        if (emit) hipLaunchKernelGGL(pf_kmers_kernel<true>, grid, block, 0, s, A);
        else hipLaunchKernelGGL(pf_kmers_kernel<false>, grid, block, 0, s, A);
    }
    if (A.q_kind) {      // the batch holds profile queries: their positions are enumerated by the profile generator
        if (A.k == 7) {
            if (emit) hipLaunchKernelGGL((pf_kmers_prof_kernel<7, true>), grid, block, 0, s, A);
            else hipLaunchKernelGGL((pf_kmers_prof_kernel<7, false>), grid, block, 0, s, A);
        } else if (A.k == 5) {
            if (emit) hipLaunchKernelGGL((pf_kmers_prof_kernel<5, true>), grid, block, 0, s, A);
            else hipLaunchKernelGGL((pf_kmers_prof_kernel<5, false>), grid, block, 0, s, A);
        } else {
            if (emit) hipLaunchKernelGGL((pf_kmers_prof_kernel<6, true>), grid, block, 0, s, A);
            else hipLaunchKernelGGL((pf_kmers_prof_kernel<6, false>), grid, block, 0, s, A);
        }
    }
    return hipGetLastError();
}

// one bit per k-mer: does its index list hold an entry?  (a sparse index - a shard of a multi-GPU run, a small database - answers most
// similar k-mers of a query from this 8 MB table instead of a 64-byte sector of the 256 MB offset table)
__global__ __launch_bounds__(256) void pf_bitmap_kernel(const uint32_t *offsets, uint64_t table, uint32_t *bitmap, unsigned long long *nonempty) {
    const uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t k0 = w * 32u;
    uint32_t bits = 0;
    if (k0 < table) {
        uint32_t prev = offsets[k0];
        for (uint32_t b = 0; b < 32u && k0 + b < table; b++) {
            const uint32_t next = offsets[k0 + b + 1];
            if (next != prev) bits |= 1u << b;
            prev = next;
        }
        bitmap[w] = bits;
    }
    unsigned long long n = (unsigned long long)__popc(bits);
    for (int d = 1; d < 64; d <<= 1) n += __shfl_xor(n, d);
    if (lane_id() == 0 && n) atomicAdd(nonempty, n);
}

// tile t of query q: (tile_q, tile_idx) = (q, t) - the split kernel's work list, written where it is read
__global__ __launch_bounds__(256) void pf_tiles_kernel(const uint32_t *q_tile_base, const uint32_t *q_ntiles, uint32_t nq, uint32_t *tile_q, uint32_t *tile_idx) {
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const uint32_t base = q_tile_base[q], n = q_ntiles[q];
    for (uint32_t t = threadIdx.x; t < n; t += 256) {
        tile_q[base + t] = q;
        tile_idx[base + t] = t;
    }
}

hipError_t launch_pf_tiles(const uint32_t *q_tile_base, const uint32_t *q_ntiles, uint32_t nq, uint32_t *tile_q, uint32_t *tile_idx, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_tiles_kernel, dim3(nq), dim3(256), 0, s, q_tile_base, q_ntiles, nq, tile_q, tile_idx);
    return hipGetLastError();
}

hipError_t launch_pf_bitmap(const uint32_t *offsets, uint64_t table, uint32_t *bitmap, unsigned long long *nonempty, hipStream_t s) {
    const uint64_t words = (table + 31) / 32;
    hipLaunchKernelGGL(pf_bitmap_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, offsets, table, bitmap, nonempty);
    return hipGetLastError();
}

hipError_t launch_pf_scan(const uint32_t *in, const uint32_t *q_off, uint32_t nq, const uint64_t *base, uint32_t *out,
                          uint64_t *totals, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_scan_kernel, dim3(nq), dim3(256), 0, s, in, q_off, nq, base, out, totals);
    return hipGetLastError();
}

hipError_t launch_pf_split(const PfSplitArgs &A, uint32_t n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_split_kernel, dim3(n_tiles), dim3(SPW * 64), (size_t)SPW * A.bins * sizeof(uint16_t), s, A);
    return hipGetLastError();
}

hipError_t launch_pf_dedup(const PfDedupArgs &A, hipEvent_t after_replay, hipEvent_t after_ungapped, hipStream_t s) {
    const uint64_t buckets = (uint64_t)A.n_queries * A.bins;
    if (buckets == 0) return hipSuccess;
    const dim3 grid((unsigned)((buckets + 3) / 4)), block(256);
    hipError_t e;
    hipLaunchKernelGGL(pf_replay_kernel, grid, block, 0, s, A);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (after_replay && (e = hipEventRecord(after_replay, s)) != hipSuccess) return e;
    // (with the replay kernel's work list of the larger buckets: a fixed grid whose wavefronts walk it)
    const dim3 grid_big(A.big_list ? (unsigned)std::min<uint64_t>((buckets + 3) / 4, 2048) : grid.x);
    hipLaunchKernelGGL(pf_ungapped_kernel, grid_big, block, 0, s, A);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (after_ungapped && (e = hipEventRecord(after_ungapped, s)) != hipSuccess) return e;
    if (A.nucl) hipLaunchKernelGGL(pf_keepmax_nucl_kernel, dim3((unsigned)((buckets + 1) / 2)), dim3(128), 0, s, A);
    else if (A.big_list) hipLaunchKernelGGL(pf_keepmax_list_kernel, dim3((unsigned)std::min<uint64_t>(buckets, 1024)), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL(pf_keepmax_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_long(const PfDedupArgs &A, bool long_queries, hipStream_t s) {
    if (A.n_queries == 0 || A.q_flags == nullptr) return hipSuccess;
    if (long_queries) {
        const uint64_t buckets = (uint64_t)A.n_queries * A.bins;
        hipLaunchKernelGGL(pf_longq_kernel, dim3((unsigned)((buckets + 3) / 4)), dim3(256), 0, s, A);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(pf_long_kernel, dim3(A.n_queries), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_count(const PfDedupArgs &A, hipEvent_t after_a, hipEvent_t after_b, hipStream_t s) {
    const uint64_t buckets = (uint64_t)A.n_queries * A.bins;
    if (buckets == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_count_kernel, dim3((unsigned)((buckets + 3) / 4)), dim3(256), 0, s, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (after_a && (e = hipEventRecord(after_a, s)) != hipSuccess) return e;
    if (after_b && (e = hipEventRecord(after_b, s)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_pf_segments(const PfSegArgs &A, hipStream_t s) {
    if (A.n_ovf == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_segments_kernel, dim3(A.n_ovf), dim3(64), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_overflow(const PfOvfArgs &A, hipStream_t s) {
    const uint64_t waves = (uint64_t)A.n_ovf * A.D.bins;
    if (waves == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_overflow_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_merge(const PfMergeArgs &A, hipStream_t s) {
    if (A.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(pf_merge_kernel, dim3(A.nq), dim3(256), 0, s, A);
    return hipGetLastError();
}

hipError_t launch_pf_select(const PfSelectArgs &A, uint32_t nq, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    if (A.xhits) hipLaunchKernelGGL((pf_select_kernel<true, false>), dim3(nq), dim3(256), 0, s, A);
    else if (A.big_keys) hipLaunchKernelGGL((pf_select_kernel<false, true>), dim3(nq), dim3(256), 0, s, A);
    else hipLaunchKernelGGL((pf_select_kernel<false, false>), dim3(nq), dim3(256), 0, s, A);
    return hipGetLastError();
}

// first use of any kernel of this file loads its code object (tens of milliseconds): mmgpu_warmup does it ahead of time
void warm_pf() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&pf_kmers_kernel<true>));
}

}  // namespace mmgpu
