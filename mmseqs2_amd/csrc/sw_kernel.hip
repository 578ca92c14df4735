// Gotoh Smith-Waterman (affine gaps, local) for gfx950 / CDNA4 - score, end position, start position.
//
// What it computes: exactly what SmithWaterman::alignScoreEndPos (src/alignment/
// StripedSmithWaterman.cpp:892-941; sw_sse2_byte :98-299, sw_sse2_word :301-476) and the reverse scan of
// alignStartPosBacktrace (:1129-1204) compute on the CPU, bit for bit:
//     H = max(sat16(Hdiag + P), E, F);  E' = max(E - ge, H - go);  F' = max(F - ge, H - go)   (E, F >= 0)
//     t_end = first column in which the running maximum reaches its final value (:232-248, :414-425)
//     q_end = smallest query index holding that maximum in that column           (:262-271, :438-447)
// The reference runs a uint8 pass and re-runs in int16 on overflow; here everything is int16 (exact for
// every score the uint8 pass can represent, and saturating at 32767 exactly like sw_sse2_word), and the
// `word` flag reports which pass the reference would have ended in (score + bias >= 255, :238-242).
//
// How it is mapped to CDNA4 (this is not the striped/lazy-F formulation of the CPU code):
//  * integer DP => VALU, no MFMA.  All cell arithmetic is packed 2 x int16 (v_pk_add_i16 clamp,
//    v_pk_max_i16/u16, v_pk_sub_u16 clamp): the low half of every register belongs to target A, the
//    high half to target B, so one VALU lane-op advances two independent alignments.
//  * a group of 16 lanes (one DPP row) owns one pair of targets; lane g owns R consecutive query rows
//    (R = 1,2,..,32 -> 16..512 rows per tile) whose H/E state lives in registers for the whole scan.
//    Lanes run skewed by one column (anti-diagonal wavefront): at step s lane g works on column s - g.
//    The hand-off lane g -> g+1 (H of the strip's last row, the F leaving it, and the two target
//    letters) is three v_mov_b32_dpp row_shr:1 - no LDS, no bpermute.
//  * the query profile P[letter][row] (int16, composition bias folded in) is built once per workgroup in
//    LDS; per step a lane fetches its R scores for letter a (target A) and for letter b (target B) with
//    ds_read_b128 and interleaves them with v_perm_b32.  The per-lane row stride is padded to an odd
//    number of 16-byte slots so the 16 lanes of a group hit 64 distinct banks.
//  * four groups per wave64, four waves per workgroup: 32 targets in flight per workgroup, all against
//    the same query (the prefilter list of one query, pre-sorted by target length so the groups of a
//    wave finish together).
//  * end positions: each lane keeps the packed running maximum of its strip, the column where it was
//    last raised and a snapshot of the strip's H values at that column (v_bfi under a rarely taken
//    branch); a 16-lane DPP/shuffle reduction at the end applies the reference's tie rules.
//  * queries longer than one tile loop over tiles inside the kernel; the H/F row leaving the last lane
//    is parked in a double-buffered global scratch line and re-enters at the head lane of the next tile.
//
// Bytes per cell that ever touch HBM: (tlen + 28) / (qlen * tlen) ~ 0.003 - this kernel
// is bound by VALU issue, not by HBM.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

constexpr int GROUP = 16;            // lanes per target pair (one DPP row)
constexpr int WAVES = 4;             // waves per workgroup
constexpr int GROUPS_PER_WAVE = 64 / GROUP;
constexpr int HITS_PER_WAVE = 2 * GROUPS_PER_WAVE;   // two targets per group (lo/hi halves)
constexpr unsigned NEG2 = 0x80008000u;               // packed (-32768, -32768)

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_add_sat(unsigned a, unsigned b) {   // v_pk_add_i16 ... clamp
    return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a),
                                                                      __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ unsigned pk_max_s(unsigned a, unsigned b) {     // v_pk_max_i16
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a),
                                                                  __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ unsigned pk_max_u(unsigned a, unsigned b) {     // v_pk_max_u16
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a),
                                                                  __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_sub_sat_u(unsigned a, unsigned b) { // v_pk_sub_u16 ... clamp
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a),
                                                                      __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned bfi(unsigned mask, unsigned a, unsigned b) {   // v_bfi_b32
    return (a & mask) | (b & ~mask);
}
// value of the lane one below inside the 16-lane row; lane 0 of each row keeps `head`
__device__ __forceinline__ unsigned from_lane_above(unsigned head, unsigned v) {
    return __builtin_amdgcn_update_dpp(head, v, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
}

__host__ __device__ constexpr int lane_stride_bytes(int R) {
    // R*2 bytes of scores per lane, rounded up to an odd number of 16-byte slots (bank spread)
    int slots = (R * 2 + 15) / 16;
    if ((slots & 1) == 0) slots += 1;
    return slots * 16;
}

template <int R>
struct Tile {
    static constexpr int ROWS = GROUP * R;
    static constexpr int LANE_STRIDE = lane_stride_bytes(R);
    static constexpr int ROW_STRIDE = GROUP * LANE_STRIDE;   // bytes per letter
};

// Build P[letter][row] for rows [tile_base, tile_base + ROWS) of the query (or of the reversed query).
// Rows past the query end and the extra letter `alphabet` (padding column of a finished target) are
// -32768, which keeps H at max(E, F) there and can never raise a running maximum.
template <int R, bool REV>
__device__ __forceinline__ void build_profile(unsigned char *lds, const uint8_t *q, const int8_t *cb, int qlen,
                                              int tile_base, const int8_t *mat, int alphabet, const int8_t *prof) {
    using T = Tile<R>;
    const int total = (alphabet + 1) * T::ROWS;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int letter = idx / T::ROWS;
        const int row = idx - letter * T::ROWS;
        const int rg = tile_base + row;
        short v = (short)-32768;
        if (letter < alphabet && rg < qlen) {
            const int qi = REV ? (qlen - 1 - rg) : rg;
            // profile query: the score row of the letter (createQueryProfile<PROFILE>, StripedSmithWaterman.cpp:773-776)
            v = prof ? (short)prof[letter * qlen + qi] : (short)((int)mat[letter * alphabet + q[qi]] + (int)cb[qi]);
        }
        const int lane = row / R, r = row - lane * R;
        *reinterpret_cast<short *>(lds + letter * T::ROW_STRIDE + lane * T::LANE_STRIDE + r * 2) = v;
    }
}

// Does the pair whose forward result sits in out[slot] get a reverse scan?  ssw_align_private scans backwards for the start
// position when the score passes the E-value gate (StripedSmithWaterman.cpp:857-863) - and, in the reference's alignment modes with a
// start position, only for hits of the uint8 pass: an int16-range hit (word == 1) takes its start from the block aligner (:865-882),
// falling back to the reverse scan when that declines (rev_mode 2: the caller names those pairs afterwards).
__device__ __forceinline__ bool sw_rev_wanted(const SwLaunch &L, uint32_t slot, int min_start) {
    const mmgpu_sw_hit &f = L.out[slot];
    const int s = f.score;
    if (L.rev_mode == 2) return s > 0 && L.rev_force[slot] != 0;
    return s > 0 && s >= min_start && !(L.rev_mode == 1 && f.word != 0);
}

constexpr int SW_REV_JOB_HITS = 1024;   // most hits a reverse-scan job may hold (mmgpu_internal.h: SW_REV_JOB_MAX)
static_assert(SW_REV_JOB_HITS == SW_REV_JOB_MAX, "host and kernel disagree on the reverse job size");
constexpr int SW_LDS_HEADER = SW_REV_JOB_HITS * 2 + 64;

template <int R, bool MULTI, bool REV>
__device__ __forceinline__ void sw_body(const SwLaunch &L, const SwJob job) {
    using T = Tile<R>;
    // dynamic LDS: a 1 KB header (job-local scheduling state, below) followed by the query profile of the tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    uint16_t *live = reinterpret_cast<uint16_t *>(lds_all);                  // [SW_REV_JOB_HITS] packed live hits (reverse pass)
    uint32_t *live_wave = reinterpret_cast<uint32_t *>(lds_all + SW_REV_JOB_HITS * 2);       // [WAVES]
    uint32_t *next_chunk = reinterpret_cast<uint32_t *>(lds_all + SW_REV_JOB_HITS * 2 + 16); // next 8-hit chunk to hand to a wave
    unsigned char *lds = lds_all + SW_LDS_HEADER;
    // A multi-tile job is one long dependent chain (tiles x columns) however few targets it holds, and the batch ends
    // with the longest of them: those waves take the issue slots first, the short jobs sharing the SIMD fill the gaps.
    if (MULTI) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane & (GROUP - 1);
    const int grp = lane / GROUP;

    const uint32_t qbeg = L.q_off[job.query];
    const int qlen = (int)(L.q_off[job.query + 1] - qbeg);
    const uint8_t *q = L.q_res + qbeg;
    const int8_t *cb = L.q_cb + qbeg;
    const int qbias = L.q_bias[job.query];
    const int8_t *prof = nullptr;
    if (L.q_prof_off) {
        const uint32_t po = L.q_prof_off[job.query];
        if (po != 0xFFFFFFFFu) prof = L.q_prof + po;
    }
    const int n_tiles = MULTI ? (qlen + T::ROWS - 1) / T::ROWS : 1;

    const unsigned go2 = (unsigned)L.gap_open * 0x10001u;
    const unsigned ge2 = (unsigned)L.gap_extend * 0x10001u;
    const unsigned pad_letter = (unsigned)L.alphabet;

    if (threadIdx.x == 0) *next_chunk = 0;
    if (!MULTI && !REV) {
        build_profile<R, REV>(lds, q, cb, qlen, 0, L.mat, L.alphabet, prof);
        __syncthreads();
    }

    uint32_t n_hits = job.hit_end - job.hit_begin;
    // Reverse pass: only the pairs whose forward score reaches the query's start-score threshold are scanned
    // (ssw_align_private, StripedSmithWaterman.cpp:857-863) - typically one hit in six, scattered over the
    // length-sorted list.  They are packed to the front (order kept), so that the waves run full instead of every
    // wave waiting for its one or two live targets.
    if (REV) {
        const int min_start = L.q_minstart[job.query];
        uint32_t total = 0;
        for (uint32_t base = 0; base < n_hits; base += WAVES * 64) {   // a job of the BOTH kernels holds at most 256 hits: one round
            const uint32_t idx = base + threadIdx.x;
            bool pass = false;
            if (idx < n_hits) pass = sw_rev_wanted(L, L.hit_out[job.hit_begin + idx], min_start);
            const unsigned long long bal = __ballot(pass);
            if (lane == 0) live_wave[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, round_total = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = live_wave[w];
                before += w < wave ? c : 0u;
                round_total += c;
            }
            if (pass) live[total + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)idx;
            total += round_total;
            __syncthreads();
        }
        n_hits = total;
        if (n_hits == 0) return;      // (workgroup-uniform) no pair of this job needs a start position: no profile either
        if (!MULTI) {
            build_profile<R, REV>(lds, q, cb, qlen, 0, L.mat, L.alphabet, prof);
            __syncthreads();
        }
    }
    const uint32_t n_iter = (n_hits + WAVES * HITS_PER_WAVE - 1) / (WAVES * HITS_PER_WAVE);

    for (uint32_t it = 0; !MULTI || it < n_iter; ++it) {
        // Single-tile jobs: a wave takes the next chunk of 8 hits when it is done with its last one (the hits are
        // sorted by length, so fixed wave <-> chunk striping would give wave 0 the longest chunk of every round).
        // Multi-tile jobs rebuild the profile per tile with all threads, so their waves stay in step.
        uint32_t chunk = it * WAVES + wave;
        if (!MULTI) {
            if (lane == 0) chunk = atomicAdd(next_chunk, 1u);
            chunk = __builtin_amdgcn_readfirstlane(chunk);
        }
        const uint32_t slot0 = chunk * HITS_PER_WAVE;
        if (!MULTI && slot0 >= n_hits) break;   // wave-uniform; MULTI keeps every wave in the barrier loop

        // ---- the two targets of this group ------------------------------------------------------
        const uint32_t sA = slot0 + grp * 2, sB = sA + 1;
        const bool vA = sA < n_hits, vB = sB < n_hits;
        const uint32_t hA = job.hit_begin + (REV ? (uint32_t)live[vA ? sA : 0] : sA);
        const uint32_t hB = job.hit_begin + (REV ? (uint32_t)live[vB ? sB : 0] : sB);
        const uint32_t tA = vA ? L.hit_target[hA] : 0u, tB = vB ? L.hit_target[hB] : 0u;
        const uint8_t *pA = L.t_res + (size_t)L.t_off4[tA] * 4;
        const uint8_t *pB = L.t_res + (size_t)L.t_off4[tB] * 4;
        int colsA = vA ? (int)L.t_len[tA] : 0, colsB = vB ? (int)L.t_len[tB] : 0;
        int r0A = 0, r0B = 0;       // REV: first live row of the reversed query
        int endA = 0, endB = 0;     // REV: forward t_end
        unsigned rev_target = 0xFFFFFFFFu;
        if (REV) {
            // reverse scan over q[0..q_end] x t[0..t_end], both walked backwards (:1143-1175)
            // Lanes without a live pair must not take positions from a result slot: slot 0 may still be unwritten
            // (another workgroup's forward pass), and an arbitrary t_end there becomes a wild target address below.
            mmgpu_sw_hit fa, fb;
            fa.score = 0; fa.q_end = 0; fa.t_end = 0;
            fb = fa;
            if (vA) fa = L.out[L.hit_out[hA]];
            if (vB) fb = L.out[L.hit_out[hB]];
            // (every packed hit passed sw_rev_wanted above)
            colsA = vA ? fa.t_end + 1 : 0;
            colsB = vB ? fb.t_end + 1 : 0;
            endA = colsA > 0 ? fa.t_end : 0;
            endB = colsB > 0 ? fb.t_end : 0;
            r0A = colsA > 0 ? qlen - 1 - fa.q_end : 0;
            r0B = colsB > 0 ? qlen - 1 - fb.q_end : 0;
            // the score the reverse scan has to reach (terminate = r.score1, :1159-1175); 0xFFFF can never be reached
            rev_target = (colsA > 0 ? (unsigned)fa.score & 0xFFFFu : 0xFFFFu) | ((colsB > 0 ? (unsigned)fb.score & 0xFFFFu : 0xFFFFu) << 16);
        }
        int ncols = colsA > colsB ? colsA : colsB;
        ncols = max(ncols, __shfl_xor(ncols, 16));
        ncols = max(ncols, __shfl_xor(ncols, 32));
        ncols = __builtin_amdgcn_readfirstlane(ncols);
        const int nsteps = ncols + GROUP - 1;

        unsigned long long bestA = 0, bestB = 0;   // (score << 32) | (~col << 16) | ~row, maximised

        uint2 *scr = nullptr;
        if (MULTI) scr = L.scratch + ((size_t)((job.shape >> 8) * WAVES + wave) * GROUPS_PER_WAVE + grp) * 2 * L.scratch_cols;

        for (int tile = 0; tile < n_tiles; ++tile) {
            const int tile_base = tile * T::ROWS;
            if (MULTI) {
                __syncthreads();   // everyone is done reading the previous tile's profile
                build_profile<R, REV>(lds, q, cb, qlen, tile_base, L.mat, L.alphabet, prof);
                __syncthreads();
            }
            const uint2 *scr_in = MULTI ? scr + (size_t)((tile + 1) & 1) * L.scratch_cols : nullptr;
            uint2 *scr_out = MULTI ? scr + (size_t)(tile & 1) * L.scratch_cols : nullptr;
            const bool has_above = MULTI && tile > 0;
            const bool has_below = MULTI && tile + 1 < n_tiles;

            // forward: snap[] = the strip's H values in the column where the running maximum last rose (q_end);
            // reverse: rmask[] = rows of the reversed query that lie inside q[0..q_end] - the reverse scan needs no
            // snapshot, it ends in the first column that reaches the forward score (see below), so the two bodies of a
            // kernel need the same number of registers and the forward scan keeps its own occupancy
            unsigned Hp[R], E[R], aux[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { Hp[r] = 0; E[r] = 0; aux[r] = 0; }
            if (REV) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int rg = tile_base + g * R + r;
                    aux[r] = (rg >= r0A ? 0xFFFFu : 0u) | (rg >= r0B ? 0xFFFF0000u : 0u);
                }
            }
            unsigned vmax = 0, bestcol = 0xFFFFFFFFu;
            unsigned rev_rowA = 0, rev_rowB = 0;   // REV: first row holding the forward score in the column that reached it
            unsigned Hup_prev = 0;
            unsigned out_H = 0, out_F = 0, out_let = pad_letter * 0x101u;

            unsigned la0 = 0, lb0 = 0;
            // Four columns' letters per dword load (every lane of the group loads the same word).  Forward: targets
            // start 4-byte aligned and the residue buffer ends with max_len + 64 bytes of slack (mmgpu_load_targets), so
            // no clamping is needed: columns past a target's end are replaced by the padding letter below.  Reverse:
            // column c is byte t[t_end - c], block k = the (unaligned) word at t_end - 3 - 4k read from its top byte down;
            // words wholly before the target are clamped to offset -3 (the buffer starts with 64 bytes of padding).
            // One v_bfe per target and column instead of an address computation and a byte load: 16-18 fewer
            // instructions per column (R = 12: 194 -> 177).
            auto letters4 = [&](const uint8_t *p, int end, int k) -> unsigned {
                if (!REV) return reinterpret_cast<const unsigned *>(p)[k];
                int o = end - 3 - 4 * k;
                o = o < -3 ? -3 : o;
                unsigned w;
                __builtin_memcpy(&w, p + o, 4);
                return w;
            };
            unsigned wA = letters4(pA, endA, 0), wB = letters4(pB, endB, 0);
            unsigned wA_next = letters4(pA, endA, 1), wB_next = letters4(pB, endB, 1);
            // boundary row of the tile above: column s is needed at step s; the load of column s + UPD is issued at step s
            // (UPD columns of ~300-400 instructions each cover the L2 round trip at two waves per SIMD)
#ifndef MMGPU_SW_UP_DEPTH
#define MMGPU_SW_UP_DEPTH 4
#endif
            constexpr int UPD = MMGPU_SW_UP_DEPTH;
            uint2 up[UPD];
#pragma unroll
            for (int k = 0; k < UPD; ++k) up[k] = make_uint2(0, 0);
            if (has_above) {
                const int last = ncols - 1 < 0 ? 0 : ncols - 1;
#pragma unroll
                for (int k = 0; k < UPD; ++k) up[k] = scr_in[min(k, last)];
            }

            // blocks of four columns (one dword of letters per target) where the dword path is used, else one block
            for (int s0 = 0; s0 < nsteps; s0 += 4) {
            const int jn = min(4, nsteps - s0);
#pragma nounroll
            for (int j = 0; j < jn; ++j) {
                const int s = s0 + j;
                // ---- inputs of this step: from the lane above, or the tile boundary for the head lane ----
                {
                    const unsigned sh = (unsigned)(REV ? 3 - j : j) * 8u;   // wave-uniform
                    la0 = __builtin_amdgcn_ubfe(wA, sh, 8u);
                    lb0 = __builtin_amdgcn_ubfe(wB, sh, 8u);
                }
                const unsigned head_let = (s < colsA ? la0 : pad_letter) | ((s < colsB ? lb0 : pad_letter) << 8);
                const unsigned Hup = from_lane_above(up[0].x, out_H);
                unsigned f = from_lane_above(up[0].y, out_F);
                const unsigned let = from_lane_above(head_let, out_let);
                if (has_above) {
#pragma unroll
                    for (int k = 0; k + 1 < UPD; ++k) up[k] = up[k + 1];
                    int c = s + UPD; if (c > ncols - 1) c = ncols - 1; if (c < 0) c = 0;
                    up[UPD - 1] = scr_in[c];
                }

                const unsigned a = let & 0xFFu, b = (let >> 8) & 0xFFu;
                const uint4 *rowA = reinterpret_cast<const uint4 *>(lds + a * T::ROW_STRIDE + g * T::LANE_STRIDE);
                const uint4 *rowB = reinterpret_cast<const uint4 *>(lds + b * T::ROW_STRIDE + g * T::LANE_STRIDE);

                unsigned hd = Hup_prev;
                Hup_prev = Hup;
                unsigned cmax = 0;
                // R scores per target as ND = ceil(R / 2) packed dwords (odd R: the high half of the last one is row padding
                // nobody selects): ds_read_b128 for whole 16-byte slots, b64 / b32 for the tail
                constexpr int ND = (R + 1) / 2;
                unsigned pav[ND], pbv[ND];
#pragma unroll
                for (int k = 0; k < ND / 4; ++k) {
                    const uint4 pa = rowA[k], pb = rowB[k];
                    pav[4 * k] = pa.x; pav[4 * k + 1] = pa.y; pav[4 * k + 2] = pa.z; pav[4 * k + 3] = pa.w;
                    pbv[4 * k] = pb.x; pbv[4 * k + 1] = pb.y; pbv[4 * k + 2] = pb.z; pbv[4 * k + 3] = pb.w;
                }
                if constexpr (ND % 4 >= 2) {
                    const uint2 pa = *reinterpret_cast<const uint2 *>(rowA + ND / 4);
                    const uint2 pb = *reinterpret_cast<const uint2 *>(rowB + ND / 4);
                    pav[(ND / 4) * 4] = pa.x; pav[(ND / 4) * 4 + 1] = pa.y;
                    pbv[(ND / 4) * 4] = pb.x; pbv[(ND / 4) * 4 + 1] = pb.y;
                }
                if constexpr (ND % 2 == 1) {
                    pav[ND - 1] = reinterpret_cast<const unsigned *>(rowA)[ND - 1];
                    pbv[ND - 1] = reinterpret_cast<const unsigned *>(rowB)[ND - 1];
                }
                // Phase 1 (no dependencies between rows): everything that only needs the previous column -
                // diagonal + score, max with E, and E - ge.  Phase 2 is the serial F chain down the strip
                // (max -> sub -> max per row) with the off-chain ops of the row (f - ge, column maximum, E
                // update) available to sit behind each dependent VOP3P op.  Compared with the single-loop form
                // the compiler needs 23 fewer register moves and 10 fewer wait states per column at R = 24
                // (346 -> 313 instructions); pinning the order with sched_barrier / asm anchors was tried and
                // costs more wait states than it saves.
                unsigned pre[R], es[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // (score of target A's letter, score of target B's letter) for query row r
                    unsigned P = __builtin_amdgcn_perm(pbv[r / 2], pav[r / 2], (r & 1) ? 0x07060302u : 0x05040100u);
                    // v_bitop3 (a ? b : c, table 0xCA) as an intrinsic: written as and/or the compiler hoists NEG2 & ~mask
                    // out of the column loop - R more live registers
                    if (REV) P = __builtin_amdgcn_bitop3_b32(aux[r], P, NEG2, 0xCA);
                    pre[r] = pk_max_s(pk_add_sat(r == 0 ? hd : Hp[r - 1], P), E[r]);
                    es[r] = pk_sub_sat_u(E[r], ge2);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned h = pk_max_s(pre[r], f);
                    const unsigned fs = pk_sub_sat_u(f, ge2);
                    const unsigned t = pk_sub_sat_u(h, go2);
                    cmax = pk_max_u(cmax, h);
                    f = pk_max_u(fs, t);
                    E[r] = pk_max_u(es[r], t);
                    Hp[r] = h;
                }
                out_H = Hp[R - 1];
                out_F = f;
                out_let = let;

                const int col = s - g;
                if (has_below && g == GROUP - 1 && col >= 0 && col < ncols) scr_out[col] = make_uint2(out_H, out_F);

                // ---- running maximum / column / snapshot (rarely taken once the scores have grown) ----
                const unsigned nm = pk_max_u(vmax, cmax);
                if (nm != vmax) {
                    const unsigned diff = nm ^ vmax;
                    if (!REV) {
                        const unsigned mask = ((diff & 0xFFFFu) ? 0xFFFFu : 0u) | ((diff >> 16) ? 0xFFFF0000u : 0u);
                        bestcol = bfi(mask, ((unsigned)col & 0xFFFFu) * 0x10001u, bestcol);
#pragma unroll
                        for (int r = 0; r < R; ++r) aux[r] = bfi(mask, Hp[r], aux[r]);
                    } else {
                        // Reverse scan: nothing in q[0..q_end] x t[0..t_end] scores above the forward score, and the
                        // reference stops in the first column whose maximum equals it (terminate, :232-248 / :414-425 with
                        // :1159-1175): the column in which this strip's maximum rises TO that score is the strip's candidate,
                        // its first row holding it the row - found once per pair, no snapshot of the strip is kept.
                        const unsigned x = nm ^ rev_target;
                        const bool hitA = (diff & 0xFFFFu) && !(x & 0xFFFFu), hitB = (diff >> 16) && !(x >> 16);
                        if (hitA | hitB) {
                            asm volatile("" ::: "memory");   // a real branch: if-converted, the row search below runs in every column
                            unsigned ra = 0, rb = 0;
#pragma unroll
                            for (int r = R - 1; r >= 0; --r) {
                                if ((Hp[r] & 0xFFFFu) == (rev_target & 0xFFFFu)) ra = (unsigned)r;
                                if ((Hp[r] >> 16) == (rev_target >> 16)) rb = (unsigned)r;
                            }
                            if (hitA) { bestcol = (bestcol & 0xFFFF0000u) | ((unsigned)col & 0xFFFFu); rev_rowA = ra; }
                            if (hitB) { bestcol = (bestcol & 0xFFFFu) | ((unsigned)col << 16); rev_rowB = rb; }
                        }
                    }
                    vmax = nm;
                }
            }
            {   // the next four columns' letters; the load after that is in flight for four steps
                wA = wA_next; wB = wB_next;
                const int k = (s0 >> 2) + 2;
                wA_next = letters4(pA, endA, k); wB_next = letters4(pB, endB, k);
            }
            }

            // ---- fold this tile's lanes into the pair's running best (tie rules in the key order) ----
            unsigned rowA_first = rev_rowA, rowB_first = rev_rowB;
            unsigned sA = vmax & 0xFFFFu, sB = vmax >> 16;
            if (!REV) {
#pragma unroll
                for (int r = R - 1; r >= 0; --r) {
                    if ((aux[r] & 0xFFFFu) == sA) rowA_first = (unsigned)r;
                    if ((aux[r] >> 16) == sB) rowB_first = (unsigned)r;
                }
            } else {   // a strip that never reached the forward score has no candidate
                sA = sA == (rev_target & 0xFFFFu) ? sA : 0u;
                sB = sB == (rev_target >> 16) ? sB : 0u;
            }
            const unsigned rgA = (unsigned)(tile_base + g * R) + rowA_first;
            const unsigned rgB = (unsigned)(tile_base + g * R) + rowB_first;
            unsigned long long keyA = sA ? ((unsigned long long)sA << 32) | ((0xFFFFu - (bestcol & 0xFFFFu)) << 16) | (0xFFFFu - rgA) : 0ull;
            unsigned long long keyB = sB ? ((unsigned long long)sB << 32) | ((0xFFFFu - (bestcol >> 16)) << 16) | (0xFFFFu - rgB) : 0ull;
#pragma unroll
            for (int m = 1; m < GROUP; m <<= 1) {
                const unsigned long long oa = __shfl_xor(keyA, m), ob = __shfl_xor(keyB, m);
                keyA = oa > keyA ? oa : keyA;
                keyB = ob > keyB ? ob : keyB;
            }
            bestA = keyA > bestA ? keyA : bestA;
            bestB = keyB > bestB ? keyB : bestB;
        }

        // ---- write results -----------------------------------------------------------------------
        if (g == 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bool valid = half ? vB : vA;
                if (!valid) continue;
                const unsigned long long key = half ? bestB : bestA;
                const uint32_t h = half ? hB : hA;
                const int score = (int)(key >> 32);
                const int col = 0xFFFF - (int)((key >> 16) & 0xFFFFu);
                const int row = 0xFFFF - (int)(key & 0xFFFFu);
                mmgpu_sw_hit *o = L.out + L.hit_out[h];
                if (!REV) {
                    mmgpu_sw_hit res;
                    res.score = score;
                    res.q_end = score ? row : 0;       // all-zero saved column matches at index 0 (:263-271)
                    res.t_end = score ? col : -1;      // byte pass initialises end_db = -1 (:118)
                    res.q_start = -1;
                    res.t_start = -1;
                    res.word = (score + qbias >= 255) ? 1 : 0;
                    *o = res;
                } else {
                    const int end = half ? endB : endA;
                    const int cols = half ? colsB : colsA;
                    if (cols > 0) {
                        // column index counts backwards from t_end; row is an index into the reversed query
                        // score 0: no strip reached the forward score - "Score of forward/backward SW differ" (:1191-1201)
                        o->t_start = (score == o->score) ? end - col : -2;
                        o->q_start = (score == o->score) ? qlen - 1 - row : -2;
                    }
                }
            }
        }
    }
}

// Three launches per pass for the whole batch.  The tile shape (rows per lane R, single / multi tile) is a property
// of the job: every shape is its own instantiation of sw_body, picked per workgroup.  A launch per shape (32 kernels
// with grids from 50 to 3000 workgroups on side streams) left the chip underfilled - the runtime maps streams onto
// four hardware queues, and the long multi-tile jobs of the few long queries ran nearly alone at the end.  Shapes
// are grouped by register need instead (a kernel gets the registers of its largest body, which sets the occupancy of
// all of them): S = R <= 12 (forward 80 / reverse 119 VGPRs), M = R 14..24 (140 / 212), L = R >= 26 and every
// multi-tile shape (212 / 256).  Each group is one grid, longest job first; the three run concurrently.
// BOTH = false: forward scan only (MMGPU_SW_SCORE_END).  BOTH = true: the workgroup runs the reverse scan of its own
// pairs right after their forward scan - no grid-wide barrier between the passes, so the batch has one tail, not two.
template <int R, bool MULTI, bool BOTH>
__device__ __forceinline__ void sw_passes(const SwLaunch &L, const SwJob &job) {
    if constexpr (R <= SW_MAX_R) {
        if (!(BOTH && L.rev_only)) sw_body<R, MULTI, false>(L, job);
        // multi-tile queries get their reverse scan from sw_rev_multi_kernel (below), per query instead of per job
        if constexpr (BOTH && !MULTI) {
            __threadfence_block();   // the forward results of this job, written by other waves of the workgroup
            __syncthreads();
            sw_body<R, MULTI, true>(L, job);
        }
    }
}

// occupancy floor (waves per SIMD; 512 registers per lane and SIMD) per group and kernel kind: what the compiler is told to
// fit.  Measured on the 10k x 1M lists (profiles/r03_exp_sw_occupancy_floors.txt): the floor changes the schedule even
// where the register count already fits.
#ifndef MMGPU_SW_WAVES_G0F
#define MMGPU_SW_WAVES_G0F 4
#endif
#ifndef MMGPU_SW_WAVES_G0B
#define MMGPU_SW_WAVES_G0B 2
#endif
#ifndef MMGPU_SW_WAVES_G1F
#define MMGPU_SW_WAVES_G1F 3
#endif
#ifndef MMGPU_SW_WAVES_G1B
#define MMGPU_SW_WAVES_G1B 2
#endif
// (a query cut into tiles gets at least ceil(SW_MAX_R / 2) rows per lane: R = 8 .. 32 covers every SW_MAX_R the header allows)
#define MMGPU_SW_MULTI_ALL                                                                                                          \
    MMGPU_SW_MULTI(8) MMGPU_SW_MULTI(9) MMGPU_SW_MULTI(10) MMGPU_SW_MULTI(11) MMGPU_SW_MULTI(12) MMGPU_SW_MULTI(13) MMGPU_SW_MULTI(14) \
    MMGPU_SW_MULTI(15) MMGPU_SW_MULTI(16) MMGPU_SW_MULTI(17) MMGPU_SW_MULTI(18) MMGPU_SW_MULTI(19) MMGPU_SW_MULTI(20) MMGPU_SW_MULTI(21) \
    MMGPU_SW_MULTI(22) MMGPU_SW_MULTI(23) MMGPU_SW_MULTI(24) MMGPU_SW_MULTI(25) MMGPU_SW_MULTI(26) MMGPU_SW_MULTI(27) MMGPU_SW_MULTI(28) \
    MMGPU_SW_MULTI(29) MMGPU_SW_MULTI(30) MMGPU_SW_MULTI(31) MMGPU_SW_MULTI(32)
template <int G, bool BOTH>
__global__ __launch_bounds__(WAVES * 64)
__attribute__((amdgpu_waves_per_eu(G == 0 ? (BOTH ? MMGPU_SW_WAVES_G0B : MMGPU_SW_WAVES_G0F)
                                           : (G == 1 ? (BOTH ? MMGPU_SW_WAVES_G1B : MMGPU_SW_WAVES_G1F) : SW_MIN_WAVES),
                                   // upper bound: the forward + reverse kernel of the middle group is at the edge of a third wavefront per
                                   // SIMD (168 - 170 registers) and runs slower with it (83.4 against 78.7 ms for the stage, round 6) - pinned
                                   G == 1 && BOTH ? MMGPU_SW_WAVES_G1B : 8))) void sw_kernel(SwLaunch L) {
    SwJob job = L.jobs[blockIdx.x];
    if (L.q_hit_count) {   // fused prefilter -> align hand-over: the list length of the query is only known on the device
        const uint32_t lim = job.query * L.hit_stride + L.q_hit_count[job.query];
        job.hit_end = job.hit_end < lim ? job.hit_end : lim;
        if (job.hit_end <= job.hit_begin) return;
    }
    // multi-tile jobs (group 2 only): claim a slot of the column-scratch pool.  The pool has at least as many slots as
    // workgroups of this kernel can be resident at once, so the search always ends; a slot is only ever read after its
    // owner wrote it (tile t writes what tile t + 1 reads), so stale content is harmless.
    __shared__ uint32_t scratch_slot;
    const bool claims = G == SW_GROUPS - 1 && (job.shape & 0xFFu) >= 32u;
    if (claims) {
        if (threadIdx.x == 0) {
            uint32_t s = blockIdx.x % L.scratch_slots;
            while (atomicCAS(&L.scratch_busy[s], 0u, 1u) != 0u) s = s + 1 == L.scratch_slots ? 0u : s + 1;
            scratch_slot = s;
        }
        __syncthreads();
        job.shape = (job.shape & 0xFFu) | (scratch_slot << 8);
    }
#define MMGPU_SW_SINGLE(R) case (R) - 1: sw_passes<R, false, BOTH>(L, job); break;
#define MMGPU_SW_MULTI(R) case 32 + (R) - 1: sw_passes<R, true, BOTH>(L, job); break;
    if constexpr (G == 0) {
        switch (job.shape & 0xFFu) {   // workgroup-uniform
            MMGPU_SW_SINGLE(1) MMGPU_SW_SINGLE(2) MMGPU_SW_SINGLE(3) MMGPU_SW_SINGLE(4) MMGPU_SW_SINGLE(5) MMGPU_SW_SINGLE(6)
            MMGPU_SW_SINGLE(7) MMGPU_SW_SINGLE(8) MMGPU_SW_SINGLE(9) MMGPU_SW_SINGLE(10) MMGPU_SW_SINGLE(11) MMGPU_SW_SINGLE(12)
            default: break;
        }
    } else if constexpr (G == 1) {
        switch (job.shape & 0xFFu) {
            MMGPU_SW_SINGLE(13) MMGPU_SW_SINGLE(14) MMGPU_SW_SINGLE(15) MMGPU_SW_SINGLE(16) MMGPU_SW_SINGLE(17) MMGPU_SW_SINGLE(18)
            MMGPU_SW_SINGLE(19) MMGPU_SW_SINGLE(20) MMGPU_SW_SINGLE(21) MMGPU_SW_SINGLE(22) MMGPU_SW_SINGLE(23) MMGPU_SW_SINGLE(24)
            default: break;
        }
    } else if constexpr (SW_GROUPS == 4 && G == 2) {
        switch (job.shape & 0xFFu) {
            MMGPU_SW_SINGLE(25) MMGPU_SW_SINGLE(26) MMGPU_SW_SINGLE(27) MMGPU_SW_SINGLE(28) MMGPU_SW_SINGLE(29) MMGPU_SW_SINGLE(30)
            MMGPU_SW_SINGLE(31) MMGPU_SW_SINGLE(32)
            default: break;
        }
    } else if constexpr (SW_GROUPS == 4) {
        switch (job.shape & 0xFFu) {
            MMGPU_SW_MULTI_ALL
            default: break;
        }
    } else {
        switch (job.shape & 0xFFu) {
            MMGPU_SW_SINGLE(25) MMGPU_SW_SINGLE(26) MMGPU_SW_SINGLE(27) MMGPU_SW_SINGLE(28) MMGPU_SW_SINGLE(29) MMGPU_SW_SINGLE(30)
            MMGPU_SW_SINGLE(31) MMGPU_SW_SINGLE(32)
            MMGPU_SW_MULTI_ALL
            default: break;
        }
    }
#undef MMGPU_SW_SINGLE
#undef MMGPU_SW_MULTI
#undef MMGPU_SW_MULTI_ALL
    if (claims) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicExch(&L.scratch_busy[scratch_slot], 0u);
        }
    }
}

// Reverse scan of the multi-tile queries.  A forward job of a long query holds few hits (8 or 16 per workgroup for
// queries of 1024 residues or more, so that the batch has no long tail), of which one in six needs a reverse scan: done
// inside the forward job, one or two 16-lane groups of one wave worked while the workgroup held four waves' registers
// and a profile (measured: +19 ms on the 10k x 1M lists for 11 % of the pairs).  Here a job is up to SW_REV_JOB_HITS
// hits of one query - normally its whole list - whose live pairs are packed before they are dealt to the waves.
// Launched behind the group-2 kernel on the same stream (it reads that kernel's forward results).
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(SW_MIN_WAVES))) void sw_rev_multi_kernel(SwLaunch L) {
    SwJob job = L.jobs[blockIdx.x];
    if (L.q_hit_count) {
        const uint32_t lim = job.query * L.hit_stride + L.q_hit_count[job.query];
        job.hit_end = job.hit_end < lim ? job.hit_end : lim;
        if (job.hit_end <= job.hit_begin) return;
    }
    __shared__ uint32_t scratch_slot;
    if (threadIdx.x == 0) {
        uint32_t s = blockIdx.x % L.scratch_slots;
        while (atomicCAS(&L.scratch_busy[s], 0u, 1u) != 0u) s = s + 1 == L.scratch_slots ? 0u : s + 1;
        scratch_slot = s;
    }
    __syncthreads();
    job.shape = (job.shape & 0xFFu) | (scratch_slot << 8);
#define MMGPU_SW_REV(R) case 32 + (R) - 1: if constexpr ((R) <= SW_MAX_R) sw_body<R, true, true>(L, job); break;
    switch (job.shape & 0xFFu) {
        MMGPU_SW_REV(8) MMGPU_SW_REV(9) MMGPU_SW_REV(10) MMGPU_SW_REV(11) MMGPU_SW_REV(12) MMGPU_SW_REV(13) MMGPU_SW_REV(14) MMGPU_SW_REV(15)
        MMGPU_SW_REV(16) MMGPU_SW_REV(17) MMGPU_SW_REV(18) MMGPU_SW_REV(19) MMGPU_SW_REV(20) MMGPU_SW_REV(21) MMGPU_SW_REV(22) MMGPU_SW_REV(23)
        MMGPU_SW_REV(24) MMGPU_SW_REV(25) MMGPU_SW_REV(26) MMGPU_SW_REV(27) MMGPU_SW_REV(28) MMGPU_SW_REV(29) MMGPU_SW_REV(30) MMGPU_SW_REV(31)
        MMGPU_SW_REV(32)
        default: break;
    }
#undef MMGPU_SW_REV
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicExch(&L.scratch_busy[scratch_slot], 0u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fused prefilter -> align hand-over: what mmgpu_sw_prepare does on the host for caller
// supplied lists, done on the device for the hit lists of a prefilter batch.  One workgroup per query: the list is
// sorted by target length (longest first, stable) so that the targets a wavefront runs together end together;
// hit_target / hit_out are written for the list's slots [q * stride, q * stride + count).
__global__ __launch_bounds__(256) void sw_from_pf_kernel(SwFromPfArgs A) {
    extern __shared__ uint32_t key[];      // next power of two >= the lists' stride (launch_sw_from_pf)
    const uint32_t q = blockIdx.x;
    unsigned long long *const stat = A.cells + (size_t)(blockIdx.x % (uint32_t)SW_FROM_PF_STAT_SLOTS) * 3;      // (cells, pairs, longest target)
    const uint32_t n = min(A.hit_count[q], A.stride);
    const mmgpu_pf_hit *hits = A.pf_hits + (size_t)q * A.pf_stride;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    unsigned long long cells = 0;
    const uint32_t qlen = A.q_off[q + 1] - A.q_off[q];
    for (uint32_t k = threadIdx.x; k < np2; k += 256) {
        uint32_t v = 0xFFFFFFFFu;
        if (k < n) {
            const uint32_t len = A.t_len[hits[k].id];
            v = ((0xFFFFu - min(len, 0xFFFFu)) << 16) | k;      // length descending, list position ascending
            cells += (unsigned long long)qlen * len;
        }
        key[k] = v;
    }
    __syncthreads();
    const uint32_t base = q * A.stride;
    if (n <= 512) {
        // a list of --max-seqs 300: every element counts the keys below its own (the keys are distinct; all lanes read the same
        // word of LDS at a time) - one pass instead of the 45 barrier-separated steps of the network below
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint32_t mine = key[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < n; j++) r += key[j] < mine ? 1u : 0u;
            A.hit_target[base + r] = hits[i].id;
            A.hit_out[base + r] = base + i;
            A.slot_target[base + i] = hits[i].id;
            if (r == 0) atomicMax(stat + 2, (unsigned long long)(0xFFFFu - (mine >> 16)));   // longest target of any list
        }
    } else {
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < np2 / 2; k += 256) {
                const uint32_t i = 2 * k - (k & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint32_t a = key[i], b = key[j];
                if ((a > b) == up) {
                    key[i] = b;
                    key[j] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t r = threadIdx.x; r < n; r += 256) {
        const uint32_t k = key[r] & 0xFFFFu;
        A.hit_target[base + r] = hits[k].id;
        A.hit_out[base + r] = base + k;
        A.slot_target[base + r] = hits[r].id;
    }
    if (threadIdx.x == 0 && n) atomicMax(stat + 2, (unsigned long long)(0xFFFFu - (key[0] >> 16)));   // longest target of any list
    }
    for (uint32_t r = n + threadIdx.x; r < A.stride; r += 256) {   // unused slots: defined contents
        A.hit_target[base + r] = 0;
        A.hit_out[base + r] = base + r;
        A.slot_target[base + r] = 0;
    }
    if (threadIdx.x == 0) A.count_copy[q] = n;
    // statistics (cells = forward DP cells, Alignment.cpp:380,530 convention)
    for (int d = 1; d < 64; d <<= 1) cells += __shfl_xor(cells, d);
    if ((threadIdx.x & 63u) == 0 && cells) atomicAdd(stat, cells);
    if (threadIdx.x == 0 && n) atomicAdd(stat + 1, (unsigned long long)n);
}

}  // namespace

hipError_t launch_sw_from_pf(const SwFromPfArgs &A, uint32_t nq, hipStream_t stream) {
    if (nq == 0) return hipSuccess;
    uint32_t np2 = 1;
    while (np2 < A.stride) np2 <<= 1;
    hipLaunchKernelGGL(sw_from_pf_kernel, dim3(nq), dim3(256), (size_t)np2 * sizeof(uint32_t), stream, A);
    return hipGetLastError();
}

uint32_t sw_multi_resident_blocks(size_t lds_bytes, bool both_passes, int compute_units) {
    int per_cu = 0;
    const size_t lds = lds_bytes + SW_LDS_HEADER;
    hipError_t e = both_passes ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sw_kernel<SW_GROUPS - 1, true>, WAVES * 64, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sw_kernel<SW_GROUPS - 1, false>, WAVES * 64, lds);
    if (e == hipSuccess && both_passes) {   // the reverse kernel of the multi-tile queries draws on the same pool
        int rev = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&rev, sw_rev_multi_kernel, WAVES * 64, lds) == hipSuccess) per_cu = std::max(per_cu, rev);
    }
    // a CU holds at most 32 wavefronts = 8 workgroups of 4; one more than the occupancy query says, as a margin
    if (e != hipSuccess || per_cu < 1) per_cu = 8;
    per_cu = per_cu + 1 > 8 ? 8 : per_cu + 1;
    return (uint32_t)per_cu * (uint32_t)(compute_units > 0 ? compute_units : 256);
}

size_t sw_lds_bytes(int rows_per_lane, int alphabet) {
    return (size_t)(alphabet + 1) * GROUP * lane_stride_bytes(rows_per_lane);
}

hipError_t launch_sw_rev_multi(const SwLaunch &L, size_t lds_bytes, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_rev_multi_kernel, dim3(L.n_jobs), dim3(WAVES * 64), lds_bytes + SW_LDS_HEADER, stream, L);
    return hipGetLastError();
}

int sw_shape_group(uint32_t shape) {
    if (shape >= 32) return SW_GROUPS - 1;
    const int R = (int)shape + 1;
    return R <= 12 ? 0 : (R <= 24 ? 1 : 2);
}

hipError_t launch_sw(const SwLaunch &L, int group, size_t lds_bytes, bool both_passes, hipStream_t stream) {
    static_assert(GROUP == 16, "the shape codes assume 16-lane groups");
    if (L.n_jobs == 0) return hipSuccess;
    dim3 grid(L.n_jobs), block(WAVES * 64);
    const size_t lds = lds_bytes + SW_LDS_HEADER;
    switch (group * 2 + (both_passes ? 1 : 0)) {
        case 0: hipLaunchKernelGGL((sw_kernel<0, false>), grid, block, lds, stream, L); break;
        case 1: hipLaunchKernelGGL((sw_kernel<0, true>), grid, block, lds, stream, L); break;
        case 2: hipLaunchKernelGGL((sw_kernel<1, false>), grid, block, lds, stream, L); break;
        case 3: hipLaunchKernelGGL((sw_kernel<1, true>), grid, block, lds, stream, L); break;
        case 4: hipLaunchKernelGGL((sw_kernel<2, false>), grid, block, lds, stream, L); break;
        case 5: hipLaunchKernelGGL((sw_kernel<2, true>), grid, block, lds, stream, L); break;
        case 6: hipLaunchKernelGGL((sw_kernel<SW_GROUPS - 1, false>), grid, block, lds, stream, L); break;      // (SW_GROUPS == 4 only)
        case 7: hipLaunchKernelGGL((sw_kernel<SW_GROUPS - 1, true>), grid, block, lds, stream, L); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// first use of any kernel of this file loads its code object (tens of milliseconds): mmgpu_warmup does it ahead of time
void warm_sw() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_rev_multi_kernel));
}

}  // namespace mmgpu
