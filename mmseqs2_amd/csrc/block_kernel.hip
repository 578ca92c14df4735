// Start position + backtrace of int16-range hits (s_align::word == 1) on the device: the block aligner as the reference calls
// it - SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ> (src/alignment/StripedSmithWaterman.cpp:943-1127) ->
// Block<TRACE, X_DROP>::align_aa of the vendored lib/block-aligner 0.4.0 in its AVX2 configuration (16 int16 lanes, ZERO = 1 << 14,
// MIN = 0; scan_block.rs:120-632,1016-1052,1449-1613, avx2.rs) - SURVEY.md section 8 row a15.  oracle/block_oracle.c is the plain-C
// restatement this kernel is tested against, vector operation by vector operation.
//
// Mapping: ONE WAVEFRONT per pair.  The crate walks a column of the block in chunks of 16 rows (one AVX2 vector) and carries
// three things from chunk to chunk: the last D of the previous chunk (D00's first lane), the last R (`R01`, broadcast + the
// 1..16 x gap_extend ladder) and the last trace_R flag.  A wavefront takes FOUR chunks (64 rows) per step: lane l = row
// i0 + l, the in-vector shifts are lane shifts confined to 8 / 16 lanes exactly as the byte shifts of the 128-bit halves are
// (zero fill included - the crate "relies on min score = 0"), the chunk-to-chunk carries are resolved with three readlanes.
// Same operations in the same order on the same int16 values (saturating adds), so the result - score, end position,
// every trace bit - equals the crate's; the block's border arrays (D_col, C_col, D_row, R_row, their checkpoints) live in LDS,
// the trace (2 x 2 bits per cell, as ballots) and the block list in a scratch slot in HBM, the walk back is serial.
// Three instantiations: blocks up to 512 rows with the border arrays in LDS (8 KB; every pair goes here first, with a scratch
// slot sized for the usual case), up to 2048 rows in LDS (32 KB) and the crate's full 4096 rows with the border arrays in the
// pair's scratch slot in HBM - later launches only for the pairs the one before answered MMGPU_BLOCK_TOO_LARGE (the crate would
// have grown the block further, the score was not reached with the minimum sizes this instantiation tries, or the slot overflowed).
// A single wavefront owns a slot, its stores and loads go through one L1 in program order, so the HBM borders need no fence
// beyond the wave barriers the LDS form already has.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

constexpr int BK_STEP = 8, BK_ZERO = 16384, BK_MIN = 0, BK_X_DROP_ITER = 2;

__device__ __forceinline__ int adds16(int a, int b) { const int s = a + b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }
__device__ __forceinline__ int subs16(int a, int b) { const int s = a - b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }
__device__ __forceinline__ int sext16(int a) { return (int)(short)a; }
// simd_sllz_i16!(x, n): byte shift inside each 128-bit half (8 lanes), zeros shifted in (avx2.rs:152-164)
// (cross-lane moves are DPP row shifts / readlanes, not ds_bpermute: the column loop is one long dependent chain and the
// LDS-crossbar round trip of a bpermute was most of its latency)
template <int N>
__device__ __forceinline__ int sllz8(int x, int lane) {
    const int v = __builtin_amdgcn_update_dpp(0, x, 0x110 + N /* row_shr:N */, 0xF, 0xF, true);
    return (lane & 7) >= N ? v : 0;
}
// lane 7 of the caller's 16-lane row (DPP row_newbcast: one move instead of four readlanes and three selects)
__device__ __forceinline__ int row_lane7(int v, int lane) {
    (void)lane;
    return __builtin_amdgcn_update_dpp(0, v, 0x157 /* row_newbcast:7 */, 0xF, 0xF, false);
}
// lane 15 of the row below the caller's (row 0 receives `first`): DPP row_bcast:15
__device__ __forceinline__ int prev_row_lane15(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x142 /* row_bcast:15 */, 0xE, 0xF, false);
}
// lane - 1's value; lane 0 receives `first`
__device__ __forceinline__ int shift_up1(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

struct BkConsts { int gap_all, consts; };   // per lane (lane & 15): avx2.rs:294-309

__device__ __forceinline__ BkConsts bk_consts(int g, int lane) {
    const int s1 = adds16(sllz8<1>(g, lane), g);
    const int s2 = adds16(sllz8<2>(s1, lane), s1);
    const int s4 = adds16(sllz8<4>(s2, lane), s2);
    const int w7 = row_lane7(s4, lane);
    BkConsts c;
    c.gap_all = adds16((lane & 15) < 8 ? 0 : w7, s4);
    c.consts = s4;
    return c;
}

// simd_prefix_scan_i16 (avx2.rs:311-337) on every 16-lane row of the wavefront
__device__ __forceinline__ int bk_prefix_scan(int R, int g, int consts, int lane) {
    const int s1 = max(R, adds16(sllz8<1>(R, lane), g));
    const int s2 = max(s1, adds16(sllz8<2>(s1, lane), sext16(g << 1)));
    const int s4 = max(s2, adds16(sllz8<4>(s2, lane), sext16(g << 2)));
    const int k = lane & 15;
    // source lane: itself (k < 4), four lanes down (k < 8), lane 7 of the row (k >= 8)
    const int down4 = __builtin_amdgcn_update_dpp(0, s4, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    const int l7 = row_lane7(s4, lane);      // (outside the selection: a DPP move reads nothing from lanes that are switched off)
    const int from = k < 4 ? s4 : (k < 8 ? down4 : l7);
    const int c1 = adds16(from, consts);
    return max(s4, c1);
}

struct BkSeq {
    const uint8_t *res;   // forward sequence; reversed prefix position p (1-based DP index) is res[end - (p - 1)]
    const int8_t *bias;   // per-position bias of the forward sequence (nullptr = 0)
    int end;              // forward index of reversed position 1
    int len;              // reversed prefix length (end + 1)
    // Profile query (alignStartPosBacktraceBlock<PROFILE_SEQ>, StripedSmithWaterman.cpp:963-990): this "sequence" is the AAProfile
    // the reference builds from the reversed prefix of the query's score rows - prof = int8 [alphabet][plen] (ssw_init's
    // profile->mat, the X row neutral), its elements are POSITIONS, the score of a cell is the row of the other sequence's letter
    // at that position; letters beyond the alphabet and padding positions score i8::MIN (AAProfile::new, scores.rs:494-507).
    // With the gap costs the reference sets (every opening gap_open, closing 0, :988-990) place_block_profile (scan_block.rs:
    // 649-815) is place_block_aa with this score: one kernel body serves both.  res = the consensus sequence (identities, :1076).
    const int8_t *prof;
    int plen, alphabet;
};
// PaddedBytes::get / PosBias::get at DP index p: index 0 and everything past the end is the padding letter / bias 0
__device__ __forceinline__ int bk_letter(const BkSeq &s, int p) { return (p >= 1 && p <= s.len) ? (int)s.res[s.end - (p - 1)] : 26; }
__device__ __forceinline__ int bk_bias(const BkSeq &s, int p) { return (s.bias && p >= 1 && p <= s.len) ? (int)s.bias[s.end - (p - 1)] : 0; }
// a profile's element at DP index p: the forward position, -1 for the padding in front and behind
__device__ __forceinline__ int bk_pos(const BkSeq &s, int p) { return (p >= 1 && p <= s.len) ? s.end - (p - 1) : -1; }
__device__ __forceinline__ int bk_prof_score(const BkSeq &pr, int pos, int letter) {
    return (pos >= 0 && letter < pr.alphabet) ? (int)pr.prof[(size_t)letter * pr.plen + pos] : -128;
}


struct BkState {
    int16_t *D_col, *C_col, *D_row, *R_row, *D_col_ck, *C_col_ck, *D_row_ck, *R_row_ck, *temp1, *temp2;
    unsigned long long *trace;      // [entries][4]: D == C, D == R, C == C_open, trace_R (ballots over the 64 rows of a step)
    BkBlock *blocks;
    uint32_t trace_idx, block_idx, ck_trace_idx, ck_block_idx, trace_cap, block_cap;
    bool overflow;
};

struct BkMax { int dmax, ai, aj; };   // per lane: running maximum of its rows, chunk base / column of its LAST occurrence

// place_block_aa (scan_block.rs:1449-1613).  rows run over `query`, columns over `reference` (a down shift swaps the roles).
__device__ __forceinline__ BkMax bk_place_block(const BlockLaunch &L, const int8_t *scores, BkState &S, const BkSeq &query, const BkSeq &reference,
                                                int start_i, int start_j, int width, int height, int16_t *D_col, int16_t *C_col,
                                                int16_t *D_row, int16_t *R_row, int D_corner, const BkConsts &K, int lane) {
    const int go = L.gap_open, ge = L.gap_extend;
    BkMax M;
    M.dmax = BK_MIN; M.ai = 0; M.aj = 0;
    if (width == 0 || height == 0) return M;
    const int iters = (height + 63) >> 6;
    const bool rows_prof = query.prof != nullptr, cols_prof = reference.prof != nullptr;      // wave-uniform; at most one of them
    // blocks of up to 64 rows (nearly all): a lane's row - its letter (position, for a profile) and bias - is the same for every column
    const int row_letter0 = rows_prof ? bk_pos(query, start_i + lane) : bk_letter(query, start_i + lane), row_bias0 = bk_bias(query, start_i + lane);
    const int row = lane >> 4;
    for (int j = 0; j < width; j++) {
        const int c = cols_prof ? bk_pos(reference, start_j + j) : bk_letter(reference, start_j + j);
        const int rbias = bk_bias(reference, start_j + j);
        int carryR = BK_MIN, corner = D_corner, carry_tr = 0;
        int D11 = BK_MIN, R11 = BK_MIN;
        for (int it = 0; it < iters; it++) {
            const int i = (it << 6) + lane;
            const bool act = i < height;
            const bool more = it + 1 < iters;      // (uniform) another 64-row step of this column follows: it needs the carries
            const int D10 = act ? (int)D_col[i] : BK_MIN, C10 = act ? (int)C_col[i] : BK_MIN;
            const int D00 = shift_up1(D10, corner);
            const int last = min(63, height - 1 - (it << 6));
            if (more) corner = __builtin_amdgcn_readlane(D10, 63);
            const int ql = iters == 1 ? row_letter0 : (rows_prof ? bk_pos(query, start_i + i) : bk_letter(query, start_i + i));
            int sc;
            if (cols_prof) sc = bk_prof_score(reference, c, ql);            // get_scores_pos: the column's position, the rows' letters
            else if (rows_prof) sc = bk_prof_score(query, ql, c);           // get_scores_aa: the column's letter, the rows' positions
            else sc = (int)scores[c * 32 + (ql & 31)];
            const int pos_bias = adds16(rbias, iters == 1 ? row_bias0 : bk_bias(query, start_i + i));
            D11 = adds16(D00, adds16(sc, pos_bias));
            if (start_i + i == 0 && start_j + j == 0) D11 = BK_ZERO;
            const int C11_open = adds16(D10, go);
            const int C11 = max(adds16(C10, ge), C11_open);
            D11 = max(D11, C11);
            const int D11_open = adds16(D11, subs16(go, ge));
            R11 = bk_prefix_scan(D11_open, ge, K.consts, lane);
            // R11 = max(R11, broadcasthi(R01) + gap_extend_all), chunk by chunk (R01 = the previous chunk's R11): row 0 takes the
            // carry of the step before, rows 1..3 the last lane of the row below, one after the other
            {
                const int add = adds16(carryR, K.gap_all);
                if (row == 0) R11 = max(R11, add);
            }
#pragma unroll
            for (int ch = 1; ch < 4; ch++) {
                const int add = adds16(prev_row_lane15(R11, BK_MIN), K.gap_all);
                if (row == ch) R11 = max(R11, add);
            }
            // the carry into the next step: the last lane of the last chunk that exists (every chunk of a step but the column's last does)
            if (more) carryR = __builtin_amdgcn_readlane(R11, 63);
            D11 = max(D11, R11);
            const bool tempR = R11 == D11_open;
            const int trR = shift_up1(tempR ? 1 : 0, carry_tr);
            if (more) carry_tr = __builtin_amdgcn_readlane(tempR ? 1 : 0, 63);
            (void)last;
            const unsigned long long bDC = __ballot(act && D11 == C11), bDR = __ballot(act && D11 == R11);
            const unsigned long long bCO = __ballot(act && C11 == C11_open), bTR = __ballot(act && trR != 0);
            if (S.trace_idx >= S.trace_cap) S.overflow = true;
            else if (lane == 0) {
                unsigned long long *t = S.trace + (size_t)S.trace_idx * 4;
                t[0] = bDC; t[1] = bDR; t[2] = bCO; t[3] = bTR;
            }
            S.trace_idx++;
            if (act) {
                M.dmax = max(M.dmax, D11);
                if (M.dmax == D11) { M.ai = (it << 6) + (lane & ~15); M.aj = j; }
                D_col[i] = (int16_t)D11;
                C_col[i] = (int16_t)C11;
            }
        }
        const int lastl = (height - 1) & 63;
        const int dl = __builtin_amdgcn_readlane(D11, lastl), rl = __builtin_amdgcn_readlane(R11, lastl);
        if (lane == 0) { D_row[j] = (int16_t)dl; R_row[j] = (int16_t)rl; }
        D_corner = BK_MIN;
    }
    return M;
}

// the 16-lane D_max / D_argmax vectors of the crate from the per-lane maxima: lanes l, l + 16, l + 32, l + 48 are one vector lane;
// its arg-max is the LAST update that reached its maximum (order: column, then chunk)
__device__ __forceinline__ BkMax bk_fold_vector_lanes(BkMax m) {
#pragma unroll
    for (int d = 16; d < 64; d <<= 1) {
        BkMax o;
        o.dmax = __shfl_xor(m.dmax, d, 64); o.ai = __shfl_xor(m.ai, d, 64); o.aj = __shfl_xor(m.aj, d, 64);
        const bool take = o.dmax > m.dmax || (o.dmax == m.dmax && (o.aj > m.aj || (o.aj == m.aj && o.ai > m.ai)));
        if (take) m = o;
    }
    return m;
}
__device__ __forceinline__ int bk_wave_max(int v) {
    // inclusive max scan along every 16-lane row (lanes without a source keep their own value), then the four row ends
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112 /* row_shr:2 */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114 /* row_shr:4 */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118 /* row_shr:8 */, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

__device__ __forceinline__ void bk_add_block(BkState &S, int i, int j, int width, int height, int right, int lane) {
    if (S.block_idx >= S.block_cap) { S.overflow = true; S.block_idx++; return; }
    if (lane == 0) {
        BkBlock b;
        b.i = (uint32_t)i; b.j = (uint32_t)j; b.h = (uint16_t)height; b.w = (uint16_t)width; b.right = (uint32_t)right; b.tstart = S.trace_idx;
        S.blocks[S.block_idx] = b;
    }
    S.block_idx++;
}

__device__ __forceinline__ void bk_copy(int16_t *dst, const int16_t *src, int n, int lane) {
    for (int k = lane; k < n; k += 64) dst[k] = src[k];
}

// align_core_gen! (scan_block.rs:120-632), TRACE = X_DROP = true
__device__ void bk_align(const BlockLaunch &L, const int8_t *scores, BkState &S, const BkSeq &query, const BkSeq &reference, int min_size,
                         int max_size, int x_drop, const BkConsts &K, int lane, int *res_score, int *res_i, int *res_j) {
    enum { RIGHT, DOWN, GROW };
    int best_max = 0, best_i = 0, best_j = 0;
    int prev_dir = GROW, dir = GROW;
    int prev_size = 0, block_size = min_size;
    int off = 0, prev_off, off_max = 0;
    int y_drop_iter = 0, x_drop_iter = 0;
    int st_i = 0, st_j = 0, i_ck = 0, j_ck = 0, off_ck = 0;
    int D_corner = BK_MIN;
    const int qlen = query.len, rlen = reference.len;
    for (;;) {
        prev_off = off;
        BkMax G;     // grow_D_max
        G.dmax = BK_MIN; G.ai = 0; G.aj = 0;
        BkMax M;
        int right_max, down_max;
        auto prefix_max = [&](const int16_t *buf) { int v = lane < BK_STEP ? (int)buf[lane] : -32768; return bk_wave_max(v); };
        auto shift_and_offset = [&](int16_t *b1, int16_t *b2, int off_add) {     // :1102-1123
            const int corner = adds16((int)b1[BK_STEP - 1], off_add);
            for (int x0 = 0; x0 < block_size; x0 += 64) {
                const int x = x0 + lane;
                int v1 = 0, v2 = 0;
                if (x < block_size) {
                    if (x < block_size - BK_STEP) { v1 = adds16((int)b1[x + BK_STEP], off_add); v2 = adds16((int)b2[x + BK_STEP], off_add); }
                    else { v1 = (int)S.temp1[x - (block_size - BK_STEP)]; v2 = (int)S.temp2[x - (block_size - BK_STEP)]; }
                }
                __builtin_amdgcn_wave_barrier();      // every lane has read its source before any lane overwrites it
                if (x < block_size) { b1[x] = (int16_t)v1; b2[x] = (int16_t)v2; }
            }
            return corner;
        };
        auto just_offset = [&](int16_t *b1, int16_t *b2, int off_add) {
            for (int x = lane; x < block_size; x += 64) { b1[x] = (int16_t)adds16((int)b1[x], off_add); b2[x] = (int16_t)adds16((int)b2[x], off_add); }
        };
        if (dir == RIGHT) {
            off = off_max;
            const int d = prev_off - off;
            const int off_add = d < -32768 ? -32768 : (d > 32767 ? 32767 : d);
            bk_add_block(S, st_i, st_j + block_size - BK_STEP, BK_STEP, block_size, 1, lane);
            just_offset(S.D_col, S.C_col, off_add);
            M = bk_place_block(L, scores, S, query, reference, st_i, st_j + block_size - BK_STEP, BK_STEP, block_size, S.D_col, S.C_col, S.temp1,
                               S.temp2, prev_dir == DOWN ? adds16(D_corner, off_add) : BK_MIN, K, lane);
            right_max = prefix_max(S.D_col);
            D_corner = shift_and_offset(S.D_row, S.R_row, off_add);
            down_max = prefix_max(S.D_row);
        } else if (dir == DOWN) {
            off = off_max;
            const int d = prev_off - off;
            const int off_add = d < -32768 ? -32768 : (d > 32767 ? 32767 : d);
            bk_add_block(S, st_i + block_size - BK_STEP, st_j, block_size, BK_STEP, 0, lane);
            just_offset(S.D_row, S.R_row, off_add);
            M = bk_place_block(L, scores, S, reference, query, st_j, st_i + block_size - BK_STEP, BK_STEP, block_size, S.D_row, S.R_row, S.temp1,
                               S.temp2, prev_dir == RIGHT ? adds16(D_corner, off_add) : BK_MIN, K, lane);
            down_max = prefix_max(S.D_row);
            D_corner = shift_and_offset(S.D_col, S.C_col, off_add);
            right_max = prefix_max(S.D_col);
        } else {
            D_corner = BK_MIN;
            const int grow_step = block_size - prev_size;
            bk_add_block(S, st_i + prev_size, st_j, prev_size, grow_step, 0, lane);
            G = bk_place_block(L, scores, S, reference, query, st_j, st_i + prev_size, grow_step, prev_size, S.D_row, S.R_row, S.D_col + prev_size,
                               S.C_col + prev_size, BK_MIN, K, lane);
            bk_add_block(S, st_i, st_j + prev_size, grow_step, block_size, 1, lane);
            M = bk_place_block(L, scores, S, query, reference, st_i, st_j + prev_size, grow_step, block_size, S.D_col, S.C_col, S.D_row + prev_size,
                               S.R_row + prev_size, BK_MIN, K, lane);
            right_max = prefix_max(S.D_col);
            down_max = prefix_max(S.D_row);
            bk_copy(S.D_col_ck, S.D_col, block_size, lane); bk_copy(S.C_col_ck, S.C_col, block_size, lane);
            bk_copy(S.D_row_ck, S.D_row, block_size, lane); bk_copy(S.R_row_ck, S.R_row, block_size, lane);
            S.ck_trace_idx = S.trace_idx;
            S.ck_block_idx = S.block_idx;
        }
        if (S.overflow) break;
        prev_dir = dir;
        M = bk_fold_vector_lanes(M);
        G = bk_fold_vector_lanes(G);
        const int D_max_max = bk_wave_max(M.dmax), grow_max = bk_wave_max(G.dmax);
        const int mx = max(D_max_max, grow_max);
        off_max = off + mx - BK_ZERO;
        y_drop_iter++;
        bool grow_no_max = dir == GROW;
        if (off_max > best_max) {
            {   // location of the maximum: ties to the larger column, then the larger row (:374-444)
                const bool grow = dir == GROW && D_max_max < grow_max;
                const int curr_max = grow ? grow_max : D_max_max;
                const BkMax C = grow ? G : M;
                const int k = lane & 15;
                // (int16 -> usize: a negative chunk base cannot occur, values are < 512)
                const int r = C.ai + k, c = (block_size - BK_STEP) + C.aj;
                int gi, gj;
                if (grow) { gi = st_i + prev_size + C.aj; gj = st_j + C.ai + k; }
                else if (dir == RIGHT) { gi = st_i + r; gj = st_j + c; }
                else if (dir == DOWN) { gi = st_i + c; gj = st_j + r; }
                else { gi = st_i + C.ai + k; gj = st_j + prev_size + C.aj; }
                // best (gj, gi) over the vector lanes whose maximum is the block's; start value (0, 0) as the crate's
                long long key = (lane < 16 && C.dmax == curr_max) ? (((long long)gj << 32) | (unsigned)gi) : 0ll;
                for (int d = 1; d < 16; d <<= 1) {
                    const long long o = __shfl_xor(key, d, 64);
                    key = o > key ? o : key;
                }
                key = __shfl(key, 0, 64);
                best_j = (int)(key >> 32);
                best_i = (int)(key & 0xFFFFFFFFll);
            }
            if (block_size < BLOCK_REF_MAX_SIZE) {      // the reference's max_size (4096), not this kernel's: same checkpoints, same grow_no_max
                i_ck = st_i; j_ck = st_j; off_ck = off;
                bk_copy(S.D_col_ck, S.D_col, block_size, lane); bk_copy(S.C_col_ck, S.C_col, block_size, lane);
                bk_copy(S.D_row_ck, S.D_row, block_size, lane); bk_copy(S.R_row_ck, S.R_row, block_size, lane);
                S.ck_trace_idx = S.trace_idx;
                S.ck_block_idx = S.block_idx;
                grow_no_max = false;
            }
            best_max = off_max;
            y_drop_iter = 0;
        }
        if (off_max < best_max - x_drop) {
            if (x_drop_iter < BK_X_DROP_ITER - 1) x_drop_iter++;
            else break;
        } else {
            x_drop_iter = 0;
        }
        if (st_i + block_size > qlen && st_j + block_size > rlen) break;
        if (st_j + block_size > rlen) { st_i += BK_STEP; dir = DOWN; continue; }
        if (st_i + block_size > qlen) { st_j += BK_STEP; dir = RIGHT; continue; }
        const int next_size = block_size * 2;
        if (next_size <= max_size) {
            if (y_drop_iter > (block_size / BK_STEP) - 1 || grow_no_max) {
                prev_size = block_size;
                block_size = next_size;
                dir = GROW;
                st_i = i_ck; st_j = j_ck; off = off_ck;
                bk_copy(S.D_col, S.D_col_ck, prev_size, lane); bk_copy(S.C_col, S.C_col_ck, prev_size, lane);
                bk_copy(S.D_row, S.D_row_ck, prev_size, lane); bk_copy(S.R_row, S.R_row_ck, prev_size, lane);
                S.trace_idx = S.ck_trace_idx;
                S.block_idx = S.ck_block_idx;
                y_drop_iter = 0;
                continue;
            }
        } else if (next_size <= BLOCK_REF_MAX_SIZE && (y_drop_iter > (block_size / BK_STEP) - 1 || grow_no_max)) {
            S.overflow = true;      // the crate would grow beyond what this kernel holds: not decided here
            break;
        }
        if (block_size > min_size && y_drop_iter == 0) {      // SHRINK (:542-586)
            const int s1 = max((int)S.D_row[block_size - 1], (int)S.D_row[block_size - 2]);
            const int s2 = max((int)S.D_col[block_size - 1], (int)S.D_col[block_size - 2]);
            if (max(s1, s2) >= mx) {
                prev_dir = GROW;
                block_size /= 2;
                for (int x0 = 0; x0 < block_size; x0 += 64) {      // copy_vec(i, i + block_size)
                    const int x = x0 + lane;
                    int a = 0, b = 0, c = 0, d = 0;
                    if (x < block_size) { a = S.D_col[x + block_size]; b = S.C_col[x + block_size]; c = S.D_row[x + block_size]; d = S.R_row[x + block_size]; }
                    __builtin_amdgcn_wave_barrier();
                    if (x < block_size) { S.D_col[x] = (int16_t)a; S.C_col[x] = (int16_t)b; S.D_row[x] = (int16_t)c; S.R_row[x] = (int16_t)d; }
                }
                st_i += block_size;
                st_j += block_size;
                i_ck = st_i; j_ck = st_j; off_ck = off;
                bk_copy(S.D_col_ck, S.D_col, block_size, lane); bk_copy(S.C_col_ck, S.C_col, block_size, lane);
                bk_copy(S.D_row_ck, S.D_row, block_size, lane); bk_copy(S.R_row_ck, S.R_row, block_size, lane);
                right_max = prefix_max(S.D_col);
                down_max = prefix_max(S.D_row);
                S.ck_trace_idx = S.trace_idx;
                S.ck_block_idx = S.block_idx;
                y_drop_iter = 0;
            }
        }
        if (down_max > right_max) { st_i += BK_STEP; dir = DOWN; }
        else { st_j += BK_STEP; dir = RIGHT; }
    }
    *res_score = best_max;
    *res_i = best_i;
    *res_j = best_j;
}

template <int MAXB, bool LDS_BORDERS>
__global__ __launch_bounds__(64) void sw_block_kernel(BlockLaunch L) {
    __shared__ int16_t s_lds[LDS_BORDERS ? 8 * MAXB : 8];
    __shared__ int16_t s_temp[2][64];
    __shared__ int8_t s_scores[27 * 32];
    __shared__ uint32_t s_slot;
    const int lane = (int)threadIdx.x;
    for (int k = lane; k < 27 * 32; k += 64) s_scores[k] = L.scores[k];
    if (lane == 0) {      // scratch slot of the pool (at least as many slots as workgroups can be resident)
        uint32_t s = blockIdx.x % L.n_pool_slots;
        while (atomicCAS(&L.pool_busy[s], 0u, 1u) != 0u) s = s + 1 == L.n_pool_slots ? 0u : s + 1;
        s_slot = s;
    }
    __syncthreads();
    const BlockJob J = L.jobs[blockIdx.x];
    BkState S;
    uint8_t *slot = L.pool + (size_t)s_slot * L.slot_bytes;
    uint64_t slot_bytes = L.slot_bytes;
    int16_t *s_buf;
    if (LDS_BORDERS) {
        s_buf = s_lds;
    } else {      // the first 8 * MAXB int16 of the slot
        s_buf = reinterpret_cast<int16_t *>(slot);
        slot += (size_t)8 * MAXB * sizeof(int16_t);
        slot_bytes -= (uint64_t)8 * MAXB * sizeof(int16_t);
    }
    S.D_col = s_buf; S.C_col = s_buf + MAXB; S.D_row = s_buf + 2 * MAXB; S.R_row = s_buf + 3 * MAXB;
    S.D_col_ck = s_buf + 4 * MAXB; S.C_col_ck = s_buf + 5 * MAXB; S.D_row_ck = s_buf + 6 * MAXB; S.R_row_ck = s_buf + 7 * MAXB;
    S.temp1 = s_temp[0]; S.temp2 = s_temp[1];
    const int qa = J.q_end + 1, ta = J.t_end + 1;
    S.block_cap = (uint32_t)(qa + ta + 64);
    S.blocks = reinterpret_cast<BkBlock *>(slot);
    const size_t blocks_bytes = ((size_t)S.block_cap * sizeof(BkBlock) + 31) & ~(size_t)31;
    S.trace = reinterpret_cast<unsigned long long *>(slot + blocks_bytes);
    S.trace_cap = blocks_bytes < slot_bytes ? (uint32_t)std::min<uint64_t>((slot_bytes - blocks_bytes) / 32, 0xFFFFFFFFull) : 0u;
    BkSeq Q, T;
    Q.res = L.q_res + L.q_off[J.query];
    Q.bias = L.q_cb + L.q_off[J.query];
    Q.end = J.q_end;
    Q.len = qa;
    Q.prof = nullptr; Q.plen = 0; Q.alphabet = L.alphabet;
    if (L.q_prof_off != nullptr && L.q_prof_off[J.query] != 0xFFFFFFFFu) {      // profile query (wave-uniform)
        Q.prof = L.q_prof + L.q_prof_off[J.query];
        Q.plen = (int)(L.q_off[J.query + 1] - L.q_off[J.query]);
        Q.bias = nullptr;      // (no composition bias on profile queries)
    }
    const bool prof = Q.prof != nullptr;
    T.res = L.t_res + (size_t)L.t_off4[J.target] * 4;
    T.bias = nullptr;
    T.end = J.t_end;
    T.len = ta;
    T.prof = nullptr; T.plen = 0; T.alphabet = L.alphabet;
    const BkConsts K = bk_consts(L.gap_extend, lane);
    mmgpu_sw_block out;
    out.q_start = -1; out.t_start = -1; out.ident = 0; out.bt_len = 0; out.bt_off = L.bt_off[J.slot];
    out.status = MMGPU_BLOCK_DECLINED;
    int score = -1000000000, ri = 0, rj = 0;
    bool too_large = blocks_bytes >= slot_bytes;
    const long long t_begin = clock64();
    long long t_walk = t_begin;
    int attempts = 0;
    for (int min_size = 32; min_size <= MAXB && score < J.score && !too_large; min_size *= 2, attempts++) {      // :1021-1038
        for (int k = lane; k < 8 * MAXB; k += 64) s_buf[k] = BK_MIN;      // Allocated::clear
        s_temp[0][lane] = BK_MIN;
        s_temp[1][lane] = BK_MIN;
        S.trace_idx = S.block_idx = S.ck_trace_idx = S.ck_block_idx = 0;
        S.overflow = false;
        const int x_drop = -(min_size * L.gap_extend + L.gap_open);
        // a profile query is the crate's `reference` (the columns), the target its `query` (block_align_profile_aa_trace_xdrop(trace,
        // target, queryProfile, ..), :1047): ri then counts target residues, rj profile positions
        if (prof) bk_align(L, s_scores, S, T, Q, min_size, MAXB, x_drop, K, lane, &score, &ri, &rj);
        else bk_align(L, s_scores, S, Q, T, min_size, MAXB, x_drop, K, lane, &score, &ri, &rj);
        if (S.overflow) too_large = true;
    }
    // MAXB < 4096: the crate would go on to larger minimum sizes when the score is not reached - not decided by this instantiation
    if (MAXB < BLOCK_REF_MAX_SIZE && !too_large && score < J.score) too_large = true;
    if (too_large) {
        out.status = MMGPU_BLOCK_TOO_LARGE;
    } else if (!(score != J.score && !(J.score == 32767 && score >= J.score))) {      // :1058
        // Trace::cigar_core (scan_block.rs:1844-2006): serial walk from the end position to the origin; the reference
        // reverses the run order and then the string (:1071-1110), which leaves exactly the walk order
        __threadfence_block();
        t_walk = clock64();
        // The walk is one dependent chain, but what each step needs comes from memory the wavefront can fetch 64 items at a time:
        // all lanes walk together on identical state, and a step reads its trace bits and the two letters out of register
        // windows (16 trace entries = 64 qwords, 64 letters of each sequence, 64 operations to write) with readlanes - one
        // coalesced load per ~16 - 64 steps instead of four dependent loads per step by a single lane (~770 cycles a step before).
        uint32_t block_idx = S.block_idx;
        int i = __builtin_amdgcn_readfirstlane(ri), j = __builtin_amdgcn_readfirstlane(rj), table = 0;      // 0 = D, 1 = C, 2 = R
        uint32_t n = 0, ids = 0;
        char *bt = L.bt + out.bt_off;
        uint32_t win_lo = 0xFFFFFFFFu;                  // trace entries [win_lo, win_lo + 16): lane l holds qword (l & 3) of entry win_lo + (l >> 2)
        unsigned win_w0 = 0, win_w1 = 0;
        int qf_lo = -1, tf_lo = -1;                     // forward indices [lo, lo + 64) of the two sequences, one letter per lane
        int qwin = 0, twin = 0;
        int my_op = 0;                                  // operation n of the current group of 64 sits in lane n & 63
        while (i > 0 || j > 0) {
            int bi, bj, bh, bw, bright;
            uint32_t btstart;
            for (;;) {
                block_idx--;
                const BkBlock b = S.blocks[block_idx];
                bi = __builtin_amdgcn_readfirstlane((int)b.i); bj = __builtin_amdgcn_readfirstlane((int)b.j);
                bh = __builtin_amdgcn_readfirstlane((int)b.h); bw = __builtin_amdgcn_readfirstlane((int)b.w);
                bright = __builtin_amdgcn_readfirstlane((int)b.right);
                btstart = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.tstart);
                if (i >= bi && j >= bj) break;
            }
            while (i >= bi && j >= bj && (i > 0 || j > 0)) {
                const int ci = i - bi, cj = j - bj;
                uint32_t idx;
                int bit;
                if (bright) { idx = btstart + (uint32_t)(ci >> 6) + (uint32_t)cj * (uint32_t)((bh + 63) >> 6); bit = ci & 63; }
                else { idx = btstart + (uint32_t)(cj >> 6) + (uint32_t)ci * (uint32_t)((bw + 63) >> 6); bit = cj & 63; }
                if (idx < win_lo || idx - win_lo >= 16u) {      // (also the first step: win_lo = 2^32 - 1)
                    win_lo = idx >= 15u ? idx - 15u : 0u;
                    const unsigned long long v = S.trace[(size_t)win_lo * 4 + lane];
                    win_w0 = (unsigned)v;
                    win_w1 = (unsigned)(v >> 32);
                }
                const unsigned half = bit & 32 ? win_w1 : win_w0;
                const int e = (int)(idx - win_lo) * 4;
                const unsigned sh = (unsigned)bit & 31u;
                const unsigned tt = (((unsigned)__builtin_amdgcn_readlane((int)half, e) >> sh) & 1u) |
                                    ((((unsigned)__builtin_amdgcn_readlane((int)half, e + 1) >> sh) & 1u) << 1);
                const unsigned t2 = (((unsigned)__builtin_amdgcn_readlane((int)half, e + 2) >> sh) & 1u) |
                                    ((((unsigned)__builtin_amdgcn_readlane((int)half, e + 3) >> sh) & 1u) << 1);
                int op, di, dj, nt;      // OP_LUT (:1870-1933)
                if (bright) {
                    if (table == 1) { op = 5; di = 0; dj = 1; nt = (t2 & 1u) ? 0 : 1; }
                    else if (table == 2) { op = 4; di = 1; dj = 0; nt = (t2 & 2u) ? 0 : 2; }
                    else if (tt == 0) { op = 1; di = 1; dj = 1; nt = 0; }
                    else if (tt & 1u) { op = 5; di = 0; dj = 1; nt = (t2 & 1u) ? 0 : 1; }
                    else { op = 4; di = 1; dj = 0; nt = (t2 & 2u) ? 0 : 2; }
                } else {
                    if (table == 2) { op = 4; di = 1; dj = 0; nt = (t2 & 1u) ? 0 : 2; }
                    else if (table == 1) { op = 5; di = 0; dj = 1; nt = (t2 & 2u) ? 0 : 1; }
                    else if (tt == 0) { op = 1; di = 1; dj = 1; nt = 0; }
                    else if (tt & 1u) { op = 4; di = 1; dj = 0; nt = (t2 & 1u) ? 0 : 2; }
                    else { op = 5; di = 0; dj = 1; nt = (t2 & 2u) ? 0 : 1; }
                }
                if (op == 1) {      // identities: the letters of DP indices i, j = forward indices end - (i - 1), end - (j - 1), both rising along the walk
                    const int qf = Q.end - ((prof ? j : i) - 1), tf = T.end - ((prof ? i : j) - 1);
                    if ((unsigned)(qf - qf_lo) >= 64u || qf_lo < 0) {
                        qf_lo = qf;
                        qwin = qf_lo + lane <= Q.end ? (int)Q.res[qf_lo + lane] : 255;
                    }
                    if ((unsigned)(tf - tf_lo) >= 64u || tf_lo < 0) {
                        tf_lo = tf;
                        twin = tf_lo + lane <= T.end ? (int)T.res[tf_lo + lane] : 254;
                    }
                    ids += __builtin_amdgcn_readlane(qwin, qf - qf_lo) == __builtin_amdgcn_readlane(twin, tf - tf_lo) ? 1u : 0u;
                }
                // (a profile query sits on the crate's reference side: its I consumes the target = the reference's 'D', :1083-1101)
                if (lane == (int)(n & 63u)) my_op = op == 1 ? 'M' : ((op == 4) != prof ? 'I' : 'D');
                n++;
                if ((n & 63u) == 0) bt[n - 64 + lane] = (char)my_op;
                i -= di;
                j -= dj;
                table = nt;
            }
        }
        if ((n & 63u) != 0 && lane < (int)(n & 63u)) bt[(n & ~63u) + lane] = (char)my_op;
        out.status = MMGPU_BLOCK_OK;
        out.q_start = J.q_end + 1 - (prof ? rj : ri);       // :1111-1112
        out.t_start = J.t_end + 1 - (prof ? ri : rj);
        out.ident = ids;
        out.bt_len = n;
    }
    {   // profiling aid in the reserved field: share of the pair's time spent in the serial walk back, in 1/1000 (low 16 bits),
        // and the number of minimum block sizes tried (32, 64, ...: bits 16-23)
        const long long t_end = clock64();
        const int share = (t_walk > t_begin && t_end > t_begin) ? (int)(((t_end - t_walk) * 1000) / (t_end - t_begin)) : 0;
        out.reserved = (int32_t)((share & 0xFFFF) | ((attempts & 0xFF) << 16));
    }
    if (L.growth != nullptr) {      // test aid: the block list of the last run (wave-uniform branch; off in every product call)
        __threadfence_block();
        uint32_t *g = L.growth + (size_t)J.slot * (1 + 4 * (size_t)L.growth_cap);
        const uint32_t nb = too_large ? 0u : S.block_idx;
        if (lane == 0) g[0] = nb;
        for (uint32_t k = (uint32_t)lane; k < nb && k < L.growth_cap; k += 64) {
            const BkBlock bb = S.blocks[k];
            g[1 + 4 * k] = bb.i;
            g[2 + 4 * k] = bb.j;
            g[3 + 4 * k] = (uint32_t)bb.h << 16 | bb.w;
            g[4 + 4 * k] = bb.right;
        }
    }
    if (lane == 0) {
        L.out[J.slot] = out;
        __threadfence();
        atomicExch(&L.pool_busy[s_slot], 0u);
    }
}

}  // namespace

hipError_t launch_sw_block(const BlockLaunch &L, int tier, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    if (tier == 2) hipLaunchKernelGGL((sw_block_kernel<BLOCK_REF_MAX_SIZE, false>), dim3(L.n_jobs), dim3(64), 0, stream, L);
    else if (tier == 1) hipLaunchKernelGGL((sw_block_kernel<BLOCK_MID_SIZE, true>), dim3(L.n_jobs), dim3(64), 0, stream, L);
    else hipLaunchKernelGGL((sw_block_kernel<BLOCK_MAX_SIZE, true>), dim3(L.n_jobs), dim3(64), 0, stream, L);
    return hipGetLastError();
}

// first use of any kernel of this file loads its code object (tens of milliseconds): mmgpu_warmup does it ahead of time
void warm_block() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&(sw_block_kernel<BLOCK_MAX_SIZE, true>)));
}

}  // namespace mmgpu
