// The block aligner for int16-range hits (row a15), FOUR PAIRS PER WAVEFRONT: SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ>
// (src/alignment/StripedSmithWaterman.cpp:943-1127) -> Block<TRACE, X_DROP>::align_aa of lib/block-aligner 0.4.0, AVX2 configuration
// (scan_block.rs:120-632 align_core, :1449-1613 place_block, avx2.rs:294-337).  Same vector operations on the same saturating int16
// values as block_kernel.hip (one pair per wavefront; it stays for profile queries and for blocks beyond this kernel's rows) and as
// oracle/block_oracle.c.  The mapping:
//
//  * ONE 16-LANE DPP ROW = ONE AVX2 VECTOR of the crate = one pair.  A wavefront carries four pairs, each in its own row, as plain
//    SIMT: every variable of align_core is a per-lane value that is uniform inside a row, every branch of it is ordinary divergence
//    between rows, every cross-lane operation of the crate (the byte shifts inside 128-bit halves, the prefix scan, broadcasts of
//    the last lane) is a DPP move confined to a row - rows never exchange anything.  block_kernel.hip spends a whole wavefront
//    (half of it idle at the usual 32-row blocks) and ~190 instructions on one 64-row step of ONE pair's column;
//  * saturating int16 arithmetic is v_pk_add_i16 clamp / v_pk_max_i16 on the low half of a register (the high half stays zero);
//  * a place_block job runs in OCTETS of eight columns (STEP = 8; every job's width is a multiple of it), and inside an octet
//    CHUNK BY CHUNK (16 rows = one vector of the crate), all eight columns of a chunk before the next chunk: the chunk's D / C
//    state and its rows' letters stay in registers for the eight columns, what the crate carries from a chunk to the one below -
//    the last lane's D (the corner of D00), R (R01) and trace_R - is kept per column in the banks of two registers each (a DPP
//    write with a bank mask, a row_newbcast to read).  The dataflow is the crate's, only the order of independent cells differs;
//    the lane's running maximum keeps the crate's "last cell that reached it" through a key (score, column, chunk);
//  * every trip of the main loop is TWO chunks of each row's octet: rows at the usual 32-row blocks finish an octet and take their
//    align_core step on every trip, in step with each other; rows at larger blocks take more trips and hold nobody up;
//  * the border arrays D_col / C_col / D_row / R_row live in LDS (2 KB per pair at 256 rows), their checkpoints in an HBM scratch
//    slot of the row;
//  * rows are persistent: a row that finishes its pair takes the next one off a global queue (longest pairs first);
//  * the trace is four bits per cell, one dword per row and octet, written coalesced; the walk back is a second kernel with one
//    LANE per pair (sw_block4_walk_kernel);
//  * TRACE = false (the caller wants start positions only - `mmseqs search` without -a, alignment mode 2, where neither the
//    identities nor the string reach the output, Matcher.cpp:107-127): no trace, no block list, no scratch memory, no walk.
// A pair this kernel does not decide (blocks would grow beyond MAXB rows, its trace slot overflows) is answered
// MMGPU_BLOCK_TOO_LARGE and goes on to block_kernel.hip's tiers.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

constexpr int B4_STEP = 8, B4_ZERO = 16384, B4_X_DROP_ITER = 2;      // (MIN = 0)

typedef short b4_s16x2 __attribute__((ext_vector_type(2)));

// DP values: the int16 bit pattern in the low half of a register, the high half zero
__device__ __forceinline__ unsigned b4_adds(unsigned a, unsigned b) {      // v_pk_add_i16 clamp
    return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(b4_s16x2, a), __builtin_bit_cast(b4_s16x2, b)));
}
__device__ __forceinline__ unsigned b4_maxs(unsigned a, unsigned b) {      // v_pk_max_i16
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(b4_s16x2, a), __builtin_bit_cast(b4_s16x2, b)));
}
__device__ __forceinline__ unsigned b4_pat(int v) { return (unsigned)v & 0xFFFFu; }
__device__ __forceinline__ int b4_sx(unsigned p) { return (int)(short)p; }
__device__ __forceinline__ int b4_adds_i(int a, int b) { const int s = a + b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }
__device__ __forceinline__ int b4_subs_i(int a, int b) { const int s = a - b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }

// a DPP move that keeps `old` where a lane has no source or its bank is masked
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ unsigned b4_dpp(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK, false);
}
// ... and one without an old value: zeros where a lane has no source (no register to initialise)
template <int CTRL>
__device__ __forceinline__ unsigned b4_mov(unsigned src) { return (unsigned)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xF, 0xF, true); }
template <int LANE> __device__ __forceinline__ unsigned b4_bcast(unsigned v) { return b4_mov<0x150 + LANE>(v); }      // row_newbcast:LANE
// maximum over the 16 lanes of the row, in every lane of it
__device__ __forceinline__ int b4_rowmax(int v) {
    v = max(v, (int)b4_mov<0x128>((unsigned)v));      // row_ror:8
    v = max(v, (int)b4_mov<0x124>((unsigned)v));
    v = max(v, (int)b4_mov<0x122>((unsigned)v));
    v = max(v, (int)b4_mov<0x121>((unsigned)v));
    return v;
}

enum { B4_RIGHT = 0, B4_DOWN = 1, B4_GROW = 2 };
enum { B4_JOB_SHIFT = 1, B4_JOB_GROW1 = 2, B4_JOB_GROW2 = 3 };

// SKEW = false: four pairs per wavefront, as above.  SKEW = true: ONE pair per wavefront for the pairs whose blocks outgrow the
// first form (MAXB 1024, then the crate's 4096) - the four rows work on the same octet, row r on columns r and r + 4, one chunk
// behind row r - 1: cell (chunk, column) needs (chunk - 1, column) - the row's own previous trip - and (chunk, column - 1) - row
// r - 1's previous trip, handed over through a two-slot ring in LDS; column 3's results go into the border array in place, where
// row 0 picks them up for column 4.  An octet of n chunks takes max(n, 4) + n + 3 trips of ONE column-chunk each instead of
// block_kernel.hip's 8 x n / 4 steps of 64 rows: a quarter of the dependent chain, and the chain is what these pairs cost.
template <bool TRACE, int MAXB, bool SKEW>
__global__ __launch_bounds__(64) void sw_block4_kernel(Block2Launch L) {
    __shared__ uint32_t s_sc[27 * 32];          // AAMatrix::scores as int16 patterns, a dword each (no masking after the load)
    __shared__ int16_t s_arr[SKEW ? 1 : 4][4 * MAXB];      // per pair: D_col, C_col, D_row, R_row
    __shared__ uint32_t s_ring[4][2][2][16];    // SKEW: row r's D11 / C11 (both halves) of its last trip, for row r + 1 (row 3's low half: for row 0's high half)
    __shared__ int16_t s_out[2][8];             // SKEW: the octet's last row (D_row[j] / R_row[j])
    __shared__ uint32_t s_col[2][8];            // SKEW: the octet's columns (score-row offset, bias)
    const int lane = (int)threadIdx.x;
    for (int x = lane; x < 27 * 32; x += 64) s_sc[x] = (uint32_t)(uint16_t)(int16_t)L.scores[x];
    __syncthreads();
    const int k = lane & 15;
    const int lid = SKEW ? lane : k, lstride = SKEW ? 64 : 16;      // loops over a pair's arrays: the pair's lanes
    int16_t *const A = s_arr[SKEW ? 0 : lane >> 4];
    // the four checkpoint arrays of the row's pair: scratch in HBM (written at every new maximum, read when a block grows; the row's
    // own stores and loads, same lane, same address, one L1, program order)
    uint32_t *const CK = reinterpret_cast<uint32_t *>(L.ck_pool) + (SKEW ? (size_t)blockIdx.x : (size_t)blockIdx.x * 4 + (size_t)(lane >> 4)) * (size_t)(2 * MAXB);
    const int go = L.gap_open, ge = L.gap_extend;
    constexpr unsigned SPL = SKEW ? 0x00010001u : 1u;      // SKEW: both int16 halves of a register carry a cell
    const unsigned m1 = (k & 7) >= 1 ? 0xFFFFu * SPL : 0u, m2 = (k & 7) >= 2 ? 0xFFFFu * SPL : 0u;
    unsigned gap_all, consts;
    {   // avx2.rs:294-309: the 1 .. 16 x gap_extend ladder of a vector
        const unsigned g = b4_pat(ge) * SPL;
        const unsigned s1 = b4_adds(b4_mov<0x111>(g) & m1, g);
        const unsigned s2 = b4_adds(b4_mov<0x112>(s1) & m2, s1);
        const unsigned s4 = b4_adds(b4_dpp<0x114, 0xB>(0u, s2), s2);
        const unsigned w7 = b4_bcast<7>(s4);
        gap_all = b4_adds(k < 8 ? 0u : w7, s4);
        consts = s4;
    }
    const unsigned g1 = b4_pat(ge) * SPL, g2 = b4_pat((int)(short)(ge << 1)) * SPL, g4 = b4_pat((int)(short)(ge << 2)) * SPL;
    const unsigned gop = b4_pat(go) * SPL, gep = b4_pat(ge) * SPL, gome = b4_pat(b4_subs_i(go, ge)) * SPL;

    // ---- the row's pair (every value uniform inside the row) ----
    bool active = false;
    uint32_t Qoff = 0, To4 = 0, slot = 0;
    int q_end = 0, t_end = 0, Qlen = 0, Tlen = 0, target = 0;
    uint64_t pool_off = 0;
    uint32_t blocks_bytes = 0;
    // align_core (scan_block.rs:120-632)
    int best_max = 0, best_i = 0, best_j = 0, prev_dir = B4_GROW, dir = B4_GROW, prev_size = 0, bs = 32, off = 0, off_max = 0;
    int y_drop_iter = 0, x_drop_iter = 0, st_i = 0, st_j = 0, i_ck = 0, j_ck = 0, off_ck = 0, D_corner = 0, off_add = 0;
    int min_size = 32, x_drop = 0, score = 0, ri = 0, rj = 0, resume_min = 32;
    uint32_t trace_idx = 0, block_idx = 0, ck_trace_idx = 0, ck_block_idx = 0, trace_cap = 0, block_cap = 0;
    bool overflow = false, slot_overflow = false;      // (slot_overflow: the pair's trace / block-list slot was too small - not its blocks too large)
    // the current place_block job (scan_block.rs:1449-1613): eight columns (an octet) at a time, two chunks of the octet per trip
    int J_kind = B4_JOB_SHIFT, J_start_row = 0, J_start_col = 0, J_width = 8, J_height = 0, J_oct = 0, J_c = 0, J_corner = 0, J_out_base = 0, J_off_add = 0;
    bool J_rq = true, J_origin = false;      // rq: rows run over the query (shift right, second grow job), else over the reference
    // per lane: D_max, column and chunk of the LAST cell that reached it (this job's / the first grow job's) as ONE ordered key:
    // D_max << 16 | column << 8 | chunk (SKEW, blocks up to 4096 rows: D_max << 32 | column << 8 | chunk)
    typedef typename std::conditional<SKEW, long long, int>::type key_t;
    key_t Mkey = 0, Gkey = 0;
    auto make_key = [](unsigned d, unsigned column, unsigned chunk) -> key_t {
        if (SKEW) return (key_t)(((long long)(int)(short)d << 32) | (long long)((column << 8) | chunk));
        return (key_t)(int)((d << 16) | (column << 8) | chunk);
    };
    auto key_dm = [](key_t key) { return SKEW ? (int)((long long)key >> 32) : (int)key >> 16; };
    auto key_aj = [](key_t key) { return SKEW ? (int)(((unsigned)key >> 8) & 0xFFFFFu) : ((int)key >> 8) & 0xFF; };
    auto key_ai = [](key_t key) { return ((int)key & 0xFF) * 16; };
    // what a chunk hands to the one below, per column of the octet: its last lane's D (the corner of D00 one column later), R (R01)
    // and trace_R flag - column b in bank b of _lo (b < 4) / bank b - 4 of _hi.  After the octet's last chunk: D_row[j] / R_row[j]
    unsigned pD_lo = 0, pD_hi = 0, pR_lo = 0, pR_hi = 0, pT_lo = 0, pT_hi = 0, init15_prev = 0;
    unsigned colv = 0, colb = 0;             // lane j (and j + 8): column j of the octet - its letter as the byte offset of a score row, its bias

    auto q_ptr = [&]() { return L.q_res + Qoff; };
    auto t_ptr = [&]() { return L.t_res + (size_t)To4 * 4; };

    // ---- the octet's columns (first trip of an octet) ----
    auto octet_setup = [&]() {
        const int pc = J_start_col + J_oct * 8 + (k & 7);
        const int len = J_rq ? Tlen : Qlen, end = J_rq ? t_end : q_end;
        const bool in = pc >= 1 && pc <= len;      // PaddedBytes::get: index 0 and everything past the end is the padding letter
        const int idx = in ? end - (pc - 1) : 0;
        const uint8_t *base = J_rq ? t_ptr() : q_ptr();
        const int letter = in ? (int)base[idx] : 26;
        const int bias = (int)L.q_cb[Qoff + (J_rq ? 0 : idx)];
        colv = (unsigned)letter * 128u;
        colb = (!J_rq && in) ? b4_pat(bias) : 0u;
        pD_lo = 0; pD_hi = 0; pR_lo = 0; pR_hi = 0; pT_lo = 0; pT_hi = 0;
        init15_prev = J_oct == 0 ? b4_pat(J_corner) : 0u;      // D_corner: the first column of the job only
    };
    // ---- one chunk (16 rows) of the octet: its eight columns ----
    auto chunk_pass = [&](int c) {
        const bool first = J_oct == 0;
        // just_offset (scan_block.rs:1102-1123, the branch without a shift) of a shift job's own arrays: on the way in
        const unsigned offp = (J_kind == B4_JOB_SHIFT && first) ? b4_pat(J_off_add) : 0u;
        int16_t *const rA = A + (J_rq ? 0 : 2 * MAXB), *const rB = rA + MAXB;
        const int rlen = J_rq ? Qlen : Tlen, rend = J_rq ? q_end : t_end;
        const uint8_t *const rbase = J_rq ? q_ptr() : t_ptr();
        const int x = 16 * c + k;
        const int p = J_start_row + x;
        const bool in = p >= 1 && p <= rlen;
        const int idx = in ? rend - (p - 1) : 0;
        const unsigned rowidx = (unsigned)((in ? (int)rbase[idx] : 26) & 31) * 4u;
        const int rbias = (int)L.q_cb[Qoff + (J_rq ? idx : 0)];
        const unsigned rowb = (J_rq && in) ? b4_pat(rbias) : 0u;
        const char *const srow = reinterpret_cast<const char *>(s_sc) + rowidx;
        unsigned D10 = b4_adds((unsigned)(uint16_t)rA[x], offp), C10 = b4_adds((unsigned)(uint16_t)rB[x], offp);
        const unsigned init15 = b4_bcast<15>(D10);
        unsigned nD_lo = 0, nD_hi = 0, nR_lo = 0, nR_hi = 0, nT_lo = 0, nT_hi = 0;
        unsigned acc = 0;
        const bool omask = J_origin && first && c == 0 && k == 0;
        auto column = [&](auto JC) {
            constexpr int j = decltype(JC)::value;
            constexpr int bank = 1 << (j & 3), rd = 0x150 + 4 * (j & 3), rdp = 0x150 + 4 * ((j + 3) & 3);
            const unsigned corner = j == 0 ? init15_prev : b4_mov<rdp>((j - 1) < 4 ? pD_lo : pD_hi);
            const unsigned D00 = b4_dpp<0x111>(corner, D10);      // row_shr:1, lane 0 keeps the corner
            const unsigned sc = *reinterpret_cast<const uint32_t *>(srow + b4_bcast<j>(colv));
            unsigned D11 = b4_adds(D00, b4_adds(sc, b4_adds(b4_bcast<j>(colb), rowb)));
            if (j == 0) D11 = omask ? (unsigned)B4_ZERO : D11;      // the cell (0, 0)
            const unsigned C11o = b4_adds(D10, gop);
            const unsigned C11 = b4_maxs(b4_adds(C10, gep), C11o);
            D11 = b4_maxs(D11, C11);
            const unsigned D11o = b4_adds(D11, gome);
            // simd_prefix_scan_i16 (avx2.rs:311-337); simd_sllz_i16!: byte shifts inside the 128-bit halves, zeros shifted in
            const unsigned p1 = b4_maxs(D11o, b4_adds(b4_mov<0x111>(D11o) & m1, g1));
            const unsigned p2 = b4_maxs(p1, b4_adds(b4_mov<0x112>(p1) & m2, g2));
            const unsigned p4 = b4_maxs(p2, b4_adds(b4_dpp<0x114, 0xB>(0u, p2), g4));      // (bank 2 = lanes 8 - 11 masked: zeros)
            // lanes 0 - 3: themselves (no source: they keep `old`), 4 - 7: four lanes down, 8 - 15: lane 7
            const unsigned from = b4_dpp<0x157, 0xC>(b4_dpp<0x114>(p4, p4), p4);
            unsigned R11 = b4_maxs(p4, b4_adds(from, consts));
            // R11 = max(R11, broadcast(R01's last lane) + gap_extend_all): R01 = the chunk above, this column
            R11 = b4_maxs(R11, b4_adds(b4_mov<rd>(j < 4 ? pR_lo : pR_hi), gap_all));
            D11 = b4_maxs(D11, R11);
            if (TRACE) {
                acc = acc * 2u + (D11 == C11 ? 1u : 0u);
                acc = acc * 2u + (D11 == R11 ? 1u : 0u);
                acc = acc * 2u + (C11 == C11o ? 1u : 0u);
                const unsigned tempR = R11 == D11o ? 1u : 0u;
                const unsigned trR = b4_dpp<0x111>(b4_mov<rd>(j < 4 ? pT_lo : pT_hi), tempR);
                acc = acc * 2u + trR;
                if (j < 4) nT_lo = b4_dpp<0x15F, bank>(nT_lo, tempR);
                else nT_hi = b4_dpp<0x15F, bank>(nT_hi, tempR);
            }
            Mkey = max(Mkey, make_key(D11, (unsigned)(J_oct * 8 + j), (unsigned)c));
            if (j < 4) { nD_lo = b4_dpp<0x15F, bank>(nD_lo, D11); nR_lo = b4_dpp<0x15F, bank>(nR_lo, R11); }
            else { nD_hi = b4_dpp<0x15F, bank>(nD_hi, D11); nR_hi = b4_dpp<0x15F, bank>(nR_hi, R11); }
            D10 = D11;
            C10 = C11;
        };
        column(std::integral_constant<int, 0>{}); column(std::integral_constant<int, 1>{});
        column(std::integral_constant<int, 2>{}); column(std::integral_constant<int, 3>{});
        column(std::integral_constant<int, 4>{}); column(std::integral_constant<int, 5>{});
        column(std::integral_constant<int, 6>{}); column(std::integral_constant<int, 7>{});
        rA[x] = (int16_t)D10;
        rB[x] = (int16_t)C10;
        if (TRACE) {      // column j's four bits at 28 - 4 j: D11 == C11, D11 == R11, C11 == C11_open, trace_R
            const bool ok = trace_idx + (uint32_t)(J_height >> 4) * 16u <= trace_cap;
            uint32_t *const trace = reinterpret_cast<uint32_t *>(L.pool + pool_off + blocks_bytes);
            if (ok) trace[trace_idx + (uint32_t)x] = acc;
        }
        pD_lo = nD_lo; pD_hi = nD_hi; pR_lo = nR_lo; pR_hi = nR_hi; pT_lo = nT_lo; pT_hi = nT_hi;
        init15_prev = init15;
    };
    // the octet's last-row values (pD / pR after its last chunk) into entries base .. base + 7 of two arrays
    auto put_outputs = [&](int16_t *a1, int16_t *a2, int base) {
        if (SKEW) {
            if (k < 8) { a1[base + k] = s_out[0][k]; a2[base + k] = s_out[1][k]; }
        } else if ((k & 3) == 0) {
            const int b = base + (k >> 2);
            a1[b] = (int16_t)pD_lo; a1[b + 4] = (int16_t)pD_hi;
            a2[b] = (int16_t)pR_lo; a2[b + 4] = (int16_t)pR_hi;
        }
    };

    auto add_block = [&](int i, int j, int width, int height, int right) {
        if (!TRACE) return;
        if (block_idx < block_cap) {
            if (k == 0) {
                BkBlock b;
                b.i = (uint32_t)i; b.j = (uint32_t)j; b.h = (uint16_t)height; b.w = (uint16_t)width; b.right = (uint32_t)right; b.tstart = trace_idx;
                reinterpret_cast<BkBlock *>(L.pool + pool_off)[block_idx] = b;
            }
        } else {
            overflow = true;
            slot_overflow = true;
        }
        block_idx++;
    };
    auto checkpoint = [&](int n) {      // n entries of the four arrays -> the checkpoint (as dwords)
        for (int x = lid; x < n / 2; x += lstride)
            for (int a = 0; a < 4; a++) CK[a * (MAXB / 2) + x] = reinterpret_cast<const uint32_t *>(A + a * MAXB)[x];
    };
    auto restore = [&](int n) {
        for (int x = lid; x < n / 2; x += lstride)
            for (int a = 0; a < 4; a++) reinterpret_cast<uint32_t *>(A + a * MAXB)[x] = CK[a * (MAXB / 2) + x];
    };
    auto pmax8 = [&](int a) {      // prefix_max: the maximum of the first STEP entries (scan_block.rs:1125-1140)
        return b4_rowmax(k < B4_STEP ? (int)A[a * MAXB + k] : -32768);
    };

    auto fetch = [&]() {      // the next pair of the queue
        uint32_t idx = 0;
        if (SKEW) {
            if (lane == 0) idx = atomicAdd(L.counter, 1u);
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        } else {
            if (k == 0) idx = atomicAdd(L.counter, 1u);
            idx = b4_bcast<0>(idx);
        }
        active = idx < L.n_jobs;
        if (!active) return;
        const Block2Job J = L.jobs[idx];
        target = J.score; slot = J.slot; q_end = J.q_end; t_end = J.t_end;
        Qlen = q_end + 1; Tlen = t_end + 1;
        Qoff = L.q_off[J.query];
        To4 = L.t_off4[J.target];
        if (TRACE) {
            const uint32_t cap = (uint32_t)(Qlen + Tlen + 64);
            blocks_bytes = (cap * (uint32_t)sizeof(BkBlock) + 31u) & ~31u;
            pool_off = J.pool_off;
            trace_cap = J.pool_bytes > blocks_bytes ? (J.pool_bytes - blocks_bytes) / (SKEW ? 1u : 4u) : 0u;      // dwords (SKEW: bytes)
            block_cap = J.pool_bytes > blocks_bytes ? cap : 0u;
        }
        min_size = J.pad ? (int)J.pad : 32;      // (minimum sizes below it were tried by the launch before and did not reach the score)
        score = -1000000000; ri = 0; rj = 0;
    };

    auto start_attempt = [&]() {      // Allocated::clear + align_core's initial state
        for (int x = lid; x < 2 * MAXB; x += lstride) reinterpret_cast<uint32_t *>(A)[x] = 0u;      // (the checkpoint is written before it is read)
        best_max = 0; best_i = 0; best_j = 0;
        prev_dir = B4_GROW; dir = B4_GROW;
        prev_size = 0; bs = min_size;
        off = 0; off_max = 0;
        y_drop_iter = 0; x_drop_iter = 0;
        st_i = 0; st_j = 0; i_ck = 0; j_ck = 0; off_ck = 0;
        D_corner = 0; off_add = 0;
        trace_idx = 0; block_idx = 0; ck_trace_idx = 0; ck_block_idx = 0;
        overflow = false; slot_overflow = false;
        x_drop = -(min_size * ge + go);
    };

    auto set_job = [&](int kind, bool rq, int start_row, int start_col, int width, int height, int corner, int out_base, int joff) {
        J_kind = kind; J_rq = rq; J_start_row = start_row; J_start_col = start_col; J_width = width; J_height = height;
        J_oct = 0; J_c = 0; J_corner = corner; J_out_base = out_base; J_off_add = joff;
        J_origin = start_row == 0 && start_col == 0;
        Mkey = 0;
    };
    // the second grow job (and the only one of the first block): place_block(query, reference, st_i, st_j + prev_size, grow_step, block_size)
    auto begin_grow2 = [&]() {
        add_block(st_i, st_j + prev_size, bs - prev_size, bs, 1);
        set_job(B4_JOB_GROW2, true, st_i, st_j + prev_size, bs - prev_size, bs, 0, prev_size, 0);
    };
    // top of align_core's loop: the next job of `dir`
    auto begin_step = [&]() {
        const int prev_off = off;
        Gkey = 0;
        if (dir == B4_RIGHT || dir == B4_DOWN) {
            const bool r = dir == B4_RIGHT;
            off = off_max;
            const int d = prev_off - off;
            off_add = d < -32768 ? -32768 : (d > 32767 ? 32767 : d);
            add_block(r ? st_i : st_i + bs - B4_STEP, r ? st_j + bs - B4_STEP : st_j, r ? B4_STEP : bs, r ? bs : B4_STEP, r ? 1 : 0);
            const int corner = prev_dir == (r ? B4_DOWN : B4_RIGHT) ? b4_adds_i(D_corner, off_add) : 0;
            set_job(B4_JOB_SHIFT, r, r ? st_i : st_j, (r ? st_j : st_i) + bs - B4_STEP, B4_STEP, bs, corner, 0, off_add);
        } else {
            D_corner = 0;
            const int grow_step = bs - prev_size;
            add_block(st_i + prev_size, st_j, prev_size, grow_step, 0);
            // place_block(reference, query, st_j, st_i + prev_size, grow_step, prev_size): nothing for the first block
            if (prev_size > 0) set_job(B4_JOB_GROW1, false, st_j, st_i + prev_size, grow_step, prev_size, 0, prev_size, 0);
            else begin_grow2();
        }
    };

    // the pair is decided: status, start positions; with TRACE the walk kernel finishes the record
    auto finish_pair = [&](bool too_large) {
        mmgpu_sw_block out;
        out.q_start = -1; out.t_start = -1; out.ident = 0; out.bt_len = 0; out.bt_off = L.bt_off ? L.bt_off[slot] : 0;
        out.reserved = 0;
        if (too_large) {
            out.status = MMGPU_BLOCK_TOO_LARGE;
            out.reserved = resume_min | (slot_overflow ? 0x10000 : 0);      // the first minimum size the next launch has to try; bit 16: only the slot was too small
        } else if (!(score != target && !(target == 32767 && score >= target))) {      // StripedSmithWaterman.cpp:1058
            out.status = MMGPU_BLOCK_OK;
            out.q_start = q_end + 1 - ri;       // :1111-1112
            out.t_start = t_end + 1 - rj;
            if (TRACE) {      // for the walk kernel: end cell and number of blocks
                out.ident = (uint32_t)ri; out.bt_len = (uint32_t)rj; out.reserved = (int32_t)block_idx;
            }
        } else {
            out.status = MMGPU_BLOCK_DECLINED;
        }
        if (TRACE && L.growth != nullptr) {      // test aid: the block list of the last run
            __threadfence_block();
            uint32_t *g = L.growth + (size_t)slot * (1 + 4 * (size_t)L.growth_cap);
            const uint32_t n = too_large ? 0u : block_idx;
            if (k == 0) g[0] = n;
            const BkBlock *blocks = reinterpret_cast<const BkBlock *>(L.pool + pool_off);
            for (uint32_t x = (uint32_t)k; x < n && x < L.growth_cap; x += 16) {
                const BkBlock bb = blocks[x];
                g[1 + 4 * x] = bb.i;
                g[2 + 4 * x] = bb.j;
                g[3 + 4 * x] = (uint32_t)bb.h << 16 | bb.w;
                g[4 + 4 * x] = bb.right;
            }
        }
        if (k == 0) L.out[slot] = out;
    };

    // runs after the row's job is complete (or at a fresh pair): everything of align_core up to the next job.  Leaves active = false
    // when the queue is empty.
    auto advance = [&]() {
        if (SKEW) {      // the vector lane's maximum over the four rows' columns
            for (int d = 16; d < 64; d <<= 1) {
                const long long o = __shfl_xor((long long)Mkey, d, 64);
                Mkey = (key_t)max((long long)Mkey, o);
            }
        }
        if (J_kind == B4_JOB_GROW1) {      // grow_D_max = this job's maxima; the second job follows
            Gkey = Mkey;
            begin_grow2();
            return;
        }
        int right_max, down_max;
        if (J_kind == B4_JOB_SHIFT) {
            // shift_and_offset (scan_block.rs:1102-1123) of the arrays the job did NOT run along (D_row / R_row after a shift right):
            // entries move down by STEP and take off_add, the last STEP entries are the job's last row
            int16_t *const o1 = A + (dir == B4_RIGHT ? 2 : 0) * MAXB, *const o2 = o1 + MAXB;
            D_corner = b4_adds_i((int)o1[B4_STEP - 1], off_add);
            const unsigned offp = b4_pat(off_add);
            for (int x = lid; x < bs - B4_STEP; x += lstride) {
                const unsigned v1 = b4_adds((unsigned)(uint16_t)o1[x + B4_STEP], offp), v2 = b4_adds((unsigned)(uint16_t)o2[x + B4_STEP], offp);
                o1[x] = (int16_t)v1;
                o2[x] = (int16_t)v2;
            }
            put_outputs(o1, o2, bs - B4_STEP);
            right_max = pmax8(0);
            down_max = pmax8(2);
        } else {
            right_max = pmax8(0);
            down_max = pmax8(2);
            checkpoint(bs);
            ck_trace_idx = trace_idx;
            ck_block_idx = block_idx;
        }
        bool done = false;      // align_core returns
        if (overflow) {
            done = true;
        } else {
            const int cur = dir;
            prev_dir = cur;
            const int D_max_max = b4_rowmax(key_dm(Mkey));
            const int grow_max = b4_rowmax(key_dm(Gkey));      // (MIN unless this step grew)
            const int mx = max(D_max_max, grow_max);
            off_max = off + mx - B4_ZERO;
            y_drop_iter++;
            bool grow_no_max = cur == B4_GROW;
            if (off_max > best_max) {
                {   // location of the maximum: per vector lane the last cell that reached its maximum, over the vector lanes the
                    // largest column, then the largest row (:374-444)
                    const bool grow = cur == B4_GROW && D_max_max < grow_max;
                    const int curr_max = grow ? grow_max : D_max_max;
                    const key_t key = grow ? Gkey : Mkey;
                    const int dm = key_dm(key), aj = key_aj(key), ai = key_ai(key);
                    const int r = ai + k, cc = (bs - B4_STEP) + aj;
                    int gi, gj;
                    if (grow) { gi = st_i + prev_size + aj; gj = st_j + ai + k; }
                    else if (cur == B4_RIGHT) { gi = st_i + r; gj = st_j + cc; }
                    else if (cur == B4_DOWN) { gi = st_i + cc; gj = st_j + r; }
                    else { gi = st_i + ai + k; gj = st_j + prev_size + aj; }
                    const bool valid = dm == curr_max;
                    const int mj = b4_rowmax(valid ? gj : 0);
                    const int mi = b4_rowmax((valid && gj == mj) ? gi : 0);
                    best_j = mj;
                    best_i = mi;
                }
                if (bs < BLOCK_REF_MAX_SIZE) {
                    i_ck = st_i; j_ck = st_j; off_ck = off;
                    checkpoint(bs);
                    ck_trace_idx = trace_idx;
                    ck_block_idx = block_idx;
                    grow_no_max = false;
                }
                best_max = off_max;
                y_drop_iter = 0;
            }
            bool cont = false;      // `continue` of the crate's loop: the next step is decided
            if (off_max < best_max - x_drop) {
                if (x_drop_iter < B4_X_DROP_ITER - 1) x_drop_iter++;
                else done = true;
            } else {
                x_drop_iter = 0;
            }
            if (!done) {
                if (st_i + bs > Qlen && st_j + bs > Tlen) {
                    done = true;
                } else if (st_j + bs > Tlen) {
                    st_i += B4_STEP; dir = B4_DOWN; cont = true;
                } else if (st_i + bs > Qlen) {
                    st_j += B4_STEP; dir = B4_RIGHT; cont = true;
                }
            }
            if (!done && !cont) {
                const int next_size = bs * 2;
                const bool want_grow = y_drop_iter > (bs / B4_STEP) - 1 || grow_no_max;
                if (next_size <= MAXB) {
                    if (want_grow) {
                        prev_size = bs;
                        bs = next_size;
                        dir = B4_GROW;
                        st_i = i_ck; st_j = j_ck; off = off_ck;
                        restore(prev_size);
                        trace_idx = ck_trace_idx;
                        block_idx = ck_block_idx;
                        y_drop_iter = 0;
                        cont = true;
                    }
                } else if (next_size <= BLOCK_REF_MAX_SIZE && want_grow) {
                    overflow = true;      // the crate would grow beyond what this kernel holds: not decided here
                    done = true;
                }
            }
            if (!done && !cont) {
                if (bs > min_size && y_drop_iter == 0) {      // SHRINK (:542-586)
                    const int s1 = max((int)A[2 * MAXB + bs - 1], (int)A[2 * MAXB + bs - 2]);
                    const int s2 = max((int)A[bs - 1], (int)A[bs - 2]);
                    if (max(s1, s2) >= mx) {
                        prev_dir = B4_GROW;
                        bs /= 2;
                        for (int x = lid; x < bs; x += lstride)      // copy_vec(i, i + block_size)
                            for (int a = 0; a < 4; a++) A[a * MAXB + x] = A[a * MAXB + x + bs];
                        st_i += bs;
                        st_j += bs;
                        i_ck = st_i; j_ck = st_j; off_ck = off;
                        checkpoint(bs);
                        right_max = pmax8(0);
                        down_max = pmax8(2);
                        ck_trace_idx = trace_idx;
                        ck_block_idx = block_idx;
                        y_drop_iter = 0;
                    }
                }
                if (down_max > right_max) { st_i += B4_STEP; dir = B4_DOWN; }
                else { st_j += B4_STEP; dir = B4_RIGHT; }
            }
        }
        if (done) {
            // ---- align_core returned: the loop over minimum sizes (StripedSmithWaterman.cpp:1021-1038) ----
            bool too_large = overflow;
            if (!too_large) { score = best_max; ri = best_i; rj = best_j; }
            resume_min = too_large ? min_size : min_size * 2;
            min_size *= 2;
            if (!too_large && score < target && min_size <= MAXB) {
                start_attempt();
            } else {
                // (the crate would go on to larger minimum sizes when the score is not reached - not decided by this kernel)
                if (MAXB < BLOCK_REF_MAX_SIZE && !too_large && score < target) too_large = true;
                finish_pair(too_large);
                fetch();
                if (!active) return;
                start_attempt();
            }
        }
        begin_step();
    };

    fetch();
    if (active) {
        start_attempt();
        begin_step();
    }
    if (SKEW) {
        // Eight pipeline stages on four rows: stage s = column s of the octet, one chunk behind stage s - 1.  Row r carries stage r in
        // the LOW int16 half of its registers and stage r + 4 in the HIGH half - the packed operations, the DPP moves and the
        // carries serve both at once; what differs per half (the chunk, the rows' letters, the column, where results go) is a
        // few unpacked instructions around them.  An octet of n chunks takes n + 7 trips.
        const int r = lane >> 4;
        typedef unsigned short b4_u16x2 __attribute__((ext_vector_type(2)));
        auto ne2 = [](unsigned a, unsigned b) {      // per half: 0 where equal, 1 where not (v_pk_min_u16)
            return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(b4_u16x2, a ^ b), __builtin_bit_cast(b4_u16x2, 0x00010001u)));
        };
        while (active) {      // (uniform over the wavefront: one pair)
            const int nch = J_height >> 4;
            const bool first = J_oct == 0;
            const unsigned offp = (J_kind == B4_JOB_SHIFT && first) ? b4_pat(J_off_add) : 0u;
            int16_t *const rA = A + (J_rq ? 0 : 2 * MAXB), *const rB = rA + MAXB;
            const int rlen = J_rq ? Qlen : Tlen, rend = J_rq ? q_end : t_end, clen = J_rq ? Tlen : Qlen, cend = J_rq ? t_end : q_end;
            const uint8_t *const rbase = J_rq ? q_ptr() : t_ptr(), *const cbase = J_rq ? t_ptr() : q_ptr();
            const int ntrip = nch + 7;
            const bool ok = !TRACE || trace_idx + (uint32_t)nch * 128u <= trace_cap;
            uint8_t *const trace = L.pool + pool_off + blocks_bytes;
            if (lane < 8) {      // the octet's columns, through LDS
                const int pc = J_start_col + J_oct * 8 + lane;
                const bool in = pc >= 1 && pc <= clen;
                const int idx = in ? cend - (pc - 1) : 0;
                const int letter = in ? (int)cbase[idx] : 26;
                const int bias = (int)L.q_cb[Qoff + (J_rq ? 0 : idx)];
                s_col[0][lane] = (unsigned)letter * 128u;
                s_col[1][lane] = (!J_rq && in) ? b4_pat(bias) : 0u;
            }
            const unsigned soff_lo = s_col[0][r], soff_hi = s_col[0][r + 4];
            const unsigned cbc = s_col[1][r] | (s_col[1][r + 4] << 16);
            unsigned cR = 0, cT = 0x00010001u, cD15 = 0;      // from the chunk above, per half: R's last lane, trace_R false (as "not"), D10's last lane
            // the rows' letters and biases one trip ahead of their use (a trip is short, a load is not): low half chunk c, high half chunk c - 4
            unsigned n_ri_lo, n_ri_hi, n_rowb;
            auto load_rows = [&](int c_lo) {
                unsigned ri[2], rb[2];
                for (int h = 0; h < 2; h++) {
                    const int c = c_lo - 4 * h;
                    const int p = J_start_row + 16 * (c < 0 ? 0 : c) + k;
                    const bool in = p >= 1 && p <= rlen;
                    const int idx = in ? rend - (p - 1) : 0;
                    const int letter = (int)rbase[idx];
                    const int rbias = (int)L.q_cb[Qoff + (J_rq ? idx : 0)];
                    ri[h] = (unsigned)((in ? letter : 26) & 31) * 4u;
                    rb[h] = (J_rq && in) ? b4_pat(rbias) : 0u;
                }
                n_ri_lo = ri[0]; n_ri_hi = ri[1]; n_rowb = rb[0] | (rb[1] << 16);
            };
            load_rows(0);
            for (int t = 0; t < ntrip; t++) {
                const int c_lo = t - r, c_hi = c_lo - 4;
                const bool v_lo = c_lo >= 0 && c_lo < nch, v_hi = c_hi >= 0 && c_hi < nch;
                if (v_lo || v_hi) {
                    const unsigned ri_lo = n_ri_lo, ri_hi = n_ri_hi, rowb = n_rowb;
                    load_rows(c_lo + 1);
                    if (c_lo == 0) {      // the low half's column begins
                        cR &= 0xFFFF0000u; cT |= 0x1u;
                        cD15 = (cD15 & 0xFFFF0000u) | ((r == 0 && first) ? b4_pat(J_corner) : 0u);
                    }
                    if (c_hi == 0) { cR &= 0xFFFFu; cT |= 0x10000u; cD15 &= 0xFFFFu; }
                    const int x_lo = 16 * (v_lo ? c_lo : 0) + k, x_hi = 16 * (v_hi ? c_hi : 0) + k;
                    const int sl = (t + 1) & 1;
                    unsigned D10, C10;
                    if (r == 0) {      // column 0 out of the border arrays, column 4 after row 3's column 3
                        D10 = b4_adds((unsigned)(uint16_t)rA[x_lo], offp) | (s_ring[3][sl][0][k] << 16);
                        C10 = b4_adds((unsigned)(uint16_t)rB[x_lo], offp) | (s_ring[3][sl][1][k] << 16);
                    } else {
                        D10 = s_ring[r - 1][sl][0][k];
                        C10 = s_ring[r - 1][sl][1][k];
                    }
                    const unsigned D00 = b4_dpp<0x111>(cD15, D10);
                    cD15 = b4_bcast<15>(D10);
                    const unsigned sc = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_sc) + ri_lo + soff_lo) |
                                        (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_sc) + ri_hi + soff_hi) << 16);
                    unsigned D11 = b4_adds(D00, b4_adds(sc, b4_adds(cbc, rowb)));
                    if (J_origin && first && r == 0 && c_lo == 0 && k == 0) D11 = (D11 & 0xFFFF0000u) | (unsigned)B4_ZERO;
                    const unsigned C11o = b4_adds(D10, gop);
                    const unsigned C11 = b4_maxs(b4_adds(C10, gep), C11o);
                    D11 = b4_maxs(D11, C11);
                    const unsigned D11o = b4_adds(D11, gome);
                    const unsigned p1 = b4_maxs(D11o, b4_adds(b4_mov<0x111>(D11o) & m1, g1));
                    const unsigned p2 = b4_maxs(p1, b4_adds(b4_mov<0x112>(p1) & m2, g2));
                    const unsigned p4 = b4_maxs(p2, b4_adds(b4_dpp<0x114, 0xB>(0u, p2), g4));
                    const unsigned from = b4_dpp<0x157, 0xC>(b4_dpp<0x114>(p4, p4), p4);
                    unsigned R11 = b4_maxs(p4, b4_adds(from, consts));
                    R11 = b4_maxs(R11, b4_adds(cR, gap_all));
                    cR = b4_bcast<15>(R11);
                    D11 = b4_maxs(D11, R11);
                    if (TRACE) {      // a byte per cell, stored as "not": D11 != C11, D11 != R11, C11 != C11_open, trace_R false
                        const unsigned nT = ne2(R11, D11o);
                        const unsigned ntr = b4_dpp<0x111>(cT, nT);
                        cT = b4_bcast<15>(nT);
                        const unsigned nib = (ne2(D11, C11) << 3) | (ne2(D11, R11) << 2) | (ne2(C11, C11o) << 1) | ntr;
                        if (ok && v_lo) trace[trace_idx + (uint32_t)((c_lo * 8 + r) * 16 + k)] = (uint8_t)(nib & 0xFu);
                        if (ok && v_hi) trace[trace_idx + (uint32_t)((c_hi * 8 + r + 4) * 16 + k)] = (uint8_t)((nib >> 16) & 0xFu);
                    }
                    if (v_lo) Mkey = max(Mkey, make_key(D11 & 0xFFFFu, (unsigned)(J_oct * 8 + r), (unsigned)c_lo));
                    if (v_hi) Mkey = max(Mkey, make_key(D11 >> 16, (unsigned)(J_oct * 8 + r + 4), (unsigned)c_hi));
                    s_ring[r][t & 1][0][k] = D11;
                    s_ring[r][t & 1][1][k] = C11;
                    if (r == 3 && v_hi) { rA[x_hi] = (int16_t)(D11 >> 16); rB[x_hi] = (int16_t)(C11 >> 16); }      // column 7: the border arrays, in place
                    if (k == 15) {
                        if (c_lo == nch - 1) { s_out[0][r] = (int16_t)D11; s_out[1][r] = (int16_t)R11; }
                        if (c_hi == nch - 1) { s_out[0][r + 4] = (int16_t)(D11 >> 16); s_out[1][r + 4] = (int16_t)(R11 >> 16); }
                    }
                }
            }
            if (TRACE) {
                overflow = overflow || !ok;
                slot_overflow = slot_overflow || !ok;
                trace_idx += (uint32_t)nch * 128u;
            }
            if (J_kind != B4_JOB_SHIFT) {
                int16_t *const o1 = A + (J_kind == B4_JOB_GROW1 ? 0 : 2) * MAXB;
                put_outputs(o1, o1 + MAXB, J_out_base + J_oct * 8);
            }
            J_oct++;
            if (J_oct * 8 >= J_width) advance();
        }
        return;
    }
    // Every trip: two chunks of the row's octet (block sizes are multiples of 32 rows, so rows at 32-row blocks - three shifts of
    // four - complete an octet and take their align_core step on every trip, together; a row at larger blocks needs more trips for
    // its octet and holds nobody up)
    while (active) {
        if (J_c == 0) octet_setup();
        chunk_pass(J_c);
        chunk_pass(J_c + 1);
        J_c += 2;
        if (J_c * 16 >= J_height) {
            if (TRACE) {
                slot_overflow = slot_overflow || trace_idx + (uint32_t)(J_height >> 4) * 16u > trace_cap;
                overflow = overflow || slot_overflow;
                trace_idx += (uint32_t)(J_height >> 4) * 16u;
            }
            if (J_kind != B4_JOB_SHIFT) {
                // the first grow job's last row -> D_col / C_col entries, the second's -> D_row / R_row
                int16_t *const o1 = A + (J_kind == B4_JOB_GROW1 ? 0 : 2) * MAXB;
                put_outputs(o1, o1 + MAXB, J_out_base + J_oct * 8);
            }
            J_oct++;
            J_c = 0;
            if (J_oct * 8 >= J_width) advance();
        }
    }
}

// Trace::cigar_core (scan_block.rs:1844-2006) for the pairs sw_block4_kernel<true> answered MMGPU_BLOCK_OK: one LANE per pair walks
// from the end cell to the origin; identities, the string (forward order: the reference reverses twice, :1071-1110) and its length.
__global__ __launch_bounds__(64) void sw_block4_walk_kernel(Block2Launch L) {
    const uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    if (idx >= L.n_jobs) return;
    const Block2Job J = L.jobs[idx];
    mmgpu_sw_block out = L.out[J.slot];
    if (out.status != MMGPU_BLOCK_OK) return;
    int i = (int)out.ident, j = (int)out.bt_len;
    uint32_t block_idx = (uint32_t)out.reserved;
    const int qa = J.q_end + 1, ta = J.t_end + 1;
    const uint32_t block_cap = (uint32_t)(qa + ta + 64);
    const uint32_t blocks_bytes = (block_cap * (uint32_t)sizeof(BkBlock) + 31u) & ~31u;
    const BkBlock *blocks = reinterpret_cast<const BkBlock *>(L.pool + J.pool_off);
    const uint32_t *trace = reinterpret_cast<const uint32_t *>(L.pool + J.pool_off + blocks_bytes);
    const uint8_t *q = L.q_res + L.q_off[J.query];
    const uint8_t *t = L.t_res + (size_t)L.t_off4[J.target] * 4;
    char *bt = L.bt ? L.bt + out.bt_off : nullptr;      // bt_off is a multiple of four
    int table = 0;      // 0 = D, 1 = C, 2 = R
    uint32_t n = 0, ids = 0, word = 0;
    // one flat loop - an iteration either fetches the next block of the list or takes one step inside the current one - so that the
    // lanes of a wavefront, each somewhere else in its own walk, share every trip (nested loops would serialise them)
    int bi = 0x7FFFFFFF, bj = 0x7FFFFFFF, bright = 0;
    uint32_t btstart = 0, nch = 2;
    while (i > 0 || j > 0) {
        if (!(i >= bi && j >= bj)) {
            block_idx--;
            const BkBlock b = blocks[block_idx];
            bi = (int)b.i; bj = (int)b.j; bright = (int)b.right; btstart = b.tstart;
            nch = (uint32_t)(bright ? b.h : b.w) >> 4;      // the job's rows (place_block's height) in chunks of 16
            continue;
        }
        const int ci = i - bi, cj = j - bj;
        const int r = bright ? ci : cj, c = bright ? cj : ci;
        unsigned nib;      // D == C, D == R, C == C_open, trace_R
        if (L.trace_bytes) {      // the skewed form's trace: a byte per cell, [octet][chunk][column][lane]
            nib = (unsigned)reinterpret_cast<const uint8_t *>(trace)[btstart + (((uint32_t)(c >> 3) * nch + (uint32_t)(r >> 4)) * 8u + (uint32_t)(c & 7)) * 16u + (uint32_t)(r & 15)] ^ 0xFu;
        } else {
            const uint32_t w = trace[btstart + ((uint32_t)(c >> 3) * nch + (uint32_t)(r >> 4)) * 16u + (uint32_t)(r & 15)];
            nib = (w >> (28 - 4 * (c & 7))) & 0xFu;
        }
        const unsigned tt = ((nib >> 3) & 1u) | (((nib >> 2) & 1u) << 1), t2 = ((nib >> 1) & 1u) | ((nib & 1u) << 1);
        int op, nt;      // OP_LUT (:1870-1933): 1 = match / mismatch, 4 = the row index moves (I), 5 = the column index moves (D)
        if (bright) {
            if (table == 1) { op = 5; nt = (t2 & 1u) ? 0 : 1; }
            else if (table == 2) { op = 4; nt = (t2 & 2u) ? 0 : 2; }
            else if (tt == 0) { op = 1; nt = 0; }
            else if (tt & 1u) { op = 5; nt = (t2 & 1u) ? 0 : 1; }
            else { op = 4; nt = (t2 & 2u) ? 0 : 2; }
        } else {
            if (table == 2) { op = 4; nt = (t2 & 1u) ? 0 : 2; }
            else if (table == 1) { op = 5; nt = (t2 & 2u) ? 0 : 1; }
            else if (tt == 0) { op = 1; nt = 0; }
            else if (tt & 1u) { op = 4; nt = (t2 & 1u) ? 0 : 2; }
            else { op = 5; nt = (t2 & 2u) ? 0 : 1; }
        }
        if (op == 1) ids += q[J.q_end - (i - 1)] == t[J.t_end - (j - 1)] ? 1u : 0u;
        if (bt) {
            word |= (uint32_t)(op == 1 ? 'M' : (op == 4 ? 'I' : 'D')) << (8 * (n & 3u));
            if ((n & 3u) == 3u) { *reinterpret_cast<uint32_t *>(bt + (n & ~3u)) = word; word = 0; }
        }
        n++;
        i -= (op != 5) ? 1 : 0;
        j -= (op != 4) ? 1 : 0;
        table = nt;
    }
    if (bt && (n & 3u) != 0u) {
        for (uint32_t x = 0; x < (n & 3u); x++) bt[(n & ~3u) + x] = (char)(word >> (8 * x));
    }
    out.ident = ids;
    out.bt_len = n;
    out.reserved = 0;
    L.out[J.slot] = out;
}

}  // namespace

// form 0 / 1: four pairs per wavefront, blocks up to BLOCK4_MAX_SIZE / BLOCK4_LARGE_SIZE rows; form 2 / 3: the skewed form, one
// pair per wavefront, blocks up to BLOCK4_SKEW_SIZE / BLOCK_REF_MAX_SIZE rows
hipError_t launch_sw_block4(const Block2Launch &L, bool trace, int form, uint32_t n_waves, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    if (form == 3) {
        if (trace) hipLaunchKernelGGL((sw_block4_kernel<true, BLOCK_REF_MAX_SIZE, true>), dim3(n_waves), dim3(64), 0, stream, L);
        else hipLaunchKernelGGL((sw_block4_kernel<false, BLOCK_REF_MAX_SIZE, true>), dim3(n_waves), dim3(64), 0, stream, L);
    } else if (form == 2) {
        if (trace) hipLaunchKernelGGL((sw_block4_kernel<true, BLOCK4_SKEW_SIZE, true>), dim3(n_waves), dim3(64), 0, stream, L);
        else hipLaunchKernelGGL((sw_block4_kernel<false, BLOCK4_SKEW_SIZE, true>), dim3(n_waves), dim3(64), 0, stream, L);
    } else if (form == 1) {
        if (trace) hipLaunchKernelGGL((sw_block4_kernel<true, BLOCK4_LARGE_SIZE, false>), dim3(n_waves), dim3(64), 0, stream, L);
        else hipLaunchKernelGGL((sw_block4_kernel<false, BLOCK4_LARGE_SIZE, false>), dim3(n_waves), dim3(64), 0, stream, L);
    } else {
        if (trace) hipLaunchKernelGGL((sw_block4_kernel<true, BLOCK4_MAX_SIZE, false>), dim3(n_waves), dim3(64), 0, stream, L);
        else hipLaunchKernelGGL((sw_block4_kernel<false, BLOCK4_MAX_SIZE, false>), dim3(n_waves), dim3(64), 0, stream, L);
    }
    return hipGetLastError();
}

hipError_t launch_sw_block4_walk(const Block2Launch &L, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_block4_walk_kernel, dim3((L.n_jobs + 63) / 64), dim3(64), 0, stream, L);
    return hipGetLastError();
}

void warm_block4() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&(sw_block4_kernel<false, BLOCK4_LARGE_SIZE, false>)));
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&(sw_block4_kernel<true, BLOCK4_LARGE_SIZE, false>)));
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_block4_walk_kernel));
}

}  // namespace mmgpu
