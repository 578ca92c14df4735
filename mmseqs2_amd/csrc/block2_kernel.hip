// The block aligner for int16-range hits (row a15), TWO PAIRS PER WAVEFRONT: SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ>
// (src/alignment/StripedSmithWaterman.cpp:943-1127) -> Block<TRACE, X_DROP>::align_aa of lib/block-aligner 0.4.0, AVX2 configuration
// (scan_block.rs:120-632 align_core, :1449-1613 place_block, avx2.rs:294-337).  Same vector operations in the same order on the same
// saturating int16 values as block_kernel.hip (the one-pair-per-wavefront form, which stays for profile queries and for blocks
// beyond 128 rows) and as oracle/block_oracle.c; what changed is where things live:
//
//  * packed 2 x int16: the LOW half of every register is pair A, the HIGH half pair B.  v_pk_add_i16 clamp / v_pk_max_i16 /
//    DPP moves / readlanes serve both pairs with one instruction (block_kernel.hip: 32-bit registers, add + med3 per saturating add);
//  * lane = block row (64 rows per chunk = four of the crate's 16-lane vectors, two chunks = blocks up to 128 rows: 99 % of the
//    pairs of configs[2]); the block's border arrays D_col / C_col / D_row / R_row and their checkpoints are REGISTERS (lane x of
//    chunk c = entry 64 c + x), not LDS: a shift is one ds_bpermute per array, a checkpoint one v_bfi per array;
//  * the two pairs run in lock step by OCTETS of eight columns - every place_block job of the crate is a multiple of eight columns
//    wide (STEP = 8, block sizes are powers of two >= 32) - and each pair's align_core is a resumable state machine (`advance`)
//    that runs between octets: shift right / down, grow (two jobs), shrink, checkpoint, x-drop, the minimum sizes 32, 64, 128;
//  * wavefronts are persistent: a half that finishes a pair takes the next one off a global queue;
//  * the trace is four bits per cell in one dword per row and octet, written coalesced; the walk back is a second kernel with one
//    LANE per pair (sw_block2_walk_kernel) instead of a serial tail in every wavefront;
//  * TRACE = false (the caller wants start positions only - `mmseqs search` without -a, alignment mode 2, where neither the
//    identities nor the string reach the output, Matcher.cpp:107-127): no trace, no block list, no scratch memory, no walk.
// A pair this kernel does not decide (blocks would grow beyond 128 rows, its trace slot overflows) is answered
// MMGPU_BLOCK_TOO_LARGE and goes to block_kernel.hip's tiers.
#include "mmgpu_internal.h"

namespace mmgpu {

namespace {

constexpr int B2_STEP = 8, B2_ZERO = 16384, B2_MIN = 0, B2_X_DROP_ITER = 2;
constexpr unsigned B2_NEG2 = 0x80008000u, B2_ONE2 = 0x00010001u, B2_LO = 0x0000FFFFu, B2_HI = 0xFFFF0000u;

typedef short b2_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short b2_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) {      // v_pk_add_i16 clamp
    return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(b2_s16x2, a), __builtin_bit_cast(b2_s16x2, b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {      // v_pk_max_i16
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(b2_s16x2, a), __builtin_bit_cast(b2_s16x2, b)));
}
__device__ __forceinline__ unsigned pk_minu(unsigned a, unsigned b) {     // v_pk_min_u16
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(b2_u16x2, a), __builtin_bit_cast(b2_u16x2, b)));
}
__device__ __forceinline__ unsigned pk_subw(unsigned a, unsigned b) {     // v_pk_sub_u16 (wraps, per half)
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(b2_u16x2, a) - __builtin_bit_cast(b2_u16x2, b));
}
__device__ __forceinline__ unsigned b2_bfi(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ unsigned b2_splat(int v) { return ((unsigned)v & 0xFFFFu) | ((unsigned)v << 16); }
__device__ __forceinline__ int b2_adds16(int a, int b) { const int s = a + b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }
__device__ __forceinline__ int b2_subs16(int a, int b) { const int s = a - b; return s > 32767 ? 32767 : (s < -32768 ? -32768 : s); }
template <int P> __device__ __forceinline__ int b2_half(unsigned x) { return P ? ((int)x >> 16) : (int)(short)x; }
template <int P> __device__ __forceinline__ unsigned b2_put(unsigned old, unsigned nw) { return b2_bfi(P ? B2_HI : B2_LO, nw, old); }
template <int P> __device__ __forceinline__ int b2_shalf(int x) { return P ? (x >> 16) : (int)(short)x; }      // of a scalar (readlane result)

// simd_sllz_i16!(x, N): byte shift inside each 128-bit half (8 lanes), zeros shifted in (avx2.rs:152-164); `m` = all ones where
// (lane & 7) >= N
template <int N> __device__ __forceinline__ unsigned b2_sllz(unsigned x, unsigned m) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + N /* row_shr:N */, 0xF, 0xF, true) & m;
}
__device__ __forceinline__ unsigned b2_row_lane7(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x157 /* row_newbcast:7 */, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned b2_prev_row_lane15(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xE, 0xF, false);
}
__device__ __forceinline__ unsigned b2_shift_up1(unsigned v, unsigned first) {      // lane - 1's value; lane 0 receives `first`
    return (unsigned)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned b2_lane_from(unsigned v, int src_lane) {         // v of lane src_lane (per lane)
    return (unsigned)__builtin_amdgcn_ds_bpermute((src_lane & 63) << 2, (int)v);
}
__device__ __forceinline__ int b2_wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112 /* row_shr:2 */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114 /* row_shr:4 */, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118 /* row_shr:8 */, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

struct B2Seq {
    const uint8_t *res;   // forward sequence; reversed prefix position p (1-based DP index) is res[end - (p - 1)]
    const int8_t *bias;   // per-position bias of the forward sequence (nullptr = 0)
    int end, len;
};
// PaddedBytes::get / PosBias::get at DP index p: index 0 and everything past the end is the padding letter / bias 0
// (branch-free: a load under a per-lane condition is a divergent branch, and the compiler then treats the wave-uniform state that
// meets it at a join block as per-lane values - VGPRs and exec masks instead of SGPRs and scalar branches)
__device__ __forceinline__ int b2_letter(const B2Seq &s, int p) {
    const bool in = p >= 1 && p <= s.len;
    const int v = (int)s.res[in ? s.end - (p - 1) : 0];
    return in ? v : 26;
}
__device__ __forceinline__ int b2_bias(const B2Seq &s, int p) {      // s.bias != nullptr
    const bool in = p >= 1 && p <= s.len;
    const int v = (int)s.bias[in ? s.end - (p - 1) : 0];
    return in ? v : 0;
}

enum { B2_RIGHT = 0, B2_DOWN = 1, B2_GROW = 2 };
enum { B2_JOB_NONE = 0, B2_JOB_SHIFT = 1, B2_JOB_GROW1 = 2, B2_JOB_GROW2 = 3 };

// one place_block call of the crate (scan_block.rs:1449-1613), eight columns at a time
struct B2Job {
    int kind;
    int rq;             // rows run over the query (shift right, second grow job), else over the reference
    int start_row, start_col;      // DP indices of row 0 / column 0
    int width, height;
    int oct;            // octets done
    int corner;         // D_corner of the first column
    int out_base;       // grow jobs: where column j's outputs go in the other pair of arrays
    int origin;         // the cell (0, 0) is row 0 of column 0
    int off_add;        // shift jobs: just_offset of the job's own arrays, applied on the way into the first octet
};

struct B2Pair {
    B2Seq Q, T;
    int target;         // the forward scan's score
    uint32_t slot;      // index into out
    int active;
    // align_core (scan_block.rs:120-632)
    int best_max, best_i, best_j, prev_dir, dir, prev_size, block_size, off, off_max, y_drop_iter, x_drop_iter;
    int st_i, st_j, i_ck, j_ck, off_ck, D_corner, off_add;
    int min_size, x_drop, score, ri, rj, dbg_first, dbg_steps;
    uint32_t dbg_d0, dbg_d1;
    // Trace
    uint32_t trace_idx, block_idx, ck_trace_idx, ck_block_idx, trace_cap, block_cap;   // trace in dwords
    int overflow;
    uint32_t *trace;
    BkBlock *blocks;
    B2Job job;
};

// NCH = chunks of 64 rows the border arrays hold: blocks up to 64 NCH rows (NCH = 2: the launch every pair starts in; NCH = 8: the
// launch for what that one answered TOO_LARGE)
template <bool TRACE, int NCH>
__global__ __launch_bounds__(64) void sw_block2_kernel(Block2Launch L) {
    constexpr int MAXB = 64 * NCH;
    __shared__ int16_t s_sc[27 * 32];
    const int lane = (int)threadIdx.x;
    for (int k0 = 0; k0 < 27 * 32; k0 += 64)
        if (k0 + lane < 27 * 32) s_sc[k0 + lane] = (int16_t)L.scores[k0 + lane];
    __syncthreads();
    // ---- per-lane constants ----
    const int k16 = lane & 15, row = lane >> 4;
    const unsigned m1 = (lane & 7) >= 1 ? ~0u : 0u, m2 = (lane & 7) >= 2 ? ~0u : 0u, m4 = (lane & 7) >= 4 ? ~0u : 0u;
    const int go = L.gap_open, ge = L.gap_extend;
    unsigned gap_all2, consts2;
    {   // avx2.rs:294-309
        auto sllz = [&](int x, int n) { const int v = __shfl_up(x, n, 8); return (lane & 7) >= n ? v : 0; };
        const int s1 = b2_adds16(sllz(ge, 1), ge);
        const int s2 = b2_adds16(sllz(s1, 2), s1);
        const int s4 = b2_adds16(sllz(s2, 4), s2);
        const int w7 = __shfl(s4, (lane & ~15) + 7, 64);
        gap_all2 = b2_splat(b2_adds16(k16 < 8 ? 0 : w7, s4));
        consts2 = b2_splat(s4);
    }
    const unsigned g1 = b2_splat(ge), g2 = b2_splat((int)(short)(ge << 1)), g4 = b2_splat((int)(short)(ge << 2));
    const unsigned go2 = b2_splat(go), ge2 = b2_splat(ge), gome2 = b2_splat(b2_subs16(go, ge));

    // ---- vector state: border arrays of both pairs (low half A, high half B), entry 64 c + lane in chunk c ----
    // (vector VALUES, not C arrays: `right ? ROW1 : COL1` must be a select of registers - with arrays the compiler selects between their
    // addresses and the arrays land in scratch memory)
    typedef unsigned VN __attribute__((ext_vector_type(NCH)));
    typedef int IN __attribute__((ext_vector_type(NCH)));
    VN COL1 = 0u, COL2 = 0u, ROW1 = 0u, ROW2 = 0u;          // D_col, C_col, D_row, R_row
    VN KCOL1 = 0u, KCOL2 = 0u, KROW1 = 0u, KROW2 = 0u;      // their checkpoints
    VN rowbias = 0u, act = 0u;
    IN rowidx0 = 26 * 2, rowidx1 = 26 * 2;                 // per pair and chunk: the row's letter (as a byte offset into a score row)
    unsigned Mmax = 0, Gmax = 0, OUT1 = 0, OUT2 = 0;
    unsigned Marg = 0, Garg = 0;           // per half: column | chunk << 9 of the lane's LAST maximum (D_argmax: the chunk's base row is 64 c + (lane & ~15))
    unsigned colinfo[2] = {26, 26};        // per pair: lane l = column l of the job (mod 64): letter | bias << 8
    VN tacc0 = 0u, tacc1 = 0u;             // [chunk], first / second four columns: the octet's trace bits, packed
    B2Pair S0, S1;
    S0.active = 0; S1.active = 0;
    S0.job.kind = B2_JOB_NONE; S1.job.kind = B2_JOB_NONE;
    S0.job.height = 0; S1.job.height = 0; S0.job.width = 8; S1.job.width = 8; S0.job.oct = 0; S1.job.oct = 0;
    S0.job.rq = 1; S1.job.rq = 1; S0.job.corner = 0; S1.job.corner = 0; S0.job.origin = 0; S1.job.origin = 0;
    S0.job.start_row = 0; S1.job.start_row = 0; S0.job.start_col = 0; S1.job.start_col = 0; S0.job.out_base = 0; S1.job.out_base = 0;
    S0.job.off_add = 0; S1.job.off_add = 0;

    auto chunks_of = [](int h) { return h > 64 ? (h + 63) >> 6 : 1; };

    // ---- helpers on one pair's half of the arrays ----
    auto for_arrays = [&](auto f) { f(COL1, KCOL1); f(COL2, KCOL2); f(ROW1, KROW1); f(ROW2, KROW2); };

    // the job's rows and columns: letters and biases (loaded once per job)
    auto load_job = [&](auto PC, B2Pair &S) {
        constexpr int P = decltype(PC)::value;
        const B2Job &J = S.job;
        const B2Seq &rows = J.rq ? S.Q : S.T, &cols = J.rq ? S.T : S.Q;
        const int nch = chunks_of(J.height);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            if (c < nch) {
                const int p = J.start_row + c * 64 + lane;
                if (P) rowidx1[c] = b2_letter(rows, p) * 2;
                else rowidx0[c] = b2_letter(rows, p) * 2;
                rowbias[c] = b2_put<P>(rowbias[c], (J.rq ? (unsigned)b2_bias(S.Q, p) : 0u) << (P ? 16 : 0));
                const int x = c * 64 + lane;
                act[c] = b2_put<P>(act[c], x < J.height ? ~0u : 0u);
            } else {
                act[c] = b2_put<P>(act[c], 0u);
            }
        }
        const int pc = J.start_col + lane;
        colinfo[P] = (unsigned)b2_letter(cols, pc) | (((J.rq ? 0u : (unsigned)b2_bias(S.Q, pc)) & 0xFFu) << 8);
    };
    auto reload_cols = [&](auto PC, B2Pair &S) {      // jobs wider than 64 columns: the next 64
        constexpr int P = decltype(PC)::value;
        const B2Job &J = S.job;
        const B2Seq &cols = J.rq ? S.T : S.Q;
        const int pc = J.start_col + J.oct * 8 + lane;
        colinfo[P] = (unsigned)b2_letter(cols, pc) | (((J.rq ? 0u : (unsigned)b2_bias(S.Q, pc)) & 0xFFu) << 8);
    };

    auto add_block = [&](B2Pair &S, int i, int j, int width, int height, int right) {
        if (!TRACE) return;
        const bool ok = S.block_idx < S.block_cap;      // (uniform state changes outside the per-lane branch)
        if (ok && S.active && lane == 0) {
            BkBlock b;
            b.i = (uint32_t)i; b.j = (uint32_t)j; b.h = (uint16_t)height; b.w = (uint16_t)width; b.right = (uint32_t)right; b.tstart = S.trace_idx;
            S.blocks[S.block_idx] = b;
        }
        S.overflow |= ok ? 0 : 1;
        S.block_idx++;
    };

    // prefix_max: the maximum of the first STEP entries of an array (scan_block.rs:1125-1140)
    auto pmax8 = [&](auto PC, unsigned a0) {
        constexpr int P = decltype(PC)::value;
        int v = lane < B2_STEP ? b2_half<P>(a0) : -32768;
        v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xF, 0xF, false));
        v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xF, 0xF, false));
        v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xF, false));
        return __builtin_amdgcn_readlane(v, 7);
    };
    // entry x of an array (uniform x)
    auto entry = [&](auto PC, const VN &a, int x) {
        constexpr int P = decltype(PC)::value;
        int v = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++)
            if (c == (x >> 6)) v = __builtin_amdgcn_readlane((int)a[c], x & 63);
        return b2_shalf<P>(v);
    };
    // shift_and_offset (:1102-1123): entries move down by STEP and take off_add, the last STEP entries are the job's outputs.
    // (by value: which pair of arrays is shifted depends on the direction, and a select between array ADDRESSES would put them in scratch)
    auto shift_one = [&](auto PC, VN a, unsigned out, int off_add, int bs) {
        constexpr int P = decltype(PC)::value;
        const unsigned o2 = b2_splat(off_add);
        const unsigned o = b2_lane_from(out, lane - (bs - B2_STEP));
        unsigned t[NCH + 1];
#pragma unroll
        for (int c = 0; c < NCH; c++) t[c] = c * 64 < bs ? b2_lane_from(a[c], lane + 8) : 0u;
        t[NCH] = 0u;
        VN r = a;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            if (c * 64 < bs) {
                unsigned n = pk_add(lane < 56 ? t[c] : t[c + 1], o2);
                const int x = c * 64 + lane;
                if (x >= bs - B2_STEP) n = o;
                r[c] = b2_put<P>(a[c], x < bs ? n : a[c]);
            }
        }
        return r;
    };
    // a grow job's outputs of one octet: entries base .. base + 7 of the other pair of arrays
    auto place_one = [&](auto PC, unsigned a, int c, unsigned out, int base) {
        constexpr int P = decltype(PC)::value;
        const unsigned o = b2_lane_from(out, lane - base);
        const unsigned d = (unsigned)(c * 64 + lane - base);
        return d < (unsigned)B2_STEP ? b2_put<P>(a, o) : a;
    };

    // ---- eight columns of both pairs' jobs ----
    auto octet = [&]() {
        const B2Job &JA = S0.job, &JB = S1.job;
        const int nch = max(S0.active ? chunks_of(JA.height) : 1, S1.active ? chunks_of(JB.height) : 1);
        const unsigned rqm = (JA.rq ? B2_LO : 0u) | (JB.rq ? B2_HI : 0u);
        const bool firstA = JA.oct == 0, firstB = JB.oct == 0;
        // just_offset (scan_block.rs:1102-1123, the branch without a shift) of a shift job's own arrays: here, on the way in
        const unsigned offpk = ((unsigned)((JA.kind == B2_JOB_SHIFT && firstA) ? JA.off_add : 0) & B2_LO) |
                               ((unsigned)((JB.kind == B2_JOB_SHIFT && firstB) ? JB.off_add : 0) << 16);
        VN D10, C10;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            D10[c] = pk_add(b2_bfi(rqm, COL1[c], ROW1[c]), offpk);
            C10[c] = pk_add(b2_bfi(rqm, COL2[c], ROW2[c]), offpk);
        }
        const int lastA = (JA.height - 1) & 63, lastB = (JB.height - 1) & 63;
        const int lastcA = JA.height > 0 ? (JA.height - 1) >> 6 : 0, lastcB = JB.height > 0 ? (JB.height - 1) >> 6 : 0;
        const int colA = (JA.oct * 8) & 63, colB = (JB.oct * 8) & 63;
        const unsigned jpk = ((unsigned)(JA.oct * 8) & B2_LO) | ((unsigned)(JB.oct * 8) << 16);
#pragma unroll 1
        for (int k2 = 0; k2 < 2; k2++) {
            VN acc = 0u;
#pragma unroll 1
            for (int kq = 0; kq < 4; kq++) {
                const int kk = k2 * 4 + kq;
                const int ciA = __builtin_amdgcn_readlane((int)colinfo[0], colA + kk), ciB = __builtin_amdgcn_readlane((int)colinfo[1], colB + kk);
                const int soffA = (ciA & 0xFF) * 64, soffB = (ciB & 0xFF) * 64;
                const unsigned cbias = ((unsigned)(int)(int8_t)(ciA >> 8) & 0xFFFFu) | ((unsigned)(int)(int8_t)(ciB >> 8) << 16);
                unsigned corner = (kk == 0) ? (((unsigned)(firstA ? JA.corner : B2_MIN) & 0xFFFFu) | ((unsigned)(firstB ? JB.corner : B2_MIN) << 16)) : 0u;
                unsigned carryR = 0u /* MIN, MIN */, carry_tr = B2_ONE2 /* trace_R false */;
                const unsigned cur0 = jpk + (unsigned)kk * 0x00010001u;      // per half: this column
                int dA = 0, dB = 0, rA = 0, rB = 0;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    if (c < nch) {
                        const bool more = c + 1 < nch;
                        const unsigned D00 = b2_shift_up1(D10[c], corner);
                        if (more) corner = (unsigned)__builtin_amdgcn_readlane((int)D10[c], 63);
                        const int scA = *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(s_sc) + soffA + rowidx0[c]);
                        const int scB = *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(s_sc) + soffB + rowidx1[c]);
                        const unsigned sc = ((unsigned)scA & 0xFFFFu) | ((unsigned)scB << 16);
                        const unsigned pb = pk_add(cbias, rowbias[c]);
                        unsigned D11 = pk_add(D00, pk_add(sc, pb));
                        if (kk == 0 && c == 0) {      // the cell (0, 0)
                            const unsigned om = (lane == 0) ? (((JA.origin && firstA) ? B2_LO : 0u) | ((JB.origin && firstB) ? B2_HI : 0u)) : 0u;
                            D11 = b2_bfi(om, b2_splat(B2_ZERO), D11);
                        }
                        const unsigned C11o = pk_add(D10[c], go2);
                        const unsigned C11 = pk_max(pk_add(C10[c], ge2), C11o);
                        D11 = pk_max(D11, C11);
                        const unsigned D11o = pk_add(D11, gome2);
                        // simd_prefix_scan_i16 (avx2.rs:311-337) on every 16-lane row
                        const unsigned p1 = pk_max(D11o, pk_add(b2_sllz<1>(D11o, m1), g1));
                        const unsigned p2 = pk_max(p1, pk_add(b2_sllz<2>(p1, m2), g2));
                        const unsigned p4 = pk_max(p2, pk_add(b2_sllz<4>(p2, m4), g4));
                        const unsigned down4 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p4, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
                        const unsigned l7 = b2_row_lane7(p4);
                        const unsigned from = k16 < 4 ? p4 : (k16 < 8 ? down4 : l7);
                        unsigned R11 = pk_max(p4, pk_add(from, consts2));
                        // R11 = max(R11, broadcast(R01's last lane) + gap_extend_all), vector by vector: row 0 takes the carry of the chunk
                        // before, row ch the last lane of row ch - 1 (the DPP move writes row ch only; -32768 + gap stays -32768)
                        {
                            const unsigned add = pk_add(carryR, gap_all2);
                            R11 = pk_max(R11, row == 0 ? add : B2_NEG2);
                        }
                        R11 = pk_max(R11, pk_add((unsigned)__builtin_amdgcn_update_dpp((int)B2_NEG2, (int)R11, 0x142 /* row_bcast:15 */, 0x2, 0xF, false), gap_all2));
                        R11 = pk_max(R11, pk_add((unsigned)__builtin_amdgcn_update_dpp((int)B2_NEG2, (int)R11, 0x142, 0x4, 0xF, false), gap_all2));
                        R11 = pk_max(R11, pk_add((unsigned)__builtin_amdgcn_update_dpp((int)B2_NEG2, (int)R11, 0x142, 0x8, 0xF, false), gap_all2));
                        if (more) carryR = (unsigned)__builtin_amdgcn_readlane((int)R11, 63);
                        D11 = pk_max(D11, R11);
                        if (TRACE) {
                            const unsigned n1 = pk_minu(D11 ^ C11, B2_ONE2), n2 = pk_minu(D11 ^ R11, B2_ONE2);
                            const unsigned n3 = pk_minu(C11 ^ C11o, B2_ONE2), nt = pk_minu(R11 ^ D11o, B2_ONE2);
                            const unsigned ntr = b2_shift_up1(nt, carry_tr);
                            if (more) carry_tr = (unsigned)__builtin_amdgcn_readlane((int)nt, 63);
                            const unsigned nib = n1 | (n2 << 1) | (n3 << 2) | (ntr << 3);
                            acc[c] = (acc[c] << 4) | nib;
                        }
                        {   // D_max / D_argmax of the lane (its rows of all chunks): the LAST cell that holds the maximum
                            const unsigned D11m = b2_bfi(act[c], D11, B2_NEG2);
                            const unsigned nm = pk_max(Mmax, D11m);
                            const unsigned eq = pk_subw(pk_minu(nm ^ D11m, B2_ONE2), B2_ONE2);      // per half: 0xFFFF where D11m is the maximum (0 - 1), 0 elsewhere (1 - 1)
                            Marg = b2_bfi(eq, cur0 + (unsigned)c * 0x02000200u, Marg);
                            Mmax = nm;
                        }
                        D10[c] = D11;
                        C10[c] = C11;
                        // D_row[j] / R_row[j]: the block's last row
                        if (c == lastcA) { dA = __builtin_amdgcn_readlane((int)D11, lastA); rA = __builtin_amdgcn_readlane((int)R11, lastA); }
                        if (c == lastcB) { dB = __builtin_amdgcn_readlane((int)D11, lastB); rB = __builtin_amdgcn_readlane((int)R11, lastB); }
                    }
                }
                if (lane == kk) OUT1 = ((unsigned)dA & B2_LO) | ((unsigned)dB & B2_HI);
                if (lane == kk) OUT2 = ((unsigned)rA & B2_LO) | ((unsigned)rB & B2_HI);
            }
            if (TRACE) {
                if (k2 == 0) tacc0 = acc;
                else tacc1 = acc;
            }
        }
        // back into the arrays the rows came from
        const unsigned ma = ((S0.active ? B2_LO : 0u) | (S1.active ? B2_HI : 0u));
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            if (c < nch) {
                const unsigned mc = act[c] & ma & rqm, mr = act[c] & ma & ~rqm;
                COL1[c] = b2_bfi(mc, D10[c], COL1[c]);
                COL2[c] = b2_bfi(mc, C10[c], COL2[c]);
                ROW1[c] = b2_bfi(mr, D10[c], ROW1[c]);
                ROW2[c] = b2_bfi(mr, C10[c], ROW2[c]);
            }
        }
    };

    // the octet's trace bits of one pair: one dword per row (lane) and chunk; nibble of column k at bits 12 - 4 k (k < 4) / 28 - 4 (k - 4):
    // bit 0 D11 != C11, bit 1 D11 != R11, bit 2 C11 != C11_open, bit 3 trace_R false
    auto store_trace = [&](auto PC, B2Pair &S) {
        constexpr int P = decltype(PC)::value;
        const B2Job &J = S.job;
        const int nch = chunks_of(J.height);
        const uint32_t stride = J.height < 64 ? 32u : 64u;
        const bool ok = S.trace_idx + (uint32_t)nch * stride <= S.trace_cap;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const unsigned t0 = tacc0[c], t1 = tacc1[c];
            const unsigned w = P ? ((t0 >> 16) | (t1 & B2_HI)) : ((t0 & B2_LO) | (t1 << 16));
            if (ok && c < nch && (uint32_t)lane < stride) S.trace[S.trace_idx + (uint32_t)c * stride + (uint32_t)lane] = w;
        }
        S.overflow |= ok ? 0 : 1;
        S.trace_idx += (uint32_t)nch * stride;
    };

    // ---- one pair's align_core between two jobs ----
    // the next pair of the queue.  Branch-free on purpose: past the end of the queue the half re-reads the last job and runs on as a
    // masked-off dummy (S.active = 0 keeps it out of every store), because "same store on both sides of a branch" is what the compiler
    // sinks into one store through a selected ADDRESS - and state addressed that way stays in scratch memory instead of registers
    auto fetch = [&](B2Pair &S) -> bool {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(L.counter, 1u);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        const bool ok = idx < L.n_jobs;
        const Block2Job J = L.jobs[ok ? idx : L.n_jobs - 1];
        const uint32_t query = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.query), target = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.target);
        S.target = __builtin_amdgcn_readfirstlane(J.score);
        S.slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.slot);
        const int q_end = __builtin_amdgcn_readfirstlane(J.q_end), t_end = __builtin_amdgcn_readfirstlane(J.t_end);
        const uint32_t qo = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.q_off[query]);
        const uint32_t to4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.t_off4[target]);
        S.Q.res = L.q_res + qo; S.Q.bias = L.q_cb + qo; S.Q.end = q_end; S.Q.len = q_end + 1;
        S.T.res = L.t_res + (size_t)to4 * 4; S.T.bias = nullptr; S.T.end = t_end; S.T.len = t_end + 1;
        if (TRACE) {
            const uint64_t po = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(J.pool_off >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)J.pool_off);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.pool_bytes);
            const uint32_t cap = (uint32_t)(S.Q.len + S.T.len + 64);
            const uint32_t blocks_bytes = (cap * (uint32_t)sizeof(BkBlock) + 31u) & ~31u;
            S.blocks = reinterpret_cast<BkBlock *>(L.pool + po);
            S.trace = reinterpret_cast<uint32_t *>(L.pool + po + blocks_bytes);
            S.trace_cap = pb > blocks_bytes ? (pb - blocks_bytes) / 4u : 0u;
            S.block_cap = pb > blocks_bytes ? cap : 0u;
        } else {
            S.block_cap = 0; S.trace_cap = 0; S.blocks = nullptr; S.trace = nullptr;
        }
        S.active = ok ? 1 : 0;
        S.min_size = 32;
        S.score = -1000000000; S.ri = 0; S.rj = 0; S.dbg_first = -1; S.dbg_steps = 0; S.dbg_d0 = 0; S.dbg_d1 = 0;
        return ok;
    };

    auto start_attempt = [&](auto PC, B2Pair &S) {      // Allocated::clear + align_core's initial state
        constexpr int P = decltype(PC)::value;
        for_arrays([&](VN &a, VN &k) {
#pragma unroll
            for (int c = 0; c < NCH; c++) { a[c] = b2_put<P>(a[c], 0u); k[c] = b2_put<P>(k[c], 0u); }
        });
        OUT1 = b2_put<P>(OUT1, 0u); OUT2 = b2_put<P>(OUT2, 0u);
        S.best_max = 0; S.best_i = 0; S.best_j = 0;
        S.prev_dir = B2_GROW; S.dir = B2_GROW;
        S.prev_size = 0; S.block_size = S.min_size;
        S.off = 0; S.off_max = 0;
        S.y_drop_iter = 0; S.x_drop_iter = 0;
        S.st_i = 0; S.st_j = 0; S.i_ck = 0; S.j_ck = 0; S.off_ck = 0;
        S.D_corner = B2_MIN; S.off_add = 0;
        S.trace_idx = 0; S.block_idx = 0; S.ck_trace_idx = 0; S.ck_block_idx = 0;
        S.overflow = 0;
        S.x_drop = -(S.min_size * ge + go);
    };

    auto checkpoint = [&](auto PC) {
        constexpr int P = decltype(PC)::value;
        for_arrays([&](VN &a, VN &k) {
#pragma unroll
            for (int c = 0; c < NCH; c++) k[c] = b2_put<P>(k[c], a[c]);
        });
    };
    auto restore = [&](auto PC) {
        constexpr int P = decltype(PC)::value;
        for_arrays([&](VN &a, VN &k) {
#pragma unroll
            for (int c = 0; c < NCH; c++) a[c] = b2_put<P>(a[c], k[c]);
        });
    };

    auto set_job = [&](auto PC, B2Pair &S, int kind, int rq, int start_row, int start_col, int width, int height, int corner, int out_base, int off_add) {
        constexpr int P = decltype(PC)::value;
        B2Job &J = S.job;
        J.kind = kind; J.rq = rq; J.start_row = start_row; J.start_col = start_col; J.width = width; J.height = height;
        J.oct = 0; J.corner = corner; J.out_base = out_base; J.off_add = off_add;
        J.origin = (start_row == 0 && start_col == 0) ? 1 : 0;
        Mmax = b2_put<P>(Mmax, 0u /* MIN */);
        Marg = b2_put<P>(Marg, 0u);
        load_job(PC, S);
    };

    // the second grow job (and the only one of the first block): place_block(query, reference, st_i, st_j + prev_size, grow_step, block_size)
    auto begin_grow2 = [&](auto PC, B2Pair &S) {
        add_block(S, S.st_i, S.st_j + S.prev_size, S.block_size - S.prev_size, S.block_size, 1);
        set_job(PC, S, B2_JOB_GROW2, 1, S.st_i, S.st_j + S.prev_size, S.block_size - S.prev_size, S.block_size, B2_MIN, S.prev_size, 0);
    };

    // top of align_core's loop: the next job of S.dir
    auto begin_step = [&](auto PC, B2Pair &S) {
        constexpr int P = decltype(PC)::value;
        if (S.dbg_first < 0) {      // debugging: the first attempt's directions, two bits a step
            if (S.dbg_steps < 16) S.dbg_d0 |= (uint32_t)S.dir << (2 * S.dbg_steps);
            else if (S.dbg_steps < 32) S.dbg_d1 |= (uint32_t)S.dir << (2 * (S.dbg_steps - 16));
        }
        const int prev_off = S.off;
        Gmax = b2_put<P>(Gmax, 0u);
        Garg = b2_put<P>(Garg, 0u);
        const int bs = S.block_size;
        if (S.dir == B2_RIGHT || S.dir == B2_DOWN) {
            const bool r = S.dir == B2_RIGHT;
            S.off = S.off_max;
            const int d = prev_off - S.off;
            S.off_add = d < -32768 ? -32768 : (d > 32767 ? 32767 : d);
            // (just_offset of the job's own arrays happens on the way into the octet: B2Job::off_add)
            add_block(S, r ? S.st_i : S.st_i + bs - B2_STEP, r ? S.st_j + bs - B2_STEP : S.st_j, r ? B2_STEP : bs, r ? bs : B2_STEP, r ? 1 : 0);
            const int corner = S.prev_dir == (r ? B2_DOWN : B2_RIGHT) ? b2_adds16(S.D_corner, S.off_add) : B2_MIN;
            set_job(PC, S, B2_JOB_SHIFT, r ? 1 : 0, r ? S.st_i : S.st_j, (r ? S.st_j : S.st_i) + bs - B2_STEP, B2_STEP, bs, corner, 0, S.off_add);
        } else {
            S.D_corner = B2_MIN;
            const int grow_step = bs - S.prev_size;
            add_block(S, S.st_i + S.prev_size, S.st_j, S.prev_size, grow_step, 0);
            if (S.prev_size > 0) set_job(PC, S, B2_JOB_GROW1, 0, S.st_j, S.st_i + S.prev_size, grow_step, S.prev_size, B2_MIN, S.prev_size, 0);
            else begin_grow2(PC, S);
        }
    };

    // the pair is decided: status, start positions; with TRACE the walk kernel finishes the record
    auto finish_pair = [&](B2Pair &S, bool too_large) {
        mmgpu_sw_block out;
        out.q_start = -1; out.t_start = -1; out.ident = 0; out.bt_len = 0; out.bt_off = L.bt_off ? L.bt_off[S.slot] : 0;
        out.reserved = 0;
        if (too_large) {
            out.status = MMGPU_BLOCK_TOO_LARGE;
        } else if (!(S.score != S.target && !(S.target == 32767 && S.score >= S.target))) {      // StripedSmithWaterman.cpp:1058
            out.status = MMGPU_BLOCK_OK;
            out.q_start = S.Q.end + 1 - S.ri;       // :1111-1112
            out.t_start = S.T.end + 1 - S.rj;
            if (TRACE) {      // for the walk kernel: end cell and number of blocks
                out.ident = (uint32_t)S.ri; out.bt_len = (uint32_t)S.rj; out.reserved = (int32_t)S.block_idx;
            }
        } else {
            out.status = MMGPU_BLOCK_DECLINED;
        }
        if (!TRACE) { out.reserved = S.dbg_first; out.ident = S.dbg_d0; out.bt_len = S.dbg_d1; }
        if (TRACE && L.growth != nullptr) {      // test aid: the block list of the last run
            __threadfence_block();
            uint32_t *g = L.growth + (size_t)S.slot * (1 + 4 * (size_t)L.growth_cap);
            const uint32_t n = too_large ? 0u : S.block_idx;
            if (lane == 0) g[0] = n;
            for (uint32_t k = (uint32_t)lane; k < n && k < L.growth_cap; k += 64) {
                const BkBlock bb = S.blocks[k];
                g[1 + 4 * k] = bb.i;
                g[2 + 4 * k] = bb.j;
                g[3 + 4 * k] = (uint32_t)bb.h << 16 | bb.w;
                g[4 + 4 * k] = bb.right;
            }
        }
        if (lane == 0) L.out[S.slot] = out;
    };

    // runs after the pair's job is complete (or at a fresh pair): everything up to the next job.  Leaves S.active = 0 when the queue is empty.
    // (no loop in here: a loop whose header joins "next attempt" and "next pair" with the entry makes the compiler copy the whole
    // vector state - some forty registers - into the header's registers and back on every call)
    auto advance = [&](auto PC, B2Pair &S, bool fresh) {
        constexpr int P = decltype(PC)::value;
        if (fresh) {
            start_attempt(PC, S);
        } else {
            B2Job &J = S.job;
            if (J.kind == B2_JOB_GROW1) {      // grow_D_max = this job's maxima; the second job follows
                Gmax = b2_put<P>(Gmax, Mmax);
                Garg = b2_put<P>(Garg, Marg);
                begin_grow2(PC, S);
                return;
            }
            int right_max, down_max;
            const int bs = S.block_size;
            if (J.kind == B2_JOB_SHIFT) {
                const bool r = S.dir == B2_RIGHT;      // the arrays the job did NOT run along are shifted: D_row / R_row after a shift right
                VN a1 = r ? ROW1 : COL1, a2 = r ? ROW2 : COL2;
                S.D_corner = b2_adds16(b2_shalf<P>(__builtin_amdgcn_readlane((int)a1[0], B2_STEP - 1)), S.off_add);
                a1 = shift_one(PC, a1, OUT1, S.off_add, bs);
                a2 = shift_one(PC, a2, OUT2, S.off_add, bs);
                ROW1 = r ? a1 : ROW1; ROW2 = r ? a2 : ROW2;
                COL1 = r ? COL1 : a1; COL2 = r ? COL2 : a2;
                right_max = pmax8(PC, COL1[0]);
                down_max = pmax8(PC, ROW1[0]);
            } else {
                right_max = pmax8(PC, COL1[0]);
                down_max = pmax8(PC, ROW1[0]);
                checkpoint(PC);
                S.ck_trace_idx = S.trace_idx;
                S.ck_block_idx = S.block_idx;
            }
            bool done = false;      // align_core returns
            S.dbg_steps++;
            if (S.overflow) {
                done = true;
            } else {
                const int dir = S.dir;
                S.prev_dir = dir;
                const int D_max_max = b2_wave_max(b2_half<P>(Mmax));
                const int grow_max = dir == B2_GROW ? b2_wave_max(b2_half<P>(Gmax)) : B2_MIN;
                const int mx = max(D_max_max, grow_max);
                S.off_max = S.off + mx - B2_ZERO;
                S.y_drop_iter++;
                if (TRACE && L.growth != nullptr && (L.dbg >> 8) == S.slot + 1 && S.dbg_first < 0 && S.dbg_steps <= 14) {      // debugging
                    uint32_t *g = L.growth + (size_t)S.slot * (1 + 4 * (size_t)L.growth_cap) + 1 + 2048 + 12 * (S.dbg_steps - 1);
                    if (lane == 0) {
                        g[0] = (uint32_t)dir; g[1] = (uint32_t)bs; g[2] = (uint32_t)S.off_max; g[3] = (uint32_t)S.best_max; g[4] = (uint32_t)mx; g[5] = (uint32_t)D_max_max;
                        g[6] = (uint32_t)grow_max; g[7] = (uint32_t)right_max; g[8] = (uint32_t)down_max; g[9] = (uint32_t)S.off; g[10] = (uint32_t)S.st_i; g[11] = (uint32_t)S.st_j;
                    }
                }
                bool grow_no_max = dir == B2_GROW;
                if (S.off_max > S.best_max) {
                    {   // location of the maximum: per vector lane the last cell that reached its maximum, over the vector lanes the
                        // largest column, then the largest row (:374-444)
                        const bool grow = dir == B2_GROW && D_max_max < grow_max;
                        const int curr_max = grow ? grow_max : D_max_max;
                        int dm = b2_half<P>(grow ? Gmax : Mmax);
                        const unsigned code = (unsigned)b2_half<P>(grow ? Garg : Marg) & 0xFFFFu;
                        int aj = (int)(code & 511u), ai = (int)((code >> 9) * 64u) + (lane & ~15);
#pragma unroll
                        for (int d = 16; d < 64; d <<= 1) {      // lanes l, l + 16, l + 32, l + 48 are one vector lane
                            const int odm = __shfl_xor(dm, d, 64), oai = __shfl_xor(ai, d, 64), oaj = __shfl_xor(aj, d, 64);
                            const bool take = odm > dm || (odm == dm && (oaj > aj || (oaj == aj && oai > ai)));
                            if (take) { dm = odm; ai = oai; aj = oaj; }
                        }
                        const int r = ai + k16, c = (bs - B2_STEP) + aj;
                        int gi, gj;
                        if (grow) { gi = S.st_i + S.prev_size + aj; gj = S.st_j + ai + k16; }
                        else if (dir == B2_RIGHT) { gi = S.st_i + r; gj = S.st_j + c; }
                        else if (dir == B2_DOWN) { gi = S.st_i + c; gj = S.st_j + r; }
                        else { gi = S.st_i + ai + k16; gj = S.st_j + S.prev_size + aj; }
                        long long key = (lane < 16 && dm == curr_max) ? (((long long)gj << 32) | (unsigned)gi) : 0ll;
                        for (int d = 1; d < 16; d <<= 1) {
                            const long long o = __shfl_xor(key, d, 64);
                            key = o > key ? o : key;
                        }
                        S.best_j = __builtin_amdgcn_readfirstlane((int)(key >> 32));
                        S.best_i = __builtin_amdgcn_readfirstlane((int)(key & 0xFFFFFFFFll));
                    }
                    if (bs < BLOCK_REF_MAX_SIZE) {
                        S.i_ck = S.st_i; S.j_ck = S.st_j; S.off_ck = S.off;
                        checkpoint(PC);
                        S.ck_trace_idx = S.trace_idx;
                        S.ck_block_idx = S.block_idx;
                        grow_no_max = false;
                    }
                    S.best_max = S.off_max;
                    S.y_drop_iter = 0;
                }
                bool cont = false;      // `continue` of the crate's loop: the next step is decided
                if (S.off_max < S.best_max - S.x_drop) {
                    if (S.x_drop_iter < B2_X_DROP_ITER - 1) S.x_drop_iter++;
                    else done = true;
                } else {
                    S.x_drop_iter = 0;
                }
                if (!done) {
                    const int qlen = S.Q.len, rlen = S.T.len;
                    if (S.st_i + bs > qlen && S.st_j + bs > rlen) {
                        done = true;
                    } else if (S.st_j + bs > rlen) {
                        S.st_i += B2_STEP; S.dir = B2_DOWN; cont = true;
                    } else if (S.st_i + bs > qlen) {
                        S.st_j += B2_STEP; S.dir = B2_RIGHT; cont = true;
                    }
                }
                if (!done && !cont) {
                    const int next_size = bs * 2;
                    const bool want_grow = S.y_drop_iter > (bs / B2_STEP) - 1 || grow_no_max;
                    if (next_size <= MAXB) {
                        if (want_grow) {
                            S.prev_size = bs;
                            S.block_size = next_size;
                            S.dir = B2_GROW;
                            S.st_i = S.i_ck; S.st_j = S.j_ck; S.off = S.off_ck;
                            restore(PC);
                            S.trace_idx = S.ck_trace_idx;
                            S.block_idx = S.ck_block_idx;
                            S.y_drop_iter = 0;
                            cont = true;
                        }
                    } else if (next_size <= BLOCK_REF_MAX_SIZE && want_grow) {
                        S.overflow = 1;      // the crate would grow beyond what this kernel holds: not decided here
                        done = true;
                    }
                }
                if (!done && !cont) {
                    if (bs > S.min_size && S.y_drop_iter == 0) {      // SHRINK (:542-586)
                        const int s1 = max(entry(PC, ROW1, bs - 1), entry(PC, ROW1, bs - 2));
                        const int s2 = max(entry(PC, COL1, bs - 1), entry(PC, COL1, bs - 2));
                        if (max(s1, s2) >= mx) {
                            S.prev_dir = B2_GROW;
                            const int nb = bs / 2;
                            S.block_size = nb;
                            for_arrays([&](VN &a, VN &k) {      // copy_vec(i, i + block_size)
                                (void)k;
                                if (nb < 64) {
                                    const unsigned up = b2_lane_from(a[0], lane + 32);      // (every lane takes part: outside the selection)
                                    a[0] = b2_put<P>(a[0], lane < nb ? up : a[0]);
                                } else {
                                    const int sh = nb >> 6;
#pragma unroll
                                    for (int c = 0; c < NCH / 2; c++) {
                                        unsigned src = a[c];
#pragma unroll
                                        for (int c2 = c + 1; c2 < NCH; c2++)
                                            if (c2 == c + sh) src = a[c2];
                                        if (c < sh) a[c] = b2_put<P>(a[c], src);
                                    }
                                }
                            });
                            S.st_i += nb;
                            S.st_j += nb;
                            S.i_ck = S.st_i; S.j_ck = S.st_j; S.off_ck = S.off;
                            checkpoint(PC);
                            right_max = pmax8(PC, COL1[0]);
                            down_max = pmax8(PC, ROW1[0]);
                            S.ck_trace_idx = S.trace_idx;
                            S.ck_block_idx = S.block_idx;
                            S.y_drop_iter = 0;
                        }
                    }
                    if (down_max > right_max) { S.st_i += B2_STEP; S.dir = B2_DOWN; }
                    else { S.st_j += B2_STEP; S.dir = B2_RIGHT; }
                }
            }
            if (done) {
                // ---- align_core returned: the loop over minimum sizes (StripedSmithWaterman.cpp:1021-1038) ----
                bool too_large = S.overflow != 0;
                if (!too_large) { S.score = S.best_max; S.ri = S.best_i; S.rj = S.best_j; }
                if (S.dbg_first < 0) S.dbg_first = (S.best_max & 0xFFFF) | (S.dbg_steps << 16);
                S.min_size *= 2;
                if (!too_large && S.score < S.target && S.min_size <= MAXB) {
                    start_attempt(PC, S);
                } else {
                    // (the crate would go on to larger minimum sizes when the score is not reached - not decided by this kernel)
                    if (MAXB < BLOCK_REF_MAX_SIZE && !too_large && S.score < S.target) too_large = true;
                    finish_pair(S, too_large);
                    fetch(S);      // (past the end of the queue: a dummy that computes on, masked off)
                    start_attempt(PC, S);
                }
            }
        }
        begin_step(PC, S);
    };

    auto post = [&](auto PC, B2Pair &S) {
        if (!S.active) return;
        B2Job &J = S.job;
        if (TRACE) store_trace(PC, S);
        if (J.kind == B2_JOB_GROW1 || J.kind == B2_JOB_GROW2) {
            const bool g1 = J.kind == B2_JOB_GROW1;      // the first grow job's outputs are D_col / C_col entries, the second's D_row / R_row
            const int base = J.out_base + J.oct * 8;
            VN a1 = g1 ? COL1 : ROW1, a2 = g1 ? COL2 : ROW2;      // (whole values: see VN)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                a1[c] = place_one(PC, a1[c], c, OUT1, base);
                a2[c] = place_one(PC, a2[c], c, OUT2, base);
            }
            COL1 = g1 ? a1 : COL1; COL2 = g1 ? a2 : COL2;
            ROW1 = g1 ? ROW1 : a1; ROW2 = g1 ? ROW2 : a2;
        }
        J.oct++;
        if (J.oct * 8 >= J.width) advance(PC, S, false);
        else if ((J.oct & 7) == 0) reload_cols(PC, S);
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    fetch(S0);
    advance(P0{}, S0, true);
    fetch(S1);
    advance(P1{}, S1, true);
    while (S0.active || S1.active) {
        octet();
        post(P0{}, S0);
        post(P1{}, S1);
    }
}

// Trace::cigar_core (scan_block.rs:1844-2006) for the pairs sw_block2_kernel<true> answered MMGPU_BLOCK_OK: one LANE per pair walks
// from the end cell to the origin; identities, the string (forward order: the reference reverses twice, :1071-1110) and its length.
__global__ __launch_bounds__(64) void sw_block2_walk_kernel(Block2Launch L) {
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
    uint32_t btstart = 0, nch = 1, stride = 32;
    while (i > 0 || j > 0) {
        if (!(i >= bi && j >= bj)) {
            block_idx--;
            const BkBlock b = blocks[block_idx];
            bi = (int)b.i; bj = (int)b.j; bright = (int)b.right; btstart = b.tstart;
            const int rows = bright ? (int)b.h : (int)b.w;      // the job's rows (place_block's height)
            nch = rows > 64 ? (uint32_t)(rows + 63) >> 6 : 1u;
            stride = rows < 64 ? 32u : 64u;
            continue;
        }
        const int ci = i - bi, cj = j - bj;
        const int r = bright ? ci : cj, c = bright ? cj : ci;
        const uint32_t w = trace[btstart + ((uint32_t)(c >> 3) * nch + (uint32_t)(r >> 6)) * stride + (uint32_t)(r & 63)];
        const int k = c & 7;
        const unsigned nib = ((w >> (k < 4 ? 12 - 4 * k : 44 - 4 * k)) & 0xFu) ^ 0xFu;
        const unsigned tt = nib & 3u, t2 = nib >> 2;
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
        for (uint32_t k = 0; k < (n & 3u); k++) bt[(n & ~3u) + k] = (char)(word >> (8 * k));
    }
    out.ident = ids;
    out.bt_len = n;
    out.reserved = 0;
    L.out[J.slot] = out;
}

}  // namespace

hipError_t launch_sw_block2(const Block2Launch &L, bool trace, bool large, uint32_t n_waves, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    if (large) {
        if (trace) hipLaunchKernelGGL((sw_block2_kernel<true, BLOCK2_LARGE_SIZE / 64>), dim3(n_waves), dim3(64), 0, stream, L);
        else hipLaunchKernelGGL((sw_block2_kernel<false, BLOCK2_LARGE_SIZE / 64>), dim3(n_waves), dim3(64), 0, stream, L);
    } else {
        if (trace) hipLaunchKernelGGL((sw_block2_kernel<true, BLOCK2_MAX_SIZE / 64>), dim3(n_waves), dim3(64), 0, stream, L);
        else hipLaunchKernelGGL((sw_block2_kernel<false, BLOCK2_MAX_SIZE / 64>), dim3(n_waves), dim3(64), 0, stream, L);
    }
    return hipGetLastError();
}

hipError_t launch_sw_block2_walk(const Block2Launch &L, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_block2_walk_kernel, dim3((L.n_jobs + 63) / 64), dim3(64), 0, stream, L);
    return hipGetLastError();
}

void warm_block2() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&(sw_block2_kernel<false, BLOCK2_MAX_SIZE / 64>)));
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&(sw_block2_kernel<true, BLOCK2_MAX_SIZE / 64>)));
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&sw_block2_walk_kernel));
}

}  // namespace mmgpu
