// Shared between the GPU kernel (nucl_kernel.hip) and the host lane emulator of the tests (tests/nucl_emu.cpp,
// which runs this very code with 16 cooperative contexts standing in for the 16 lanes of a group and checks it
// against the reference's vectors without a GPU).  The includer supplies:
//   NUCL_HD                      function qualifiers
//   NUCL_LANE()                  lane of the group, 0..15
//   NUCL_SHFL(v, src) / NUCL_SHFL_XOR(v, mask) / NUCL_SHFL_U64(v, src)   exchange inside the group
//   NUCL_SYNC()                  phase boundary: LDS written by one lane is read by another.  On the GPU the lanes of a
//                                group run in lock step, so this only pins the order of the LDS operations; the
//                                emulator switches contexts here.
//   NUCL_SYNC_MEM()              the same for the scratch in global memory (direction bytes, backtrack letters)
//   NUCL_ATOMIC_ADD_U32 / _U64   work queue and output cursor
#ifndef MMGPU_NUCL_CORE_H
#define MMGPU_NUCL_CORE_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/mmgpu.h"

namespace mmgpu {

// everything the kernel needs, device pointers only
struct NuclLaunch {
    const mmgpu_nucl_pair *pairs;
    const uint32_t *order;         // processing order (longest first): pair index
    uint32_t n_pairs;
    const uint8_t *q_res;          // queries, numeric, concatenated
    const uint32_t *q_off;         // [nq + 1]
    const uint8_t *t_res;
    const uint32_t *t_off4;
    const uint32_t *t_len;
    int8_t mat[25];
    uint8_t rev_lookup[8];
    int gapo, gape, zdrop;
    int past_end_q, past_end_t;
    int wrapped;                   // --wrapped-scoring: every query is its sequence written twice
    uint8_t *pscratch;             // direction bytes, one slice per 16-lane group of the grid
    uint64_t pscratch_stride;
    char *wscratch;                // backtrack letters, one slice per group
    uint64_t wscratch_stride;
    mmgpu_nucl_hit *out;
    char *bt;
    unsigned long long *bt_cursor;
    unsigned long long bt_cap;
    uint32_t *next_pair;           // work queue head
};

#ifdef NUCL_HD

#ifndef NUCL_NG
#define NUCL_NG 16   // lanes per alignment: 16 (one block of the reference's vectors per step) or 64 (four)
#endif
#ifndef NUCL_NS
#define NUCL_NS nucl
#endif

namespace NUCL_NS {

constexpr int NG = NUCL_NG;         // lanes per alignment
constexpr int BPS = NG / 16;        // blocks of 16 target positions the group handles per step
static_assert(NG == 16 || NG == 32 || NG == 64, "a group is one, two or four blocks wide");
constexpr int WIN = 256;           // window of target positions kept in LDS (power of two, >= 16 blocks)
constexpr int KSW_NEG_INF = -0x40000000;
constexpr int KSW_W = 64;          // band of the reference's calls (BandedNucleotideAligner.cpp:176,193,203)

struct GroupLds {                  // per alignment
    uint8_t u[WIN], v[WIN], x[WIN], y[WIN], s[WIN];
    int32_t H[WIN];
};

// A sequence as one ksw call sees it.  idx = reversed ? L - (off + k) : off + k walks the aligned strand, whose
// element L (one past the end) is `past` - the reference's reversed copies are shifted by one because seq_reverse is
// called with L instead of L - 1 (BandedNucleotideAligner.cpp:61,68,93).
struct SeqView {
    const uint8_t *base;
    const uint8_t *rl;     // reverse-complement table, or null: the strand is base[] itself
    int L, off, past;
    bool reversed;
    NUCL_HD uint8_t strand(int idx) const {
        if (idx >= L) return (uint8_t)past;
        return rl ? rl[base[L - 1 - idx]] : base[idx];
    }
    NUCL_HD uint8_t get(int k) const { return strand(reversed ? L - (off + k) : off + k); }
};

struct Ez {
    int max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score;
};

NUCL_HD int8_t s8(int v) { return (int8_t)(uint8_t)v; }

NUCL_HD int group_lane() { return NUCL_LANE(); }

// (H, order) maximum over the 16 lanes of the group: larger H wins, equal H the smaller order
NUCL_HD void group_best(int &h, unsigned &o) {
#pragma unroll
    for (int d = 1; d < NG; d <<= 1) {
        const int h2 = NUCL_SHFL_XOR(h, d);
        const unsigned o2 = NUCL_SHFL_XOR(o, d);
        if (h2 > h || (h2 == h && o2 < o)) { h = h2; o = o2; }
    }
}

// ksw_extz2_sse without KSW_EZ_APPROX_MAX / _RIGHT / _GENERIC_SC (the flags the reference's caller never sets);
// WITH_P: record the direction bytes (calls without KSW_EZ_SCORE_ONLY).  All lanes of the group run this in step.
template <bool WITH_P>
NUCL_HD void ksw_extz2(const SeqView &qv, int qlen, const SeqView &tv, int tlen, const int8_t *mat, int q, int e, int zdrop,
                          GroupLds &S, uint8_t *p, Ez &ez) {
    const int lane = group_lane();
    ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
    ez.max = 0;
    ez.score = ez.mqe = ez.mte = KSW_NEG_INF;
    ez.zdropped = 0;
    if (qlen <= 0 || tlen <= 0) return;
    const int m = 5, w = KSW_W, qe = q + e;
    const int8_t sc_mch = mat[0], sc_mis = mat[1];
    const uint8_t max_sc_u = (uint8_t)s8(mat[0] + qe * 2);
    const int tlen_ = (tlen + 15) / 16;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    int min_sc = mat[1];
    for (int t = 1; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
    if (-min_sc > 2 * qe) return;

    int init_blocks = 0;           // blocks [0, init_blocks) of the window hold defined contents
    int last_st = -1, last_en = -1;
    for (int r = 0; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) {
            ez.zdropped = 1;
            break;
        }
        const int st0 = st, en0 = en;
        st = st / 16 * 16;
        en = (en + 16) / 16 * 16 - 1;
        // blocks the window reaches for the first time: zeros / "never computed" (the reference's calloc and H fill)
        {
            int need = (st0 + ((en0 - st0) / 16 + 1) * 16 + 15) / 16;     // one past the last block the score pass writes
            need = need < en / 16 + 1 ? en / 16 + 1 : need;
            need = need > tlen_ ? tlen_ : need;
#if NUCL_NG == 16   // one block per step: the form the GPU parity tests were run on, kept verbatim
            for (int b = init_blocks; b < need; ++b) {
                const int k = (b * 16 + lane) & (WIN - 1);
                S.u[k] = 0; S.v[k] = 0; S.x[k] = 0; S.y[k] = 0; S.s[k] = 0;
                S.H[k] = KSW_NEG_INF;
            }
#else
            for (int b0 = init_blocks; b0 < need; b0 += BPS) {
                const int b = b0 + lane / 16;
                if (b < need) {
                    const int k = (b * 16 + (lane & 15)) & (WIN - 1);
                    S.u[k] = 0; S.v[k] = 0; S.x[k] = 0; S.y[k] = 0; S.s[k] = 0;
                    S.H[k] = KSW_NEG_INF;
                }
            }
#endif
            init_blocks = init_blocks > need ? init_blocks : need;
        }
        NUCL_SYNC();
        // what enters the lowest block from the left (:126-132)
        int8_t x1, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { x1 = (int8_t)S.x[(st - 1) & (WIN - 1)]; v1 = (int8_t)S.v[(st - 1) & (WIN - 1)]; }
            else x1 = v1 = 0;
        } else {
            x1 = 0;
            v1 = r ? (int8_t)q : (int8_t)0;
        }
        NUCL_SYNC();   // every lane has read the carry-in
        if (en >= r && lane == 0) {
            S.y[r & (WIN - 1)] = 0;
            S.u[r & (WIN - 1)] = r ? (uint8_t)q : (uint8_t)0;
        }
        // scores of this anti-diagonal, whole 16-byte groups starting at st0 (:135-145); letter m - 1 is a wildcard.
        // Positions past the target read the allocation's zeros, query positions before its start likewise.
#if NUCL_NG == 16
        for (int t0 = st0; t0 <= en0; t0 += 16) {
            const int t = t0 + lane;
            if (t < tlen_ * 16) {
#else
        const int last_scored = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;   // the reference scores whole groups of 16 from st0
        for (int t0 = st0; t0 <= last_scored; t0 += NG) {
            const int t = t0 + lane;
            if (t <= last_scored && t < tlen_ * 16) {
#endif
                const uint8_t a = t < tlen ? tv.get(t) : (uint8_t)0;
                const uint8_t b = (r - t >= 0 && r - t < qlen) ? qv.get(r - t) : (uint8_t)0;
                int8_t sc = a == b ? sc_mch : sc_mis;
                if (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) sc = 0;
                S.s[t & (WIN - 1)] = (uint8_t)sc;
            }
        }
        NUCL_SYNC();
        // core: blocks from the highest to the lowest - cell t reads x[t-1], v[t-1], which a lower block still holds
        const bool x1_neg = x1 < 0, v1_neg = v1 < 0;
#if NUCL_NG == 16
        for (int blk = en / 16; blk >= st / 16; --blk) {
            const int t = blk * 16 + lane, k = t & (WIN - 1);
            int8_t xt1, vt1;
            if (t == st) { xt1 = x1; vt1 = v1; }
            else { xt1 = (int8_t)S.x[(t - 1) & (WIN - 1)]; vt1 = (int8_t)S.v[(t - 1) & (WIN - 1)]; }
#else
        for (int hb = en / 16; hb >= st / 16; hb -= BPS) {
            const int blk = hb - lane / 16;
            const bool on = blk >= st / 16;             // a step may reach below the lowest block
            const int t = blk * 16 + (lane & 15), k = t & (WIN - 1);
            int8_t xt1 = 0, vt1 = 0;
            if (t == st) { xt1 = x1; vt1 = v1; }
            else if (on) { xt1 = (int8_t)S.x[(t - 1) & (WIN - 1)]; vt1 = (int8_t)S.v[(t - 1) & (WIN - 1)]; }
#endif
            // _mm_cvtsi32_si128 of a negative carry-in also sets bytes 1..3 of the first block's shifted vectors (:151-152)
            if (t - st >= 1 && t - st <= 3) {
                if (x1_neg) xt1 = (int8_t)0xFF;
                if (v1_neg) vt1 = (int8_t)0xFF;
            }
#if NUCL_NG == 16
            int8_t z = s8((int8_t)S.s[k] + s8(qe * 2));
            int8_t a = s8(xt1 + vt1);
            const int8_t ut = (int8_t)S.u[k];
            int8_t b = s8((int8_t)S.y[k] + ut);
            NUCL_SYNC();   // all reads of the block before its writes (lane + 1 reads this lane's x, v)
#else
            int8_t z = 0, ut = 0, b = 0;
            if (on) {
                z = s8((int8_t)S.s[k] + s8(qe * 2));
                ut = (int8_t)S.u[k];
                b = s8((int8_t)S.y[k] + ut);
            }
            int8_t a = s8(xt1 + vt1);
            NUCL_SYNC();   // all reads of the step before its writes (lane + 1 reads this lane's x, v)
            if (!on) continue;
#endif
            uint8_t d = 0;
            if (WITH_P) d = a > z ? 1 : 0;
            z = z > a ? z : a;
            if (WITH_P && b > z) d = 2;
            uint8_t zu = (uint8_t)z > (uint8_t)b ? (uint8_t)z : (uint8_t)b;
            zu = zu < max_sc_u ? zu : max_sc_u;
            z = (int8_t)zu;
            S.u[k] = (uint8_t)s8(z - vt1);
            S.v[k] = (uint8_t)s8(z - ut);
            z = s8(z - q);
            a = s8(a - z);
            b = s8(b - z);
            S.x[k] = (uint8_t)(a > 0 ? a : 0);
            S.y[k] = (uint8_t)(b > 0 ? b : 0);
            if (WITH_P) {
                if (a > 0) d |= 0x08;
                if (b > 0) d |= 0x10;
                p[(size_t)r * (size_t)(n_col_ * 16) + (size_t)(t - st)] = d;
            }
        }
        // exact maximum of the band (:207-250); ties follow the reference's scan: position en0 first, then the four
        // interleaved lanes of its 4-wide loop (lane by lane, ascending), then the scalar remainder
        int max_H, max_t;
        NUCL_SYNC();
        if (r > 0) {
            const int k0 = en0 & (WIN - 1);
            const int h_en0 = en0 > 0 ? S.H[(en0 - 1) & (WIN - 1)] + (int)S.u[k0] - qe : S.H[k0] + (int)S.v[k0] - qe;
            const int en1 = st0 + (en0 - st0) / 4 * 4;
            NUCL_SYNC();   // H[en0 - 1] is read above and updated below by another lane
            int best_h = h_en0;
            unsigned best_o = 0;
            for (int t = st0 + lane; t < en0; t += NG) {
                const int k = t & (WIN - 1);
                const int h = S.H[k] + (int)S.v[k] - qe;
                S.H[k] = h;
                const unsigned o = t < en1 ? 1u + (unsigned)((t - st0) & 3) * 0x100000u + (unsigned)((t - st0) >> 2)
                                           : 1u + 4u * 0x100000u + (unsigned)(t - en1);
                if (h > best_h || (h == best_h && o < best_o)) { best_h = h; best_o = o; }
            }
            if (lane == 0) S.H[k0] = h_en0;
            group_best(best_h, best_o);
            max_H = best_h;
            if (best_o == 0) max_t = en0;
            else if (best_o <= 4u * 0x100000u) max_t = st0 + (int)((best_o - 1u) & 0xFFFFFu) * 4 + (int)((best_o - 1u) >> 20);
            else max_t = en1 + (int)(best_o - 1u - 4u * 0x100000u);
        } else {
            const int h0 = (int)S.v[0] - qe - qe;
            if (lane == 0) S.H[0] = h0;
            max_H = h0;
            max_t = 0;
        }
        NUCL_SYNC();
        {
            const int h_en0 = S.H[en0 & (WIN - 1)], h_st0 = S.H[st0 & (WIN - 1)];
            if (en0 == tlen - 1 && h_en0 > ez.mte) { ez.mte = h_en0; ez.mte_q = r - en; }
            if (r - st0 == qlen - 1 && h_st0 > ez.mqe) { ez.mqe = h_st0; ez.mqe_t = st0; }
        }
        // ksw_apply_zdrop (ksw2.h:182-199)
        if (max_H > ez.max) {
            ez.max = max_H;
            ez.max_t = max_t;
            ez.max_q = r - max_t;
        } else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
            const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q;
            const int l = tl > ql ? tl - ql : ql - tl;
            if (zdrop >= 0 && ez.max - max_H > zdrop + l * e) {
                ez.zdropped = 1;
                break;
            }
        }
        if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = S.H[(tlen - 1) & (WIN - 1)];
        last_st = st;
        last_en = en;
        NUCL_SYNC();
    }
    NUCL_SYNC();
}

// first / last block-aligned target position of anti-diagonal r (what the reference stores in off[] / off_end[])
NUCL_HD void band_of(int r, int qlen, int tlen, int &st, int &en) {
    st = 0;
    en = tlen - 1;
    if (st < r - qlen + 1) st = r - qlen + 1;
    if (en > r) en = r;
    if (st < (r - KSW_W + 1) >> 1) st = (r - KSW_W + 1) >> 1;
    if (en > (r + KSW_W) >> 1) en = (r + KSW_W) >> 1;
    st = st / 16 * 16;
    en = (en + 16) / 16 * 16 - 1;
}

// ksw_backtrack (is_rot, no introns) from cell (i0 = target, j0 = query); writes one letter per step, last column
// first, into w[]; returns the number of letters.  One lane.
NUCL_HD int ksw_walk(const uint8_t *p, int qlen, int tlen, int i0, int j0, char *w) {
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < KSW_W + 1 ? n_col_ : KSW_W + 1) + 15) / 16 + 1;
    const size_t n_col = (size_t)n_col_ * 16;
    int i = i0, j = j0, state = 0, n = 0;
    while (i >= 0 && j >= 0) {
        const int r = i + j;
        int st, en;
        band_of(r, qlen, tlen, st, en);
        int force_state = -1;
        if (i < st) force_state = 2;
        if (i > en) force_state = 1;
        const unsigned tmp = force_state < 0 ? p[(size_t)r * n_col + (size_t)(i - st)] : 0u;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) { w[n++] = 'M'; --i; --j; }
        else if (state == 1 || state == 3) { w[n++] = 'D'; --i; }
        else { w[n++] = 'I'; --j; }
    }
    for (; i >= 0; --i) w[n++] = 'D';
    for (; j >= 0; --j) w[n++] = 'I';
    return n;
}

// DistanceCalculator::computeSubstitutionStartEndDistance along one diagonal (:178-200), the group working through
// 16 positions at a time: every lane fetches one score, the recurrence then runs over the 16 by shuffles (uniform).
NUCL_HD void seed_segment(const SeqView &qv, int qo, const SeqView &tv, int to, unsigned len, const int8_t *mat, int &start,
                             int &end, int &best) {
    const int lane = group_lane();
    int max_score = 0, max_end = 0, max_start = 0, min_pos = -1, score = 0;
    for (unsigned base = 0; base < len; base += NG) {
        const unsigned pos = base + (unsigned)lane;
        int sc = 0;
        if (pos < len) sc = mat[qv.strand(qo + (int)pos) * 5 + tv.strand(to + (int)pos)];
        const int cnt = (int)(len - base < (unsigned)NG ? len - base : (unsigned)NG);
        for (int k = 0; k < cnt; ++k) {
            score += NUCL_SHFL(sc, k);
            if (score <= 0) { score = 0; min_pos = (int)base + k; }
            if (score > max_score) { max_end = (int)base + k; max_start = min_pos + 1; max_score = score; }
        }
    }
    start = max_start;
    end = max_end;
    best = max_score;
}

struct Seed {
    int start, end, diagonal;
    unsigned score, dist;
};

NUCL_HD Seed seed_on_diagonal(const SeqView &qv, unsigned qlen, const SeqView &tv, unsigned tlen, int diagonal, const int8_t *mat) {
    Seed r;
    r.start = -1; r.end = -1; r.score = 0;
    r.dist = (unsigned)(diagonal < 0 ? -diagonal : diagonal);
    r.diagonal = diagonal;
    int s, e, sc;
    if (diagonal >= 0 && r.dist < qlen) {
        const unsigned len = tlen < qlen - r.dist ? tlen : qlen - r.dist;
        seed_segment(qv, (int)r.dist, tv, 0, len, mat, s, e, sc);
        r.start = s; r.end = e; r.score = (unsigned)sc;
    } else if (diagonal < 0 && r.dist < tlen) {
        const unsigned len = tlen - r.dist < qlen ? tlen - r.dist : qlen;
        seed_segment(qv, 0, tv, (int)r.dist, len, mat, s, e, sc);
        r.start = s; r.end = e; r.score = (unsigned)sc;
    }
    return r;
}

// The ungapped seed of BandedNucleotideAligner::align (:98-113).  Plain: every 65536-shift of the 16-bit prefilter diagonal that
// fits (computeUngappedAlignment, DistanceCalculator.h:93-112).  Wrapped scoring: the query is its sequence written twice - where
// it is at least twice the target, every window of half its length that starts at a shift of the diagonal is laid on the target
// from position 0 (computeUngappedWrappedAlignment, :56-90; the loop bounds are the reference's unsigned comparisons), otherwise
// the plain seed over the first half.
NUCL_HD Seed pick_seed(const SeqView &qv, int qlen, const SeqView &tv, int tlen, unsigned diag16, bool wrapped, const int8_t *mat) {
    Seed best;
    best.start = -1; best.end = -1; best.score = 0; best.dist = 0; best.diagonal = 0;
    if (wrapped && qlen >= tlen * 2) {
        const unsigned half = (unsigned)qlen / 2u;
        const unsigned len = (unsigned)tlen < half ? (unsigned)tlen : half;
        for (int pass = 0; pass < 2; pass++)
            for (unsigned d = pass == 0 ? 1u : 0u;
                 pass == 0 ? (0u - d * 65536u + diag16) > 0u - (unsigned)tlen : (d * 65536u + diag16) < half; d++) {
                const int real = pass == 0 ? (int)((0u - d * 65536u + diag16) + half) : (int)(d * 65536u + diag16);
                Seed t;
                t.dist = (unsigned)(real < 0 ? -real : real);
                t.diagonal = real;
                int s, e, sc;
                seed_segment(qv, real, tv, 0, len, mat, s, e, sc);
                t.start = s; t.end = e; t.score = (unsigned)sc;
                if (t.score > best.score) best = t;
            }
        return best;
    }
    const unsigned ql = wrapped ? (unsigned)qlen / 2u : (unsigned)qlen;
    for (unsigned d = 1; d <= 1u + (unsigned)tlen / 32768u; d++) {
        const Seed t = seed_on_diagonal(qv, ql, tv, (unsigned)tlen, (int)(0u - d * 65536u + diag16), mat);
        if (t.score > best.score) best = t;
    }
    for (unsigned d = 0; d <= ql / 65536u; d++) {
        const Seed t = seed_on_diagonal(qv, ql, tv, (unsigned)tlen, (int)(d * 65536u + diag16), mat);
        if (t.score > best.score) best = t;
    }
    return best;
}

// One 16-lane group: pulls pairs from the queue until it is empty.
NUCL_HD void align_group(const NuclLaunch &L, GroupLds &S, uint8_t *p, char *w) {
    const int lane = group_lane();
    for (;;) {
        // Every lane takes part in the ticket draw (only lane 0 adds something): with the draw under `if (lane == 0)`
        // the compiler threads the loop-invariant branch through the back edge, lanes 1..15 then come round to the
        // shuffle below without lane 0 and read a stale ticket - the group falls apart and never ends.  The phase
        // boundaries at both ends of the body keep the 16 lanes together across the back edge.
        NUCL_SYNC();
        unsigned pi = NUCL_ATOMIC_ADD_U32(L.next_pair, lane == 0 ? 1u : 0u);
        pi = (unsigned)NUCL_SHFL((int)pi, 0);
        if (pi >= L.n_pairs) break;
        const mmgpu_nucl_pair P = L.pairs[L.order[pi]];
        const int qlen = (int)(L.q_off[P.query + 1] - L.q_off[P.query]);
        const int tlen = (int)L.t_len[P.target];
        SeqView qv, tv;
        qv.base = L.q_res + L.q_off[P.query];
        qv.rl = P.reverse ? L.rev_lookup : nullptr;
        qv.L = qlen; qv.off = 0; qv.past = (P.past_end & 0x80u) ? (int)(P.past_end & 7u) : L.past_end_q; qv.reversed = false;
        tv.base = L.t_res + (size_t)L.t_off4[P.target] * 4;
        tv.rl = nullptr;
        tv.L = tlen; tv.off = 0; tv.past = (P.past_end & 0x80u) ? (int)((P.past_end >> 3) & 7u) : L.past_end_t; tv.reversed = false;

        // ---- ungapped seed (pick_seed); origQueryLen of the reference: half of a wrapped query
        const bool wrapped = L.wrapped != 0;
        const int orig = wrapped ? qlen / 2 : qlen;
        const Seed best = pick_seed(qv, qlen, tv, tlen, (unsigned)P.diagonal, wrapped, L.mat);
        int qs, qe_, ts, te;
        if (best.diagonal >= 0) { qs = best.start + (int)best.dist; qe_ = best.end + (int)best.dist; ts = best.start; te = best.end; }
        else { qs = best.start; qe_ = best.end; ts = best.start + (int)best.dist; te = best.end + (int)best.dist; }

        mmgpu_nucl_hit res;
        res.bt_off = 0;
        res.status = MMGPU_NUCL_OK;
        int n_bt = 0;
        bool walk_reversed = false;     // w[] holds the letters last column first
        if (qe_ - qs == orig - 1 && ts == 0 && te == tlen - 1) {
            // the seed spans both sequences (:130-160)
            res.score = (int32_t)best.score;
            res.q_start = qs; res.q_end = qe_; res.t_start = ts; res.t_end = te;
            for (int i = lane; i < orig; i += NG) w[i] = 'M';
            n_bt = orig;
        } else {
            // left extension, score only, on the (shifted) reversed sequences from the seed's end backwards (:165-181)
            const int q_start_rev = qlen - qe_ - 1, t_start_rev = tlen - te - 1;
            SeqView qr = qv, tr = tv;
            qr.reversed = true; qr.off = q_start_rev;
            tr.reversed = true; tr.off = t_start_rev;
            Ez ez, eza;
            // (wrapped scoring: neither extension runs over more than the original query, :171-174,189-191)
            const int q_rev_len = wrapped && qlen - q_start_rev > orig ? orig : qlen - q_start_rev;
            ksw_extz2<false>(qr, q_rev_len, tr, tlen - t_start_rev, L.mat, L.gapo, L.gape, L.zdrop, S, nullptr, ez);
            const int q_start = qlen - (q_start_rev + ez.max_q) - 1, t_start = tlen - (t_start_rev + ez.max_t) - 1;
            // right extension with directions from that start (:183-196)
            SeqView qf = qv, tf = tv;
            qf.off = q_start;
            tf.off = t_start;
            int wq = wrapped && qlen - q_start > orig ? orig : qlen - q_start, wt = tlen - t_start;
            ksw_extz2<true>(qf, wq, tf, wt, L.mat, L.gapo, L.gape, L.zdrop, S, p, eza);
            if (ez.max_q > eza.max_q && ez.max_t > eza.max_t) {
                // the forward pass fell short of the backward pass: the backward pass is redone with directions and
                // its CIGAR reversed (:201-210)
                wq = q_rev_len;
                wt = tlen - t_start_rev;
                ksw_extz2<true>(qr, wq, tr, wt, L.mat, L.gapo, L.gape, L.zdrop, S, p, eza);
                walk_reversed = true;
            }
            NUCL_SYNC_MEM();   // the direction bytes were written by all lanes
            if (lane == 0 && eza.max_t >= 0 && eza.max_q >= 0) n_bt = ksw_walk(p, wq, wt, eza.max_t, eza.max_q, w);
            n_bt = NUCL_SHFL(n_bt, 0);
            NUCL_SYNC_MEM();
            res.score = eza.max;
            res.q_start = q_start;
            res.q_end = q_start + eza.max_q;
            res.t_start = t_start;
            res.t_end = t_start + eza.max_t;
            // the walk produced the letters last column first; the forward pass wants them reversed, the redone
            // backward pass (CIGAR reversed once more by the caller) as they are
            walk_reversed = !walk_reversed;
        }
        NUCL_SYNC_MEM();   // w[] was written by all lanes of the group, lane 0 reads it below
        // ---- output: reserve space, copy the string in alignment order, count identities (:231-258)
        unsigned long long off = NUCL_ATOMIC_ADD_U64(L.bt_cursor, lane == 0 ? (unsigned long long)n_bt + 1ull : 0ull);
        off = NUCL_SHFL_U64(off, 0);
        const bool fits = off + (unsigned long long)n_bt + 1ull <= L.bt_cap;
        if (fits) {
            for (int i = lane; i < n_bt; i += NG) L.bt[off + (unsigned long long)i] = walk_reversed ? w[n_bt - 1 - i] : w[i];
            if (lane == 0) L.bt[off + (unsigned long long)n_bt] = 0;
        }
        unsigned ids = 0;
        if (lane == 0) {
            int tp = res.t_start, qp = res.q_start;
            for (int i = 0; i < n_bt; ++i) {
                const char c = walk_reversed ? w[n_bt - 1 - i] : w[i];
                if (c == 'M') { ids += tv.strand(tp) == qv.strand(qp) ? 1u : 0u; ++qp; ++tp; }
                else if (c == 'I') ++qp;
                else ++tp;
            }
            res.ident = ids;
            res.bt_off = fits ? off : 0ull;
            res.bt_len = (uint32_t)n_bt;
            if (!fits) res.status = MMGPU_NUCL_BT_OVERFLOW;
            L.out[L.order[pi]] = res;
        }
        NUCL_SYNC();
    }
}

}  // namespace NUCL_NS

#endif  // NUCL_HD

}  // namespace mmgpu

#endif
