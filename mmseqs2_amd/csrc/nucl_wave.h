// Nucleotide alignment step, ONE WAVEFRONT PER ALIGNMENT with the DP state in registers (the default GPU kernel;
// nucl_core.h's 16-lane LDS formulation stays as the cross-check, MMGPU_NUCL_LANES=16).
//
// ksw_extz2_sse (lib/ksw2/ksw2_extz2_sse.cpp:44-285) works on anti-diagonals: every cell of anti-diagonal r depends on
// anti-diagonal r - 1 only (u, v, x, y are updated in place; cell t reads x[t-1], v[t-1] of the previous one), the band
// (w = 64) holds at most 65 cells, block-aligned at most 96.  So the cells of one anti-diagonal are the parallel
// dimension and 64 lanes x 2 cells cover it:
//   * target position t lives in lane t & 63, slot (t >> 6) & 1 - for as long as the band is anywhere near it (live
//     positions span < 112, the two slots hold 128): u, v, x, y (bytes), the score byte and the 32-bit H of a position
//     never leave the registers of their lane; a slot is re-initialised when the band first reaches the 16-block of the
//     position it now stands for (= the reference's calloc: the block semantics of the SSE code are kept, blocks are
//     computed whole, cells outside the band see what the arrays hold there);
//   * x[t-1], v[t-1]: the four bytes of both slots packed in one register, rotated by one lane (lane 0 takes lane 63's
//     other slot) - one cross-lane move per anti-diagonal, no LDS round trip, no phase barrier;
//   * sequence letters: 512-byte rings in LDS, refilled 256 letters at a time (the only global reads of the loop);
//   * the exact band maximum with the reference's tie order: per-lane candidates of both slots, then a butterfly;
//   * direction bytes go to the wave's scratch in HBM (96 B per anti-diagonal, coalesced); the backtrack reads them back
//     through a 64-row LDS stage (lane 0 walks in LDS instead of one dependent HBM round trip per step);
//   * identities are counted by all lanes (letter counts -> prefix -> per-chunk walk).
// The anti-diagonals of an alignment are a dependent chain (~150 wave instructions each); parallelism comes from one
// alignment per wavefront x all wavefronts the chip holds.
//
// Included after nucl_core.h with NUCL_NG = 64 (SeqView, Ez, seeds, band_of come from there).  The includer supplies,
// besides nucl_core.h's macros:  NUCL_ROR1_U32(v)  value of lane - 1 (lane 0: lane 63);  NUCL_READLANE(v, l)  value of
// the (wave-uniform) lane l;  NUCL_WAVE_MAX_I32(v) / NUCL_WAVE_MIN_U32(v)  reduction over the wave, result in every lane.
#ifndef MMGPU_NUCL_WAVE_H
#define MMGPU_NUCL_WAVE_H

namespace mmgpu {
namespace nuclw {

using namespace NUCL_NS;

constexpr int RING = 512;          // staged letters per sequence (two halves of 256)
constexpr int PROWS = 64;          // rows of direction bytes staged for the walk
constexpr int PROW_BYTES = 96;     // (min(65, n) + 15) / 16 + 1 <= 6 blocks of 16

struct WaveLds {
    uint8_t qring[RING], tring[RING];
    uint8_t prow[PROWS * PROW_BYTES];
};

// 256 letters [hi, hi + 256) of the sequence as this call sees it -> ring; letters past the end are 0 (the reference's
// calloc / memset padding, ksw2_extz2_sse.cpp:88-97)
NUCL_HD void stage_letters(const SeqView &v, int len, int hi, uint8_t *ring, int lane) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = hi + j * 64 + lane;
        ring[k & (RING - 1)] = k < len ? v.get(k) : (uint8_t)0;
    }
}

template <bool WITH_P>
NUCL_HD void ksw_extz2_wave(const SeqView &qv, int qlen, const SeqView &tv, int tlen, const int8_t *mat, int q, int e, int zdrop,
                               WaveLds &S, uint8_t *p, Ez &ez) {
    const int lane = NUCL_LANE();
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

    // register state of the lane's two positions: slot a = even 64-blocks of the target, slot b = odd ones
    uint8_t ua = 0, va = 0, xa = 0, ya = 0, sa = 0, ub = 0, vb = 0, xb = 0, yb = 0, sb = 0;
    int Ha = KSW_NEG_INF, Hb = KSW_NEG_INF;
    int init_blocks = 0;
    int last_st = -1, last_en = -1;
    int q_hi = 0, t_hi = 0;        // letters [.. - RING, ..) are staged
    NUCL_SYNC();
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
        // the positions this lane stands for on this anti-diagonal
        const int w0 = st & ~127;
        int ta = w0 + lane, tb = w0 + 64 + lane;
        if (ta < st) ta += 128;
        if (tb < st) tb += 128;
        // blocks the band reaches for the first time: zeros / "never computed" (the reference's calloc and H fill)
        const int last_scored = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;   // the reference scores whole groups of 16 from st0
        {
            int need = (last_scored + 1 + 15) / 16;     // one past the last block the score pass writes
            need = need < en / 16 + 1 ? en / 16 + 1 : need;
            need = need > tlen_ ? tlen_ : need;
            if (need > init_blocks) {       // wave-uniform: once per 16 positions the band advances
                if ((ta >> 4) >= init_blocks && (ta >> 4) < need) { ua = va = xa = ya = sa = 0; Ha = KSW_NEG_INF; }
                if ((tb >> 4) >= init_blocks && (tb >> 4) < need) { ub = vb = xb = yb = sb = 0; Hb = KSW_NEG_INF; }
                init_blocks = need;
            }
        }
        // x and v of both slots as one register: the carry-in below and the neighbour exchange further down read it
        const unsigned xv_packed = (unsigned)xa | ((unsigned)va << 8) | ((unsigned)xb << 16) | ((unsigned)vb << 24);
        // what enters the lowest block from the left (:126-132)
        int8_t x1, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) {
                const unsigned w1 = (unsigned)NUCL_READLANE((int)xv_packed, (st - 1) & 63) >> ((((st - 1) >> 6) & 1) * 16);
                x1 = (int8_t)(uint8_t)(w1 & 0xFFu);
                v1 = (int8_t)(uint8_t)((w1 >> 8) & 0xFFu);
            } else x1 = v1 = 0;
        } else {
            x1 = 0;
            v1 = r ? (int8_t)q : (int8_t)0;
        }
        if (en >= r) {      // :133-134, position r
            if (ta == r) { ya = 0; ua = r ? (uint8_t)q : (uint8_t)0; }
            if (tb == r) { yb = 0; ub = r ? (uint8_t)q : (uint8_t)0; }
        }
        // letters of this anti-diagonal: target [st0, last_scored], query [r - last_scored, r - st0]
        if (last_scored >= t_hi || r - st0 >= q_hi) {
            NUCL_SYNC();
            while (last_scored >= t_hi) { stage_letters(tv, tlen, t_hi, S.tring, lane); t_hi += 256; }
            while (r - st0 >= q_hi) { stage_letters(qv, qlen, q_hi, S.qring, lane); q_hi += 256; }
            NUCL_SYNC();
        }
        // scores (:135-145); letter m - 1 is a wildcard
        if (ta >= st0 && ta <= last_scored && ta < tlen_ * 16) {
            const uint8_t a = S.tring[ta & (RING - 1)];
            const uint8_t b = (r - ta >= 0 && r - ta < qlen) ? S.qring[(r - ta) & (RING - 1)] : (uint8_t)0;
            int8_t sc = a == b ? sc_mch : sc_mis;
            if (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) sc = 0;
            sa = (uint8_t)sc;
        }
        if (tb >= st0 && tb <= last_scored && tb < tlen_ * 16) {
            const uint8_t a = S.tring[tb & (RING - 1)];
            const uint8_t b = (r - tb >= 0 && r - tb < qlen) ? S.qring[(r - tb) & (RING - 1)] : (uint8_t)0;
            int8_t sc = a == b ? sc_mch : sc_mis;
            if (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) sc = 0;
            sb = (uint8_t)sc;
        }
        // x[t-1], v[t-1] of the previous anti-diagonal: one rotate of the packed bytes
        unsigned nb = NUCL_ROR1_U32(xv_packed);
        if (lane == 0) nb = (nb >> 16) | (nb << 16);     // lane 63's other slot
        const bool x1_neg = x1 < 0, v1_neg = v1 < 0;
        uint8_t *const prow_ptr = WITH_P ? p + (size_t)r * (size_t)(n_col_ * 16) : nullptr;   // wave-uniform
#define NUCLW_CELL(T, U, V, X, Y, SC, XT1, VT1)                                                         \
        if ((T) >= st && (T) <= en) {                                                                   \
            int8_t xt1 = (int8_t)(uint8_t)(XT1), vt1 = (int8_t)(uint8_t)(VT1);                          \
            if ((T) == st) { xt1 = x1; vt1 = v1; }                                                      \
            if ((T) - st >= 1 && (T) - st <= 3) {     /* _mm_cvtsi32_si128 of a negative carry-in (:151-152) */ \
                if (x1_neg) xt1 = (int8_t)0xFF;                                                         \
                if (v1_neg) vt1 = (int8_t)0xFF;                                                         \
            }                                                                                           \
            int8_t z = s8((int8_t)(SC) + s8(qe * 2));                                                   \
            int8_t a = s8(xt1 + vt1);                                                                   \
            const int8_t ut = (int8_t)(U);                                                              \
            int8_t b = s8((int8_t)(Y) + ut);                                                            \
            uint8_t d = 0;                                                                              \
            if (WITH_P) d = a > z ? 1 : 0;                                                              \
            z = z > a ? z : a;                                                                          \
            if (WITH_P && b > z) d = 2;                                                                 \
            uint8_t zu = (uint8_t)z > (uint8_t)b ? (uint8_t)z : (uint8_t)b;                             \
            zu = zu < max_sc_u ? zu : max_sc_u;                                                         \
            z = (int8_t)zu;                                                                             \
            (U) = (uint8_t)s8(z - vt1);                                                                 \
            (V) = (uint8_t)s8(z - ut);                                                                  \
            z = s8(z - q);                                                                              \
            a = s8(a - z);                                                                              \
            b = s8(b - z);                                                                              \
            (X) = (uint8_t)(a > 0 ? a : 0);                                                             \
            (Y) = (uint8_t)(b > 0 ? b : 0);                                                             \
            if (WITH_P) {                                                                               \
                if (a > 0) d |= 0x08;                                                                   \
                if (b > 0) d |= 0x10;                                                                   \
                prow_ptr[(unsigned)((T) - st)] = d;                                                     \
            }                                                                                           \
        }
        NUCLW_CELL(ta, ua, va, xa, ya, sa, nb & 0xFFu, (nb >> 8) & 0xFFu)
        NUCLW_CELL(tb, ub, vb, xb, yb, sb, (nb >> 16) & 0xFFu, (nb >> 24) & 0xFFu)
#undef NUCLW_CELL
        // exact maximum of the band (:207-250); ties follow the reference's scan: position en0 first, then the four
        // interleaved lanes of its 4-wide loop (lane by lane, ascending), then the scalar remainder
        int max_H, max_t;
        const int l_en0 = en0 & 63, l_st0 = st0 & 63;
        const bool odd_en0 = (en0 >> 6) & 1, odd_st0 = (st0 >> 6) & 1;
        if (r > 0) {
            int h_en0;
            if (en0 > 0) {
                const int l2 = (en0 - 1) & 63;
                const bool odd2 = ((en0 - 1) >> 6) & 1;
                h_en0 = NUCL_READLANE(odd2 ? Hb : Ha, l2) + NUCL_READLANE((int)(odd_en0 ? ub : ua), l_en0) - qe;
            } else {
                h_en0 = NUCL_READLANE(odd_en0 ? Hb : Ha, l_en0) + NUCL_READLANE((int)(odd_en0 ? vb : va), l_en0) - qe;
            }
            const int en1 = st0 + (en0 - st0) / 4 * 4;
            int best_h = h_en0;
            unsigned best_o = 0;
            if (ta >= st0 && ta < en0) {
                Ha = Ha + (int)va - qe;
                const unsigned o = ta < en1 ? 1u + (unsigned)((ta - st0) & 3) * 0x100000u + (unsigned)((ta - st0) >> 2)
                                            : 1u + 4u * 0x100000u + (unsigned)(ta - en1);
                if (Ha > best_h || (Ha == best_h && o < best_o)) { best_h = Ha; best_o = o; }
            }
            if (tb >= st0 && tb < en0) {
                Hb = Hb + (int)vb - qe;
                const unsigned o = tb < en1 ? 1u + (unsigned)((tb - st0) & 3) * 0x100000u + (unsigned)((tb - st0) >> 2)
                                            : 1u + 4u * 0x100000u + (unsigned)(tb - en1);
                if (Hb > best_h || (Hb == best_h && o < best_o)) { best_h = Hb; best_o = o; }
            }
            if (ta == en0) Ha = h_en0;
            if (tb == en0) Hb = h_en0;
            // (largest H, among those the smallest order) over the wave: two reductions on DPP row operations instead of
            // six dependent shuffle pairs through the LDS crossbar
            const int wave_h = NUCL_WAVE_MAX_I32(best_h);
            best_o = NUCL_WAVE_MIN_U32(best_h == wave_h ? best_o : 0xFFFFFFFFu);
            best_h = wave_h;
            max_H = best_h;
            if (best_o == 0) max_t = en0;
            else if (best_o <= 4u * 0x100000u) max_t = st0 + (int)((best_o - 1u) & 0xFFFFFu) * 4 + (int)((best_o - 1u) >> 20);
            else max_t = en1 + (int)(best_o - 1u - 4u * 0x100000u);
        } else {
            const int h0 = NUCL_READLANE((int)va, 0) - qe - qe;
            if (ta == 0) Ha = h0;
            max_H = h0;
            max_t = 0;
        }
        if (en0 == tlen - 1) {          // wave-uniform: only while the band touches the end of the target
            const int h_en0 = NUCL_READLANE(odd_en0 ? Hb : Ha, l_en0);
            if (h_en0 > ez.mte) { ez.mte = h_en0; ez.mte_q = r - en; }
        }
        if (r - st0 == qlen - 1) {      // ... the end of the query
            const int h_st0 = NUCL_READLANE(odd_st0 ? Hb : Ha, l_st0);
            if (h_st0 > ez.mqe) { ez.mqe = h_st0; ez.mqe_t = st0; }
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
        if (r == qlen + tlen - 2 && en0 == tlen - 1) {
            const int l3 = (tlen - 1) & 63;
            ez.score = NUCL_READLANE((((tlen - 1) >> 6) & 1) ? Hb : Ha, l3);
        }
        last_st = st;
        last_en = en;
    }
    NUCL_SYNC();
}

// ksw_backtrack (is_rot, no introns) from cell (i0 = target, j0 = query): the wave stages 64 rows of direction bytes in
// LDS, lane 0 walks through them; one letter per step, last column first, into w[]; returns the number of letters.
NUCL_HD int ksw_walk_wave(const uint8_t *p, int qlen, int tlen, int i0, int j0, char *w, WaveLds &S) {
    const int lane = NUCL_LANE();
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < KSW_W + 1 ? n_col_ : KSW_W + 1) + 15) / 16 + 1;
    const int n_col = n_col_ * 16;
    int i = i0, j = j0, state = 0, n = 0;
    while (i >= 0 && j >= 0) {      // wave-uniform: i, j are broadcast at the end of every round
        const int r_hi = i + j;                       // rows [r_lo, r_hi] staged
        const int r_lo = r_hi - (PROWS - 1) > 0 ? r_hi - (PROWS - 1) : 0;
        NUCL_SYNC();
        for (int k = lane; k < (r_hi - r_lo + 1) * n_col; k += NG) S.prow[k] = p[(size_t)r_lo * (size_t)n_col + (size_t)k];
        NUCL_SYNC();
        if (lane == 0) {
            while (i >= 0 && j >= 0 && i + j >= r_lo) {
                const int r = i + j;
                int st, en;
                band_of(r, qlen, tlen, st, en);
                int force_state = -1;
                if (i < st) force_state = 2;
                if (i > en) force_state = 1;
                const unsigned tmp = force_state < 0 ? S.prow[(r - r_lo) * n_col + (i - st)] : 0u;
                if (state == 0) state = tmp & 7;
                else if (!(tmp >> (state + 2) & 1)) state = 0;
                if (state == 0) state = tmp & 7;
                if (force_state >= 0) state = force_state;
                if (state == 0) { w[n++] = 'M'; --i; --j; }
                else if (state == 1 || state == 3) { w[n++] = 'D'; --i; }
                else { w[n++] = 'I'; --j; }
            }
        }
        i = NUCL_READLANE(i, 0);
        j = NUCL_READLANE(j, 0);
        n = NUCL_READLANE(n, 0);
    }
    if (lane == 0) {
        for (; i >= 0; --i) w[n++] = 'D';
        for (; j >= 0; --j) w[n++] = 'I';
    }
    n = NUCL_READLANE(n, 0);
    return n;
}

// One wavefront: pulls pairs from the queue until it is empty.
NUCL_HD void align_wave(const NuclLaunch &L, WaveLds &S, uint8_t *p, char *w) {
    const int lane = NUCL_LANE();
    for (;;) {
        NUCL_SYNC();     // keeps the lanes together across the back edge (see nucl_core.h, align_group)
        unsigned pi = NUCL_ATOMIC_ADD_U32(L.next_pair, lane == 0 ? 1u : 0u);
        pi = (unsigned)NUCL_READLANE((int)pi, 0);
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
            ksw_extz2_wave<false>(qr, q_rev_len, tr, tlen - t_start_rev, L.mat, L.gapo, L.gape, L.zdrop, S, nullptr, ez);
            const int q_start = qlen - (q_start_rev + ez.max_q) - 1, t_start = tlen - (t_start_rev + ez.max_t) - 1;
            // right extension with directions from that start (:183-196)
            SeqView qf = qv, tf = tv;
            qf.off = q_start;
            tf.off = t_start;
            int wq = wrapped && qlen - q_start > orig ? orig : qlen - q_start, wt = tlen - t_start;
            ksw_extz2_wave<true>(qf, wq, tf, wt, L.mat, L.gapo, L.gape, L.zdrop, S, p, eza);
            if (ez.max_q > eza.max_q && ez.max_t > eza.max_t) {
                // the forward pass fell short of the backward pass: the backward pass is redone with directions and
                // its CIGAR reversed (:201-210)
                wq = q_rev_len;
                wt = tlen - t_start_rev;
                ksw_extz2_wave<true>(qr, wq, tr, wt, L.mat, L.gapo, L.gape, L.zdrop, S, p, eza);
                walk_reversed = true;
            }
            NUCL_SYNC_MEM();   // the direction bytes were written by all lanes
            if (eza.max_t >= 0 && eza.max_q >= 0) n_bt = ksw_walk_wave(p, wq, wt, eza.max_t, eza.max_q, w, S);
            NUCL_SYNC_MEM();
            res.score = eza.max;
            res.q_start = q_start;
            res.q_end = q_start + eza.max_q;
            res.t_start = t_start;
            res.t_end = t_start + eza.max_t;
            walk_reversed = !walk_reversed;
        }
        NUCL_SYNC_MEM();   // w[] was written by lane 0 (or all lanes), every lane reads it below
        // ---- output: reserve space, copy the string in alignment order, count identities (:231-258)
        unsigned long long off = NUCL_ATOMIC_ADD_U64(L.bt_cursor, lane == 0 ? (unsigned long long)n_bt + 1ull : 0ull);
        off = NUCL_SHFL_U64(off, 0);
        const bool fits = off + (unsigned long long)n_bt + 1ull <= L.bt_cap;
        if (fits) {
            for (int i = lane; i < n_bt; i += NG) L.bt[off + (unsigned long long)i] = walk_reversed ? w[n_bt - 1 - i] : w[i];
            if (lane == 0) L.bt[off + (unsigned long long)n_bt] = 0;
        }
        // identities: every lane takes a contiguous chunk of the alignment; the letters before it say where it starts
        const int chunk = (n_bt + NG - 1) / NG;
        const int c0 = lane * chunk < n_bt ? lane * chunk : n_bt, c1 = c0 + chunk < n_bt ? c0 + chunk : n_bt;
        int dq = 0, dt = 0;
        for (int i = c0; i < c1; ++i) {
            const char c = walk_reversed ? w[n_bt - 1 - i] : w[i];
            dq += c != 'D';
            dt += c != 'I';
        }
        int pq = dq, pt = dt;        // inclusive prefix over the lanes
        for (int d = 1; d < NG; d <<= 1) {
            const int oq = NUCL_SHFL(pq, lane - d), ot = NUCL_SHFL(pt, lane - d);
            if (lane >= d) { pq += oq; pt += ot; }
        }
        int qp = res.q_start + pq - dq, tp = res.t_start + pt - dt;
        unsigned ids = 0;
        for (int i = c0; i < c1; ++i) {
            const char c = walk_reversed ? w[n_bt - 1 - i] : w[i];
            if (c == 'M') { ids += tv.strand(tp) == qv.strand(qp) ? 1u : 0u; ++qp; ++tp; }
            else if (c == 'I') ++qp;
            else ++tp;
        }
        for (int d = 1; d < NG; d <<= 1) ids += (unsigned)NUCL_SHFL_XOR((int)ids, d);
        if (lane == 0) {
            res.ident = ids;
            res.bt_off = fits ? off : 0ull;
            res.bt_len = (uint32_t)n_bt;
            if (!fits) res.status = MMGPU_NUCL_BT_OVERFLOW;
            L.out[L.order[pi]] = res;
        }
        NUCL_SYNC();
    }
}

}  // namespace nuclw
}  // namespace mmgpu

#endif
