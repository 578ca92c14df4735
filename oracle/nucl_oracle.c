/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the reference's nucleotide alignment step:
 *   BandedNucleotideAligner::align                      src/alignment/BandedNucleotideAligner.cpp:76-263
 *   DistanceCalculator::computeUngappedAlignment        src/alignment/DistanceCalculator.h:93-112
 *     ungappedAlignmentByDiagonal (RESCORE_MODE_ALIGNMENT) :115-175, computeSubstitutionStartEndDistance :178-200
 *   ksw_extz2_sse                                       lib/ksw2/ksw2_extz2_sse.cpp:44-285  (ksw2, MIT, vendored)
 *   ksw_backtrack / ksw_reset_extz / ksw_apply_zdrop    lib/ksw2/ksw2.h:134-199
 *
 * ksw_extz2_sse is a banded (w = 64) anti-diagonal DP on 8-bit DIFFERENCES (u, v, x, y) processed in blocks of 16
 * target positions.  A block is computed whole even where it sticks out of the band; the cells outside use whatever
 * the byte arrays hold there (stale values of earlier anti-diagonals, the calloc zeros, and - past the end of the
 * target copy - the reversed query that follows it in the same allocation), and the band later moves over some of
 * them.  The outcome therefore depends on the block structure and on the memory layout, and this restatement keeps
 * both: one zero-initialised byte buffer laid out like the reference's (u | v | x | y | s | target copy | reversed
 * query, ksw2_extz2_sse.cpp:107-109, a 16-byte aligned allocation), and the same block loop, written per byte.
 * Pinned against the real functions (oracle/ref_shim_nucl.cpp) by tests/test_nucl_oracle.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

#define KSW_NEG_INF (-0x40000000)
#define EZ_SCORE_ONLY 0x01
#define EZ_RIGHT 0x02
#define EZ_GENERIC_SC 0x04
#define EZ_APPROX_MAX 0x08
#define EZ_EXTZ_ONLY 0x40
#define EZ_REV_CIGAR 0x80

static void ez_reset(mmo_ksw_ez *ez) {   /* ksw2.h:175-180 */
    ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
    ez->max = 0;
    ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
    ez->n_cigar = 0;
    ez->zdropped = 0;
}

/* ksw2.h:182-199 with is_rot = 1 (a = anti-diagonal r, b = target position t) */
static int ez_zdrop(mmo_ksw_ez *ez, int32_t H, int r, int t, int zdrop, int e) {
    if (H > ez->max) {
        ez->max = H;
        ez->max_t = t;
        ez->max_q = r - t;
    } else if (t >= ez->max_t && r - t >= ez->max_q) {
        const int tl = t - ez->max_t, ql = (r - t) - ez->max_q;
        const int l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && ez->max - H > zdrop + l * e) {
            ez->zdropped = 1;
            return 1;
        }
    }
    return 0;
}

/* run-length CIGAR builder (ksw2.h:119-129): op 0 = M, 1 = I, 2 = D */
typedef struct {
    uint32_t *c;
    int n, cap;
} cigar_buf;

static void cigar_push(cigar_buf *b, uint32_t op, int len) {
    if (b->n == 0 || op != (b->c[b->n - 1] & 0xf)) {
        if (b->n == b->cap) {
            b->cap = b->cap ? b->cap * 2 : 4;
            b->c = (uint32_t *)realloc(b->c, (size_t)b->cap * 4);
        }
        b->c[b->n++] = (uint32_t)len << 4 | op;
    } else {
        b->c[b->n - 1] += (uint32_t)len << 4;
    }
}

/* ksw2.h:134-173 with is_rot = 1, with_N = 0: i walks the target, j the query */
static void backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0,
                      cigar_buf *out) {
    int i = i0, j = j0, state = 0;
    while (i >= 0 && j >= 0) {
        const int r = i + j;
        int force_state = -1;
        if (i < off[r]) force_state = 2;
        if (i > off_end[r]) force_state = 1;
        const uint32_t tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) { cigar_push(out, 0, 1); --i; --j; }
        else if (state == 1 || state == 3) { cigar_push(out, 2, 1); --i; }
        else { cigar_push(out, 1, 1); --j; }
    }
    if (i >= 0) cigar_push(out, 2, i + 1);
    if (j >= 0) cigar_push(out, 1, j + 1);
    if (!is_rev)
        for (i = 0; i < out->n >> 1; ++i) {
            const uint32_t t = out->c[i];
            out->c[i] = out->c[out->n - 1 - i];
            out->c[out->n - 1 - i] = t;
        }
}

static inline int8_t s8(int v) { return (int8_t)(uint8_t)v; }   /* low 8 bits, two's complement (_mm_add/sub_epi8) */

int mmo_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int q, int e,
                  int w, int zdrop, int flag, mmo_ksw_ez *ez, uint32_t *cigar, int cigar_cap) {
    const int with_cigar = !(flag & EZ_SCORE_ONLY);
    ez_reset(ez);
    if (flag & (EZ_APPROX_MAX | EZ_RIGHT | EZ_GENERIC_SC)) return -1;   /* not used by the reference's caller */
    if (m <= 0 || qlen <= 0 || tlen <= 0) return 0;
    const int qe = q + e;
    const int8_t sc_mch = mat[0], sc_mis = mat[1];
    const uint8_t max_sc_u = (uint8_t)s8(mat[0] + qe * 2);
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    int min_sc = mat[1];
    for (int t = 1; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
    if (-min_sc > 2 * qe) return 0;

    /* one buffer, the reference's layout (:107-109); byte views of the five difference arrays */
    uint8_t *mem = (uint8_t *)calloc((size_t)tlen_ * 6 + qlen_ + 1, 16);
    uint8_t *u = mem, *v = u + (size_t)tlen_ * 16, *x = v + (size_t)tlen_ * 16, *y = x + (size_t)tlen_ * 16,
            *s = y + (size_t)tlen_ * 16, *sf = s + (size_t)tlen_ * 16, *qr = sf + (size_t)tlen_ * 16;
    int32_t *H = (int32_t *)malloc((size_t)tlen_ * 16 * 4);
    for (int t = 0; t < tlen_ * 16; ++t) H[t] = KSW_NEG_INF;
    uint8_t *p = NULL;
    int *off = NULL, *off_end = NULL;
    if (with_cigar) {
        p = (uint8_t *)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
        off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
        off_end = off + qlen + tlen - 1;
    }
    for (int t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
    memcpy(sf, target, (size_t)tlen);

    int last_st = -1, last_en = -1;
    for (int r = 0; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1;
        const uint8_t *qrr = qr + (qlen - 1 - r);
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) {
            ez->zdropped = 1;
            break;
        }
        const int st0 = st, en0 = en;
        st = st / 16 * 16;
        en = (en + 16) / 16 * 16 - 1;
        /* what enters the first block from the left (:126-132) */
        int8_t x1, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { x1 = (int8_t)x[st - 1]; v1 = (int8_t)v[st - 1]; }
            else x1 = v1 = 0;
        } else {
            x1 = 0;
            v1 = r ? (int8_t)q : 0;
        }
        if (en >= r) {
            y[r] = 0;
            u[r] = r ? (uint8_t)q : 0;
        }
        /* scores of this anti-diagonal: whole 16-byte groups starting at st0 (:135-145); the last letter m-1 is a wildcard */
        for (int t = st0; t <= en0; t += 16)
            for (int k = 0; k < 16; ++k) {
                const uint8_t a = sf[t + k], b = qrr[t + k];
                int8_t sc = a == b ? sc_mch : sc_mis;
                if (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) sc = 0;
                s[t + k] = (uint8_t)sc;
            }
        uint8_t *pr = with_cigar ? p + ((size_t)r * n_col_ - st / 16) * 16 : NULL;
        if (with_cigar) { off[r] = st; off_end[r] = en; }
        /* _mm_cvtsi32_si128(int8) sign-extends: a negative carry-in would also set bytes 1..3 of the first block's
         * shifted x / v (:151-152).  The stored differences are non-negative inside the band; kept for the stale cells. */
        const uint8_t x1_hi = x1 < 0 ? 0xFF : 0, v1_hi = v1 < 0 ? 0xFF : 0;
        for (int t = st; t <= en; ++t) {
            /* the reference shifts x and v by one byte across blocks: cell t sees x[r-1][t-1], v[r-1][t-1]; the arrays
             * are overwritten in place, so the values of the previous anti-diagonal are carried in x1 / v1 */
            int8_t xt1 = x1, vt1 = v1;
            if (t - st >= 1 && t - st <= 3) {
                xt1 = (int8_t)((uint8_t)xt1 | x1_hi);
                vt1 = (int8_t)((uint8_t)vt1 | v1_hi);
            }
            x1 = (int8_t)x[t];
            v1 = (int8_t)v[t];
            int8_t z = s8((int8_t)s[t] + s8(qe * 2));
            int8_t a = s8(xt1 + vt1);
            const int8_t ut = (int8_t)u[t];
            int8_t b = s8((int8_t)y[t] + ut);
            uint8_t d = 0;
            if (with_cigar) d = a > z ? 1 : 0;
            z = z > a ? z : a;                                   /* signed max */
            if (with_cigar && b > z) d = 2;
            uint8_t zu = (uint8_t)z > (uint8_t)b ? (uint8_t)z : (uint8_t)b;   /* unsigned max (:67) */
            zu = zu < max_sc_u ? zu : max_sc_u;
            z = (int8_t)zu;
            u[t] = (uint8_t)s8(z - vt1);
            v[t] = (uint8_t)s8(z - ut);
            z = s8(z - q);
            a = s8(a - z);
            b = s8(b - z);
            x[t] = (uint8_t)(a > 0 ? a : 0);
            y[t] = (uint8_t)(b > 0 ? b : 0);
            if (with_cigar) {
                if (a > 0) d |= 0x08;
                if (b > 0) d |= 0x10;
                pr[t] = d;   /* pr is biased by -st: index t */
            }
        }
        /* exact maximum over the band with a 32-bit score per target position (:207-250) */
        int32_t max_H, max_t;
        if (r > 0) {
            max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] - qe : H[en0] + v[en0] - qe;
            max_t = en0;
            /* the reference scans [st0, en1) four positions at a time keeping a maximum per 4-lane and then takes the
             * first lane that beats the running maximum, and the remaining positions one by one; strict '>' in both,
             * and lanes compared in order, which equals: */
            const int en1 = st0 + (en0 - st0) / 4 * 4;
            int32_t HH[4], tt[4];
            for (int i = 0; i < 4; ++i) { HH[i] = max_H; tt[i] = max_t; }
            int t;
            for (t = st0; t < en1; t += 4)
                for (int i = 0; i < 4; ++i) {
                    H[t + i] += (int32_t)v[t + i] - qe;
                    if (H[t + i] > HH[i]) { HH[i] = H[t + i]; tt[i] = t; }
                }
            for (int i = 0; i < 4; ++i)
                if (max_H < HH[i]) { max_H = HH[i]; max_t = tt[i] + i; }
            for (; t < en0; ++t) {
                H[t] += (int32_t)v[t] - qe;
                if (H[t] > max_H) { max_H = H[t]; max_t = t; }
            }
        } else {
            H[0] = v[0] - qe - qe;
            max_H = H[0];
            max_t = 0;
        }
        if (en0 == tlen - 1 && H[en0] > ez->mte) { ez->mte = H[en0]; ez->mte_q = r - en; }
        if (r - st0 == qlen - 1 && H[st0] > ez->mqe) { ez->mqe = H[st0]; ez->mqe_t = st0; }
        if (ez_zdrop(ez, max_H, r, max_t, zdrop, e)) break;
        if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
        last_st = st;
        last_en = en;
    }
    free(mem);
    free(H);
    int n = 0;
    if (with_cigar) {
        cigar_buf cb = {NULL, 0, 0};
        const int rev_cigar = !!(flag & EZ_REV_CIGAR);
        if (!ez->zdropped && !(flag & EZ_EXTZ_ONLY)) backtrack(rev_cigar, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &cb);
        else if (ez->max_t >= 0 && ez->max_q >= 0) backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &cb);
        n = cb.n;
        for (int i = 0; i < n && i < cigar_cap; ++i) cigar[i] = cb.c[i];
        free(cb.c);
        free(p);
        free(off);
    }
    ez->n_cigar = n;
    return n;
}

/* DistanceCalculator.h:178-200: best-scoring ungapped segment (Kadane with the reference's tie rules) */
static void seed_segment(const uint8_t *a, const uint8_t *b, unsigned len, const int8_t *mat, int alph, int *start, int *end,
                         int *score_out) {
    int max_score = 0, max_end = 0, max_start = 0, min_pos = -1, score = 0;
    for (unsigned pos = 0; pos < len; pos++) {
        score += mat[a[pos] * alph + b[pos]];
        const int is_min = score <= 0;
        if (is_min) { score = 0; min_pos = (int)pos; }
        if (score > max_score) { max_end = (int)pos; max_start = min_pos + 1; max_score = score; }
    }
    *start = max_start;
    *end = max_end;
    *score_out = max_score;
}

typedef struct {
    int start, end;
    unsigned score, dist;
    int diagonal;
} seed_t;

/* DistanceCalculator.h:115-175, RESCORE_MODE_ALIGNMENT */
static seed_t seed_on_diagonal(const uint8_t *qs, unsigned qlen, const uint8_t *ts, unsigned tlen, int diagonal,
                               const int8_t *mat, int alph) {
    seed_t r = {-1, -1, 0, 0, 0};
    const unsigned dist = (unsigned)abs(diagonal);
    r.dist = dist;
    r.diagonal = diagonal;
    int s, e, sc;
    if (diagonal >= 0 && dist < qlen) {
        const unsigned len = tlen < qlen - dist ? tlen : qlen - dist;
        seed_segment(qs + dist, ts, len, mat, alph, &s, &e, &sc);
        r.start = s; r.end = e; r.score = (unsigned)sc;
    } else if (diagonal < 0 && dist < tlen) {
        const unsigned len = tlen - dist < qlen ? tlen - dist : qlen;
        seed_segment(qs, ts + dist, len, mat, alph, &s, &e, &sc);
        r.start = s; r.end = e; r.score = (unsigned)sc;
    }
    return r;
}

/* DistanceCalculator.h:93-112: the prefilter diagonal is a 16-bit value, every 65536-shift that fits is tried */
static seed_t seed_best(const uint8_t *qs, unsigned qlen, const uint8_t *ts, unsigned tlen, unsigned short diagonal,
                        const int8_t *mat, int alph) {
    seed_t best = {-1, -1, 0, 0, 0};
    for (unsigned d = 1; d <= 1 + tlen / 32768; d++) {
        const int real = (int)(-d * 65536 + diagonal);
        const seed_t t = seed_on_diagonal(qs, qlen, ts, tlen, real, mat, alph);
        if (t.score > best.score) best = t;
    }
    for (unsigned d = 0; d <= qlen / 65536; d++) {
        const int real = (int)(d * 65536 + diagonal);
        const seed_t t = seed_on_diagonal(qs, qlen, ts, tlen, real, mat, alph);
        if (t.score > best.score) best = t;
    }
    return best;
}

/* DistanceCalculator.h:56-90 (computeUngappedWrappedAlignment): the query is the sequence written twice; every window of half its
 * length that starts at a 65536-shift of the 16-bit diagonal is laid on the target from position 0.  The loop bounds are the
 * reference's unsigned comparisons. */
static seed_t seed_best_wrapped(const uint8_t *qs, unsigned qlen, const uint8_t *ts, unsigned tlen, unsigned short diagonal,
                                const int8_t *mat, int alph) {
    seed_t best = {-1, -1, 0, 0, 0};
    for (unsigned d = 1; (0u - d * 65536u + diagonal) > 0u - tlen; d++) {
        const int real = (int)((0u - d * 65536u + diagonal) + qlen / 2);
        seed_t t = seed_on_diagonal(qs + real, qlen / 2, ts, tlen, 0, mat, alph);
        t.diagonal += real;
        t.dist = (unsigned)abs(real);
        if (t.score > best.score) best = t;
    }
    for (unsigned d = 0; (d * 65536u + diagonal) < qlen / 2; d++) {
        const int real = (int)(d * 65536u + diagonal);
        seed_t t = seed_on_diagonal(qs + real, qlen / 2, ts, tlen, 0, mat, alph);
        t.diagonal += real;
        t.dist = (unsigned)abs(real);
        if (t.score > best.score) best = t;
    }
    return best;
}

/* BandedNucleotideAligner::align; wrapped = the caller's --wrapped-scoring (the query is the sequence written twice, qlen its
 * doubled length; BandedNucleotideAligner.cpp:98-113,131,146-147,171-174,189-191).  q_num / t_num: numeric codes (A C T G X = 0..4);
 * rev_lookup: NucleotideMatrix::reverseResidue.  past_end_q / past_end_t: the letter the reference finds one residue
 * past the end of the aligned query strand / of the target: SmithWaterman::seq_reverse is called with L where it
 * expects L - 1 (BandedNucleotideAligner.cpp:61,68,93, StripedSmithWaterman.h:224-233), so reversed[k] = seq[L - k]
 * for k = 0..L - the reversed copies are shifted by one and begin with stale buffer content, here an explicit input.
 * Returns 0, or -1 if bt_cap is too small. */
int mmo_nucl_align(const uint8_t *q_num, int qlen, const uint8_t *t_num, int tlen, const int8_t *mat, int alph,
                   const uint8_t *rev_lookup, int gapo, int gape, int zdrop, unsigned diagonal16, int reverse,
                   int past_end_q, int past_end_t, mmo_nucl_result *res, char *bt, int bt_cap) {
    return mmo_nucl_align_wrapped(q_num, qlen, t_num, tlen, mat, alph, rev_lookup, gapo, gape, zdrop, diagonal16, reverse, past_end_q,
                                  past_end_t, 0, res, bt, bt_cap);
}

int mmo_nucl_align_wrapped(const uint8_t *q_num, int qlen, const uint8_t *t_num, int tlen, const int8_t *mat, int alph,
                           const uint8_t *rev_lookup, int gapo, int gape, int zdrop, unsigned diagonal16, int reverse,
                           int past_end_q, int past_end_t, int wrapped, mmo_nucl_result *res, char *bt, int bt_cap) {
    uint8_t *qa = (uint8_t *)malloc((size_t)qlen + 1), *qrev = (uint8_t *)malloc((size_t)qlen + 1),
            *trev = (uint8_t *)malloc((size_t)tlen + 1);
    /* the strand that is aligned (:82-90, initQuery :62-69) and the (shifted) reversed copies for the left extension */
    for (int i = 0; i < qlen; i++) qa[i] = reverse ? rev_lookup[q_num[qlen - 1 - i]] : q_num[i];
    qa[qlen] = (uint8_t)past_end_q;
    for (int k = 0; k <= qlen; k++) qrev[k] = qa[qlen - k];
    for (int k = 0; k <= tlen; k++) trev[k] = k == 0 ? (uint8_t)past_end_t : t_num[tlen - k];
    int rc = 0;
    memset(res, 0, sizeof(*res));
    const int orig = wrapped ? qlen / 2 : qlen;      /* origQueryLen */
    const seed_t sd = !wrapped ? seed_best(qa, (unsigned)qlen, t_num, (unsigned)tlen, (unsigned short)diagonal16, mat, alph)
                      : (qlen >= tlen * 2 ? seed_best_wrapped(qa, (unsigned)qlen, t_num, (unsigned)tlen, (unsigned short)diagonal16, mat, alph)
                                          : seed_best(qa, (unsigned)(qlen / 2), t_num, (unsigned)tlen, (unsigned short)diagonal16, mat, alph));
    int qs, qe_, ts, te;
    if (sd.diagonal >= 0) { qs = sd.start + (int)sd.dist; qe_ = sd.end + (int)sd.dist; ts = sd.start; te = sd.end; }
    else { qs = sd.start; qe_ = sd.end; ts = sd.start + (int)sd.dist; te = sd.end + (int)sd.dist; }
    int n_bt = 0;
    if (qe_ - qs == orig - 1 && ts == 0 && te == tlen - 1) {   /* the seed spans both sequences (:130-160) */
        res->score = (int32_t)sd.score;
        res->q_start = qs; res->q_end = qe_; res->t_start = ts; res->t_end = te;
        res->cigar_len = 1;
        uint32_t ids = 0;
        for (int i = qs; i <= qe_; i++) ids += qa[i] == t_num[ts + (i - qs)];
        res->ident = ids;
        if (orig + 1 > bt_cap) rc = -1;
        else { memset(bt, 'M', (size_t)orig); bt[orig] = 0; }
        n_bt = orig;
    } else {
        /* left extension, score only, on the reversed sequences from the seed's end backwards (:165-181) */
        const int q_start_rev = qlen - qe_ - 1, t_start_rev = tlen - te - 1;
        mmo_ksw_ez ez, eza;
        int q_rev_len = qlen - q_start_rev;      /* queryRevLenToAlign (:171-174) */
        if (wrapped && q_rev_len > orig) q_rev_len = orig;
        mmo_ksw_extz2(q_rev_len, qrev + q_start_rev, tlen - t_start_rev, trev + t_start_rev, 5, mat, gapo, gape, 64,
                      zdrop, EZ_SCORE_ONLY | EZ_EXTZ_ONLY, &ez, NULL, 0);
        const int q_start = qlen - (q_start_rev + ez.max_q) - 1, t_start = tlen - (t_start_rev + ez.max_t) - 1;
        /* right extension with CIGAR from that start (:183-196) */
        const int cap = qlen + tlen + 2;
        uint32_t *cg = (uint32_t *)malloc((size_t)cap * 4);
        int q_len_fwd = qlen - q_start;          /* queryLenToAlign (:189-191) */
        if (wrapped && q_len_fwd > orig) q_len_fwd = orig;
        int n = mmo_ksw_extz2(q_len_fwd, qa + q_start, tlen - t_start, t_num + t_start, 5, mat, gapo, gape, 64, zdrop,
                              EZ_EXTZ_ONLY, &eza, cg, cap);
        if (ez.max_q > eza.max_q && ez.max_t > eza.max_t) {
            /* the forward pass fell short of the backward pass: the backward pass is redone with CIGAR and reversed (:201-210) */
            n = mmo_ksw_extz2(q_rev_len, qrev + q_start_rev, tlen - t_start_rev, trev + t_start_rev, 5, mat, gapo, gape,
                              64, zdrop, EZ_EXTZ_ONLY, &eza, cg, cap);
            for (int i = 0; i < n / 2; i++) { const uint32_t t = cg[i]; cg[i] = cg[n - 1 - i]; cg[n - 1 - i] = t; }
        }
        res->cigar_len = n;
        res->score = eza.max;
        res->q_start = q_start;
        res->q_end = q_start + eza.max_q;
        res->t_start = t_start;
        res->t_end = t_start + eza.max_t;
        uint32_t ids = 0;
        int tp = t_start, qp = q_start;
        for (int c = 0; c < n; c++) {   /* :231-258 */
            const uint32_t op = cg[c] & 0xf, len = cg[c] >> 4;
            for (uint32_t i = 0; i < len; i++) {
                char ch;
                if (op == 0) { ids += t_num[tp] == qa[qp]; ++qp; ++tp; ch = 'M'; }
                else if (op == 1) { ++qp; ch = 'I'; }
                else { ++tp; ch = 'D'; }
                if (n_bt + 1 < bt_cap) bt[n_bt] = ch; else rc = -1;
                n_bt++;
            }
        }
        if (n_bt < bt_cap) bt[n_bt] = 0;
        res->ident = ids;
        free(cg);
    }
    res->bt_len = n_bt;
    free(qa); free(qrev); free(trev);
    return rc;
}
