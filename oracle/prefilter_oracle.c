/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the k-mer / double-diagonal / ungapped prefilter of MMseqs2
 * (SURVEY.md section 8a rows a1-a9).  Every function cites the reference lines it restates; paths are
 * relative to /root/reference/src/prefiltering unless stated.  Pinned against the real reference classes
 * (oracle/_ref/libmmref.so, oracle/ref_shim_pref.cpp) by tests/test_prefilter_oracle.py and against
 * tests/golden/prefilter_vectors.npz.
 *
 * Deliberately sequential and simple: the same loops, in the same order, as the reference, so that the order
 * dependent parts (arrival order of index entries, bin order of CacheFriendlyOperations, stable bucket sort,
 * truncation before the final sort) come out identical.
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

/* ---------------------------------------------------------------------------------------------------------
 * ExtendedSubstitutionMatrix::calcScoreMatrix (ExtendedSubstitutionMatrix.cpp:20-71).
 * For every span-mer r (index = sum res[p]*kalph^p, Indexer.h int2index) the list of all span-mers sorted by
 * descending score, ties kept in the order of the cartesian product (first residue slowest,
 * :103-128), i.e. std::stable_sort over that enumeration (:55).  Output without the SIMD padding columns:
 * score/index are [n][n] with n = kalph^span. */
typedef struct {
    int16_t s;
    uint32_t idx;
    uint32_t ord;
} mmo_sm_tmp;

static int mmo_sm_cmp(const void *a, const void *b) {
    const mmo_sm_tmp *x = (const mmo_sm_tmp *)a, *y = (const mmo_sm_tmp *)b;
    if (x->s != y->s) return x->s > y->s ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord ? 1 : 0);
}

void mmo_pf_score_matrix(const int16_t *submat, int alphabet, int kalph, int span, int16_t *score, uint32_t *index) {
    size_t n = 1;
    for (int i = 0; i < span; i++) n *= (size_t)kalph;
    mmo_sm_tmp *tmp = (mmo_sm_tmp *)malloc(n * sizeof(mmo_sm_tmp));
    unsigned char a[8], b[8];
    /* enumeration order e -> residues: first residue is the slowest digit (createCartesianProduct :103-128) */
    for (size_t ei = 0; ei < n; ei++) {
        size_t t = ei;
        for (int p = span - 1; p >= 0; p--) {
            a[p] = (unsigned char)(t % (size_t)kalph);
            t /= (size_t)kalph;
        }
        size_t i_index = 0, pw = 1;
        for (int p = 0; p < span; p++) {
            i_index += a[p] * pw;
            pw *= (size_t)kalph;
        }
        for (size_t ej = 0; ej < n; ej++) {
            size_t u = ej;
            for (int p = span - 1; p >= 0; p--) {
                b[p] = (unsigned char)(u % (size_t)kalph);
                u /= (size_t)kalph;
            }
            size_t j_index = 0;
            pw = 1;
            short sc = 0;
            for (int p = 0; p < span; p++) {
                j_index += b[p] * pw;
                pw *= (size_t)kalph;
                sc = (short)(sc + submat[a[p] * alphabet + b[p]]); /* calcScore :78-84 */
            }
            tmp[ej].s = sc;
            tmp[ej].idx = (uint32_t)j_index;
            tmp[ej].ord = (uint32_t)ej;
        }
        qsort(tmp, n, sizeof(mmo_sm_tmp), mmo_sm_cmp);
        for (size_t z = 0; z < n; z++) {
            score[i_index * n + z] = tmp[z].s;
            index[i_index * n + z] = tmp[z].idx;
        }
    }
    free(tmp);
}

/* ---------------------------------------------------------------------------------------------------------
 * KmerGenerator::generateKmerList (KmerGenerator.cpp:108-184) with the divide strategy of
 * setDivideStrategy(three, two) (:42-87): k=6 -> steps {3,3}; k=7 -> steps {2,2,3} (after the std::reverse).
 * s3/i3 and s2/i2 are mmo_pf_score_matrix outputs (row stride = element count).  Returns the number of
 * similar k-mers written to out (capacity cap; counting continues past cap so the caller can detect it). */

static size_t ipow(size_t b, int e) {
    size_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}

#define MMO_MAX_KMER_RESULT ((size_t)262144 * 32) /* KmerGenerator.h:46 */

size_t mmo_pf_kmer_list(const mmo_pf_gen *g, const uint8_t *kmer, int threshold_in, uint64_t *out, size_t cap) {
    int steps[3], nsteps;
    if (g->k % 3 == 0) {
        nsteps = g->k / 3;
        for (int i = 0; i < nsteps; i++) steps[i] = 3;
    } else if (g->k % 3 == 1) {
        /* :57-71 then reversed :85-86; for k=7: {3,2,2} -> {2,2,3} */
        nsteps = g->k / 3 + 1;
        int t[3], c = 0;
        for (int i = 0; i < g->k / 3 - 1; i++) t[c++] = 3;
        t[c++] = 2;
        t[c++] = 2;
        for (int i = 0; i < nsteps; i++) steps[i] = t[nsteps - 1 - i];
    } else {
        nsteps = g->k / 3 + 1;
        int t[3], c = 0;
        for (int i = 0; i < g->k / 3; i++) t[c++] = 3;
        t[c++] = 2;
        for (int i = 0; i < nsteps; i++) steps[i] = t[nsteps - 1 - i];
    }
    const short threshold = (short)threshold_in;
    size_t mult[3], elem[3];
    const int16_t *srow[3];
    const uint32_t *irow[3];
    short high[3], rest[3];
    int before = 0;
    for (int i = 0; i < nsteps; i++) {
        size_t idx = 0, pw = 1;
        for (int p = 0; p < steps[i]; p++) {
            idx += kmer[before + p] * pw;
            pw *= (size_t)g->kalph;
        }
        mult[i] = ipow((size_t)g->kalph, before);
        elem[i] = ipow((size_t)g->kalph, steps[i]);
        const int16_t *S = steps[i] == 3 ? g->s3 : g->s2;
        const uint32_t *I = steps[i] == 3 ? g->i3 : g->i2;
        srow[i] = S + idx * elem[i];
        irow[i] = I + idx * elem[i];
        high[i] = srow[i][0];
        before += steps[i];
    }
    rest[nsteps - 1] = 0;
    for (int i = nsteps - 1; i >= 1; i--) rest[i - 1] = (short)(high[i] + rest[i]);

    short cutoff1 = (short)(threshold - rest[0]);
    /* first array = row of step 0; its scores are read from the matrix row, its indices copied while
     * score >= cutoff1 (:135-141) */
    size_t in_n = 0;
    while (in_n < elem[0] && srow[0][in_n] >= cutoff1) in_n++; /* the product loop breaks at the first score < cutoff1 */
    short *in_s = (short *)malloc((in_n + 1) * sizeof(short));
    uint64_t *in_i = (uint64_t *)malloc((in_n + 1) * sizeof(uint64_t));
    for (size_t p = 0; p < in_n; p++) {
        in_s[p] = srow[0][p];
        in_i[p] = irow[0][p];
    }
    for (int i = 0; i < nsteps - 1; i++) {
        /* calculateArrayProduct (:187-216) */
        const int16_t *s2 = srow[i + 1];
        const uint32_t *i2 = irow[i + 1];
        size_t n2 = elem[i + 1];
        size_t ocap = 1024, counter = 0;
        short *os = (short *)malloc(ocap * sizeof(short));
        uint64_t *oi = (uint64_t *)malloc(ocap * sizeof(uint64_t));
        for (size_t a = 0; a < in_n; a++) {
            short score_i = in_s[a];
            if (score_i < cutoff1) break;
            uint64_t kmer_i = in_i[a];
            short cutoff2 = (short)(threshold - score_i - rest[i + 1]);
            for (size_t b = 0; b < n2 && (counter + 1 < MMO_MAX_KMER_RESULT) && s2[b] >= cutoff2; b++) {
                if (counter == ocap) {
                    ocap *= 2;
                    os = (short *)realloc(os, ocap * sizeof(short));
                    oi = (uint64_t *)realloc(oi, ocap * sizeof(uint64_t));
                }
                os[counter] = (short)(score_i + s2[b]);
                oi[counter] = kmer_i + (uint64_t)i2[b] * mult[i + 1];
                counter++;
            }
            if (counter + 1 >= MMO_MAX_KMER_RESULT) break;
        }
        free(in_s);
        free(in_i);
        in_s = os;
        in_i = oi;
        in_n = counter;
        cutoff1 = -1000;
    }
    size_t w = in_n < cap ? in_n : cap;
    for (size_t z = 0; z < w; z++) out[z] = in_i[z];
    free(in_s);
    free(in_i);
    return in_n;
}

/* Profile queries: KmerGenerator::setDivideStrategy(ScoreMatrix **one) (KmerGenerator.cpp:32-41) - k steps of one
 * residue, step i reads the score-sorted row of query position pos + pattern[i] (Sequence::nextProfileKmer,
 * Sequence.cpp:354-365: profile_score / profile_index, 20 entries per row sorted by rankedDescSort20); the k-mer window
 * itself is all zeros (Sequence.h:400-406), so every row index is 0.  Same product / cutoff logic as above. */
size_t mmo_pf_kmer_list_profile(const int16_t *pscore, const uint32_t *pindex, int row, int k, const uint8_t *pat, int pos,
                                int kalph, int threshold_in, uint64_t *out, size_t cap) {
    const short threshold = (short)threshold_in;
    const int16_t *srow[16];
    const uint32_t *irow[16];
    size_t mult[16];
    short high[16], rest[16];
    const size_t elem = 20; /* Sequence::PROFILE_AA_SIZE */
    for (int i = 0; i < k; i++) {
        srow[i] = pscore + (size_t)(pos + pat[i]) * row;
        irow[i] = pindex + (size_t)(pos + pat[i]) * row;
        mult[i] = ipow((size_t)kalph, i);
        high[i] = srow[i][0];
    }
    rest[k - 1] = 0;
    for (int i = k - 1; i >= 1; i--) rest[i - 1] = (short)(high[i] + rest[i]);
    short cutoff1 = (short)(threshold - rest[0]);
    size_t in_n = 0;
    while (in_n < elem && srow[0][in_n] >= cutoff1) in_n++;
    short *in_s = (short *)malloc((in_n + 1) * sizeof(short));
    uint64_t *in_i = (uint64_t *)malloc((in_n + 1) * sizeof(uint64_t));
    for (size_t p = 0; p < in_n; p++) {
        in_s[p] = srow[0][p];
        in_i[p] = irow[0][p];
    }
    for (int i = 0; i < k - 1; i++) {
        const int16_t *s2 = srow[i + 1];
        const uint32_t *i2 = irow[i + 1];
        size_t ocap = 1024, counter = 0;
        short *os = (short *)malloc(ocap * sizeof(short));
        uint64_t *oi = (uint64_t *)malloc(ocap * sizeof(uint64_t));
        for (size_t a = 0; a < in_n; a++) {
            short score_i = in_s[a];
            if (score_i < cutoff1) break;
            uint64_t kmer_i = in_i[a];
            short cutoff2 = (short)(threshold - score_i - rest[i + 1]);
            for (size_t b = 0; b < elem && (counter + 1 < MMO_MAX_KMER_RESULT) && s2[b] >= cutoff2; b++) {
                if (counter == ocap) {
                    ocap *= 2;
                    os = (short *)realloc(os, ocap * sizeof(short));
                    oi = (uint64_t *)realloc(oi, ocap * sizeof(uint64_t));
                }
                os[counter] = (short)(score_i + s2[b]);
                oi[counter] = kmer_i + (uint64_t)i2[b] * mult[i + 1];
                counter++;
            }
            if (counter + 1 >= MMO_MAX_KMER_RESULT) break;
        }
        free(in_s);
        free(in_i);
        in_s = os;
        in_i = oi;
        in_n = counter;
        cutoff1 = -1000;
    }
    size_t w = in_n < cap ? in_n : cap;
    for (size_t z = 0; z < w; z++) out[z] = in_i[z];
    free(in_s);
    free(in_i);
    return in_n;
}

/* ---------------------------------------------------------------------------------------------------------
 * Sequence k-mer iteration (src/commons/Sequence.h:94-121,399; spaced patterns Sequence.h:24-27) */
/* spaced_seed_<k> of Sequence.h:20-50, one bit per pattern position (bit i = position i), and their lengths */
static const uint32_t MMO_SPACED_BITS[16] = {0, 0, 0, 0,
    /* 4 */ 0x17u,       /* 1 1 1 0 1 */
    /* 5 */ 0xA13u,      /* 1 1 0 0 1 0 0 0 0 1 0 1 */
    /* 6 */ 0x32Bu,      /* 1 1 0 1 0 1 0 0 1 1 */
    /* 7 */ 0x66Bu,      /* 1 1 0 1 0 1 1 0 0 1 1 */
    /* 8 */ 0xCEBu,      /* 1 1 0 1 0 1 1 1 0 0 1 1 */
    /* 9 */ 0x366Bu,     /* 1 1 0 1 0 1 1 0 0 1 1 0 1 1 */
    /* 10 */ 0x6D6Bu,    /* 1 1 0 1 0 1 1 0 1 0 1 1 0 1 1 */
    /* 11 */ 0x1B66Bu,   /* 1 1 0 1 0 1 1 0 0 1 1 0 1 1 0 1 1 */
    /* 12 */ 0x6B66Bu,   /* 1 1 0 1 0 1 1 0 0 1 1 0 1 1 0 1 0 1 1 */
    /* 13 */ 0xD6CEBu,   /* 1 1 0 1 0 1 1 1 0 0 1 1 0 1 1 0 1 0 1 1 */
    /* 14 */ 0x1B6CEBu,  /* 1 1 0 1 0 1 1 1 0 0 1 1 0 1 1 0 1 1 0 1 1 */
    /* 15 */ 0x6D1BD7u}; /* 1 1 1 0 1 0 1 1 1 1 0 1 1 0 0 0 1 0 1 1 0 1 1 */
static const uint8_t MMO_SPACED_LEN[16] = {0, 0, 0, 0, 5, 12, 10, 11, 12, 14, 15, 17, 19, 20, 21, 23};

int mmo_pf_pattern(int k, int spaced, uint8_t *pos_in_pattern) {
    if (!spaced) {
        for (int i = 0; i < k; i++) pos_in_pattern[i] = (uint8_t)i;
        return k;
    }
    if (k < 4 || k > 15) return -1;
    int plen = MMO_SPACED_LEN[k], c = 0;
    for (int i = 0; i < plen; i++)
        if ((MMO_SPACED_BITS[k] >> i) & 1u) pos_in_pattern[c++] = (uint8_t)i;
    return c == k ? plen : -1;
}

/* ---------------------------------------------------------------------------------------------------------
 * IndexTable::addKmerCount / addSequence / sortDBSeqLists (IndexTable.h:135-191,350-403) driven as
 * IndexBuilder::fillDatabase does for amino-acid targets with masking off (IndexBuilder.cpp:118-166,226-270):
 * per target, every window without X whose self score (sum of matrix diagonal, IndexBuilder.cpp:11-22) is
 * >= kmer_thr contributes ONE entry per distinct k-mer: (seqId, first position of that k-mer).
 * offsets has kalph^k + 1 entries; returns the number of entries (call with ids==NULL to size). */
typedef struct {
    uint32_t kmer;
    uint16_t pos;
} mmo_kp;
static int mmo_kp_cmp(const void *a, const void *b) {
    const mmo_kp *x = (const mmo_kp *)a, *y = (const mmo_kp *)b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

uint64_t mmo_pf_index_build(const uint8_t *tdata, const uint64_t *toff, uint32_t n, const int16_t *kmer_submat,
                            int alphabet, int k, int spaced, int kmer_thr, uint64_t *offsets, uint32_t *ids,
                            uint16_t *pos) {
    const int kalph = alphabet - 1;
    const size_t table = ipow((size_t)kalph, k);
    uint8_t pat[16];
    const int plen = mmo_pf_pattern(k, spaced, pat);
    size_t maxlen = 1;
    for (uint32_t t = 0; t < n; t++)
        if (toff[t + 1] - toff[t] > maxlen) maxlen = (size_t)(toff[t + 1] - toff[t]);
    mmo_kp *buf = (mmo_kp *)malloc((maxlen + 1) * sizeof(mmo_kp));
    uint64_t *cnt = (uint64_t *)calloc(table + 1, sizeof(uint64_t));
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && ids == NULL) break;
        for (uint32_t t = 0; t < n; t++) {
            const uint8_t *s = tdata + toff[t];
            const long L = (long)(toff[t + 1] - toff[t]);
            size_t c = 0;
            for (long i = 0; i + plen <= L; i++) {
                uint32_t idx = 0, pw = 1;
                int hasx = 0, self = 0;
                for (int p = 0; p < k; p++) {
                    uint8_t r = s[i + pat[p]];
                    if (r >= kalph) hasx = 1;
                    idx += r * pw;
                    pw *= (uint32_t)kalph;
                    self += (signed char)kmer_submat[r * alphabet + r]; /* char diagonalScore[] */
                }
                if (hasx) continue;
                if (kmer_thr > 0 && self < kmer_thr) continue;
                buf[c].kmer = idx;
                buf[c].pos = (uint16_t)i;
                c++;
            }
            qsort(buf, c, sizeof(mmo_kp), mmo_kp_cmp);
            uint32_t prev = UINT32_MAX;
            for (size_t z = 0; z < c; z++) {
                if (buf[z].kmer != prev) {
                    if (pass == 0) {
                        cnt[buf[z].kmer]++;
                    } else {
                        uint64_t o = cnt[buf[z].kmer]++;
                        ids[o] = t;
                        pos[o] = buf[z].pos;
                    }
                }
                prev = buf[z].kmer;
            }
        }
        if (pass == 0) {
            /* exclusive scan -> offsets; cnt becomes the running write cursor.  Targets are visited in id
             * order, so every list comes out sorted by (seqId, pos) as sortDBSeqLists leaves it. */
            uint64_t run = 0;
            for (size_t z = 0; z < table; z++) {
                uint64_t c2 = cnt[z];
                offsets[z] = run;
                cnt[z] = run;
                run += c2;
            }
            offsets[table] = run;
        }
    }
    uint64_t total = offsets[table];
    free(buf);
    free(cnt);
    return total;
}

/* ---------------------------------------------------------------------------------------------------------
 * UngappedAlignment::createProfile (UngappedAlignment.cpp:388-421): the per-position composition-bias term
 * aaCorrectionScore (:396-400); the profile entry for (pos, aa) is (char)(mat[q[pos]][aa] + corr[pos]). */
void mmo_pf_ungapped_corr(const float *bias, int qlen, int8_t *corr) {
    for (int p = 0; p < qlen; p++) {
        float b = bias ? bias[p] : 0.0f;
        b = (b < 0.0) ? b / 4 - 0.5 : b / 4 + 0.5; /* float/4, double +-0.5, back to float */
        corr[p] = (int8_t)(char)b;
    }
}

/* scalarDiagonalScoring (:45-57) over the overlap that computeSingelSequenceScores (:423-437) selects;
 * diagonal is the signed value (short)(u16) for sequences < 32768 (scoreSingleSequence :453-460). */
/* qprof != NULL: profile query - UngappedAlignment::createProfile's profile branch (:405-411): the row of query position p
 * is the alignment profile's column, [qlen][21] here, letter 20 (X) scores 0; q / corr / mat are not read. */
static int mmo_diag_score_p(const uint8_t *q, const int8_t *corr, int qlen, const int8_t *mat, int alphabet,
                            const uint8_t *t, int tlen, int diagonal, const int8_t *qprof) {
    int mind = diagonal < 0 ? -diagonal : diagonal;
    const uint8_t *qs, *ts;
    const int8_t *cs;
    int len, q0;
    if (diagonal >= 0 && mind < qlen) {
        len = tlen < qlen - mind ? tlen : qlen - mind;
        qs = q + mind;
        cs = corr + mind;
        q0 = mind;
        ts = t;
    } else if (diagonal < 0 && mind < tlen) {
        len = tlen - mind < qlen ? tlen - mind : qlen;
        qs = q;
        cs = corr;
        q0 = 0;
        ts = t + mind;
    } else {
        return 0;
    }
    int max = 0, score = 0;
    for (int p = 0; p < len; p++) {
        int curr = qprof ? (int)qprof[(size_t)(q0 + p) * 21 + ts[p]] : (signed char)(char)(mat[qs[p] * alphabet + ts[p]] + cs[p]);
        score += curr;
        score = score < 0 ? 0 : score;
        max = score > max ? score : max;
    }
    return max;
}

static int mmo_diag_score(const uint8_t *q, const int8_t *corr, int qlen, const int8_t *mat, int alphabet,
                          const uint8_t *t, int tlen, int diagonal) {
    return mmo_diag_score_p(q, corr, qlen, mat, alphabet, t, tlen, diagonal, NULL);
}

int mmo_pf_ungapped_score(const uint8_t *q, const int8_t *corr, int qlen, const int8_t *mat, int alphabet,
                          const uint8_t *t, int tlen, uint16_t diagonal) {
    return mmo_diag_score(q, corr, qlen, mat, alphabet, t, tlen, (int)(short)diagonal);
}

/* ---------------------------------------------------------------------------------------------------------
 * QueryMatcher::matchQuery (QueryMatcher.cpp:103-241) for amino-acid sequences with diagonal scoring on. */
typedef struct {
    uint32_t id;
    uint16_t diagonal;
    uint8_t count;
} mmo_cr; /* CounterResult, CacheFriendlyOperations.h:46-68 */



/* findDuplicates for diagonal scoring (CacheFriendlyOperations.cpp:38-49,185-278, computeTotalScore=false):
 * in: arrival-ordered (id, diag16); out: CounterResult list in bin order. */
static size_t mmo_find_duplicates_mode(const uint32_t *aid, const uint16_t *adiag, size_t n, uint32_t bins,
                                       uint32_t n_targets, mmo_cr *out, size_t out_cap, int total_score);
static size_t mmo_find_duplicates(const uint32_t *aid, const uint16_t *adiag, size_t n, uint32_t bins,
                                  uint32_t n_targets, mmo_cr *out, size_t out_cap) {
    return mmo_find_duplicates_mode(aid, adiag, n, bins, n_targets, out, out_cap, 0);
}
/* total_score != 0: computeTotalScore (:218-239, --diag-score 0): one element per target that has a flagged entry in the
 * bin - the first flagged entry's diagonal, count = number of flagged entries (saturating at 255) */
static size_t mmo_find_duplicates_mode(const uint32_t *aid, const uint16_t *adiag, size_t n, uint32_t bins,
                                       uint32_t n_targets, mmo_cr *out, size_t out_cap, int total_score) {
    uint32_t bits = 0;
    while ((1u << bits) < bins) bits++;
    size_t tabsz = ((size_t)n_targets >> bits) + 2;
    uint8_t *dup = (uint8_t *)calloc(tabsz, 1);
    /* hashIndexEntry (:341-351): stable scatter into bins by id & (bins-1) */
    size_t *bcnt = (size_t *)calloc(bins + 1, sizeof(size_t));
    for (size_t e = 0; e < n; e++) bcnt[(aid[e] & (bins - 1)) + 1]++;
    for (uint32_t b = 0; b < bins; b++) bcnt[b + 1] += bcnt[b];
    uint32_t *bid = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
    uint16_t *bdg = (uint16_t *)malloc((n + 1) * sizeof(uint16_t));
    size_t *cur = (size_t *)malloc(bins * sizeof(size_t));
    for (uint32_t b = 0; b < bins; b++) cur[b] = bcnt[b];
    for (size_t e = 0; e < n; e++) {
        size_t o = cur[aid[e] & (bins - 1)]++;
        bid[o] = aid[e];
        bdg[o] = adiag[e];
    }
    uint32_t *tid = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
    uint16_t *tdg = (uint16_t *)malloc((n + 1) * sizeof(uint16_t));
    size_t outn = 0;
    for (uint32_t b = 0; b < bins; b++) {
        size_t s = bcnt[b], e = bcnt[b + 1], ec = 0;
        for (size_t z = s; z < e; z++) { /* :194-208 */
            size_t h = bid[z] >> bits;
            uint8_t curd = (uint8_t)bdg[z], prevd = dup[h];
            tid[ec] = bid[z];
            tdg[ec] = bdg[z];
            ec += (curd == prevd);
            dup[h] = curd;
        }
        if (outn + ec >= out_cap) break; /* :214-216 */
        if (total_score) {
            for (size_t z = 0; z < ec; z++) dup[tid[z] >> bits] = 0; /* :219-222 */
            for (size_t z = 0; z < ec; z++) { /* :224-227 */
                size_t h = tid[z] >> bits;
                dup[h] += (dup[h] < 255) ? 1 : 0;
            }
            for (size_t z = 0; z < ec; z++) { /* :229-239 */
                size_t h = tid[z] >> bits;
                out[outn].id = tid[z];
                out[outn].count = dup[h];
                out[outn].diagonal = tdg[z];
                outn += (dup[h] != 0);
                dup[h] = 0;
            }
        } else {
            for (size_t z = ec; z-- > 0;) dup[tid[z] >> bits] = (uint8_t)((uint8_t)tdg[z] + 1); /* :242-247 */
            for (size_t z = 0; z < ec; z++) { /* :250-265 */
                size_t h = tid[z] >> bits;
                out[outn].id = tid[z];
                out[outn].count = 0;
                out[outn].diagonal = tdg[z];
                outn += (dup[h] != (uint8_t)tdg[z]);
                dup[h] = (uint8_t)tdg[z];
            }
        }
        for (size_t z = s; z < e; z++) dup[bid[z] >> bits] = 0; /* :268-275 */
    }
    free(dup);
    free(bcnt);
    free(bid);
    free(bdg);
    free(cur);
    free(tid);
    free(tdg);
    return outn;
}

/* keepMaxScoreElementOnly -> hashElements + keepMaxElement (CacheFriendlyOperations.cpp:73-80,325-338,354-384) */
static size_t mmo_keep_max(mmo_cr *io, size_t n, uint32_t bins, uint32_t n_targets) {
    uint32_t bits = 0;
    while ((1u << bits) < bins) bits++;
    size_t tabsz = ((size_t)n_targets >> bits) + 2;
    uint8_t *dup = (uint8_t *)calloc(tabsz, 1);
    size_t *bcnt = (size_t *)calloc(bins + 1, sizeof(size_t));
    for (size_t e = 0; e < n; e++) bcnt[(io[e].id & (bins - 1)) + 1]++;
    for (uint32_t b = 0; b < bins; b++) bcnt[b + 1] += bcnt[b];
    mmo_cr *tmp = (mmo_cr *)malloc((n + 1) * sizeof(mmo_cr));
    size_t *cur = (size_t *)malloc(bins * sizeof(size_t));
    for (uint32_t b = 0; b < bins; b++) cur[b] = bcnt[b];
    for (size_t e = 0; e < n; e++) tmp[cur[io[e].id & (bins - 1)]++] = io[e];
    size_t outn = 0;
    for (uint32_t b = 0; b < bins; b++) {
        for (size_t z = bcnt[b]; z < bcnt[b + 1]; z++) {
            size_t h = tmp[z].id >> bits;
            if (tmp[z].count > dup[h]) dup[h] = tmp[z].count;
        }
        for (size_t z = bcnt[b]; z < bcnt[b + 1]; z++) {
            size_t h = tmp[z].id >> bits;
            io[outn] = tmp[z];
            int found = dup[h] == tmp[z].count;
            outn += (size_t)found;
            dup[h] = (uint8_t)(dup[h] * (1 - found));
        }
    }
    free(dup);
    free(bcnt);
    free(tmp);
    free(cur);
    return outn;
}


/* hashElements (CacheFriendlyOperations.cpp:325-338): stable scatter of CounterResults into bins by id & (bins-1);
 * returns the bin boundaries (bins + 1 entries, caller frees) and fills tmp in bin order. */
static size_t *mmo_hash_elements(const mmo_cr *in, size_t n, uint32_t bins, mmo_cr *tmp) {
    size_t *bcnt = (size_t *)calloc(bins + 1, sizeof(size_t));
    for (size_t e = 0; e < n; e++) bcnt[(in[e].id & (bins - 1)) + 1]++;
    for (uint32_t b = 0; b < bins; b++) bcnt[b + 1] += bcnt[b];
    size_t *cur = (size_t *)malloc(bins * sizeof(size_t));
    for (uint32_t b = 0; b < bins; b++) cur[b] = bcnt[b];
    for (size_t e = 0; e < n; e++) tmp[cur[in[e].id & (bins - 1)]++] = in[e];
    free(cur);
    return bcnt;
}

/* mergeElementsByDiagonal(keepScoredHits = true) -> mergeDiagonalKeepScoredHitsDuplicates
 * (CacheFriendlyOperations.cpp:60-71,119-148): per bin the diagonal bytes + 1 are written in forward order, then
 * the bin is walked BACKWARDS; an element is kept when it already carries a score or its diagonal byte differs
 * from the element of the same target seen just before in that walk.  Output order: bins ascending, each reversed. */
static size_t mmo_merge_keep_scored(mmo_cr *io, size_t n, uint32_t bins, uint32_t n_targets) {
    uint32_t bits = 0;
    while ((1u << bits) < bins) bits++;
    uint8_t *dup = (uint8_t *)calloc(((size_t)n_targets >> bits) + 2, 1);
    mmo_cr *tmp = (mmo_cr *)malloc((n + 1) * sizeof(mmo_cr));
    size_t *bcnt = mmo_hash_elements(io, n, bins, tmp);
    size_t outn = 0;
    for (uint32_t b = 0; b < bins; b++) {
        for (size_t z = bcnt[b]; z < bcnt[b + 1]; z++) dup[tmp[z].id >> bits] = (uint8_t)((uint8_t)tmp[z].diagonal + 1);
        for (size_t z = bcnt[b + 1]; z-- > bcnt[b];) {
            size_t h = tmp[z].id >> bits;
            io[outn] = tmp[z];
            outn += (io[outn].count != 0 || dup[h] != (uint8_t)tmp[z].diagonal) ? 1 : 0;
            dup[h] = (uint8_t)tmp[z].diagonal;
        }
    }
    free(dup);
    free(tmp);
    free(bcnt);
    return outn;
}

/* mergeElementsByDiagonal(keepScoredHits = false) -> mergeDiagonalDuplicates (:60-71,83-115): the diagonal bytes + 1
 * are written in REVERSE order (so a target's slot ends with its first element's), then the bin is walked forwards
 * and an element is kept when its diagonal byte differs from the previous element of the same target. */
static size_t mmo_merge_diag_dup(mmo_cr *io, size_t n, uint32_t bins, uint32_t n_targets) {
    uint32_t bits = 0;
    while ((1u << bits) < bins) bits++;
    uint8_t *dup = (uint8_t *)calloc(((size_t)n_targets >> bits) + 2, 1);
    mmo_cr *tmp = (mmo_cr *)malloc((n + 1) * sizeof(mmo_cr));
    size_t *bcnt = mmo_hash_elements(io, n, bins, tmp);
    size_t outn = 0;
    for (uint32_t b = 0; b < bins; b++) {
        for (size_t z = bcnt[b + 1]; z-- > bcnt[b];) dup[tmp[z].id >> bits] = (uint8_t)((uint8_t)tmp[z].diagonal + 1);
        for (size_t z = bcnt[b]; z < bcnt[b + 1]; z++) {
            size_t h = tmp[z].id >> bits;
            io[outn] = tmp[z];
            outn += (dup[h] != (uint8_t)tmp[z].diagonal) ? 1 : 0;
            dup[h] = (uint8_t)tmp[z].diagonal;
        }
    }
    free(dup);
    free(tmp);
    free(bcnt);
    return outn;
}

/* UngappedAlignment::computeLongScore (UngappedAlignment.cpp:295-312): a sequence of 32768 residues or more does not fit the 16-bit
 * diagonal - every 65536-shift of it is scored and the best taken.  The shifts are computed in unsigned int and read back as int
 * (-devisions * 65536 + diagonal with an unsigned short diagonal), minDistToDiagonal = abs(realDiagonal). */
static int mmo_long_score_p(const uint8_t *q, const int8_t *corr, int qlen, const int8_t *mat, int alphabet, const uint8_t *t,
                            int tlen, uint16_t diagonal, const int8_t *qprof) {
    int total = 0;
    for (unsigned d = 1; d <= 1u + (unsigned)tlen / 32768u; d++) {
        int real = (int)(0u - d * 65536u + (unsigned)diagonal);
        int sc = mmo_diag_score_p(q, corr, qlen, mat, alphabet, t, tlen, real, qprof);
        total = sc > total ? sc : total;
    }
    for (unsigned d = 0; d <= (unsigned)qlen / 65536u; d++) {
        int real = (int)(d * 65536u + (unsigned)diagonal);
        int sc = mmo_diag_score_p(q, corr, qlen, mat, alphabet, t, tlen, real, qprof);
        total = sc > total ? sc : total;
    }
    return total;
}

/* UngappedAlignment::scoreSingleSequence (:453-461): what rescoreHits / getResult call for a saturated element */
static int mmo_single_score_p(const uint8_t *q, const int8_t *corr, int qlen, const int8_t *mat, int alphabet, const uint8_t *t,
                              int tlen, uint16_t diagonal, const int8_t *qprof) {
    if (qlen >= 32768 || tlen >= 32768) return mmo_long_score_p(q, corr, qlen, mat, alphabet, t, tlen, diagonal, qprof);
    return mmo_diag_score_p(q, corr, qlen, mat, alphabet, t, tlen, (int)(short)diagonal, qprof);
}

/* coverage counters for the tests (how often each path below ran since the last read): elements of a long query, long targets in
 * batches that were not full, long targets in full batches, of those the ones that took ANOTHER element's target's score or 0 */
static uint64_t mmo_long_stats[4];
void mmo_pf_long_stats(uint64_t out[4]) {
    for (int k = 0; k < 4; k++) {
        out[k] = mmo_long_stats[k];
        mmo_long_stats[k] = 0;
    }
}

#define MMO_DIAGONALBINSIZE 8 /* UngappedAlignment.h:57, the AVX2 build (4 without AVX2: the batches below depend on it) */

/* UngappedAlignment::scoreDiagonalAndUpdateHits (:187-293) for one batch of elements on one 16-bit diagonal.  For sequences below
 * 32768 residues a batch is only a way to score eight diagonals at once; with longer ones the code paths differ:
 *   - a query of 32768 residues or more: computeLongScore for every element (:199-208);
 *   - a batch that is not full (the rest of a diagonal, :277-291): computeLongScore for the elements with a long target;
 *   - a FULL batch (:210-276): the long targets enter the length sort with length 0, i.e. they come first, in their order (std::sort
 *     of eight elements is an insertion sort in libstdc++, stable), and score 0; the loop that writes the scores back then asks, for
 *     sorted position h holding a long target, whether the batch's element h IN ARRIVAL ORDER (hits[hitIdx], not
 *     hits[seqs[hitIdx].id], :269) has a long target, and if so gives the element at sorted position h the long score of THAT
 *     element's target.  The h-th long target of a full batch so receives computeLongScore of the batch's h-th element when that
 *     one is long, and 0 otherwise. */
static void mmo_score_batch(mmo_cr **hits, unsigned n, const mmo_pf_params *P, const uint8_t *q, const int8_t *corr, int qlen,
                            const int8_t *qprof) {
    const uint16_t diag = hits[0]->diagonal;
    if (qlen >= 32768) {
        for (unsigned h = 0; h < n; h++) {
            const uint8_t *t = P->tdata + P->toff[hits[h]->id];
            int tlen = (int)(P->toff[hits[h]->id + 1] - P->toff[hits[h]->id]);
            int sc = mmo_long_score_p(q, corr, qlen, P->ungapped_mat, P->alphabet, t, tlen, diag, qprof);
            hits[h]->count = (uint8_t)(sc > 255 ? 255 : sc);
            mmo_long_stats[0]++;
        }
        return;
    }
    if (n == MMO_DIAGONALBINSIZE) {
        unsigned n_long = 0, long_pos[MMO_DIAGONALBINSIZE];
        for (unsigned h = 0; h < n; h++) {
            const uint8_t *t = P->tdata + P->toff[hits[h]->id];
            int tlen = (int)(P->toff[hits[h]->id + 1] - P->toff[hits[h]->id]);
            if (tlen >= 32768) {
                long_pos[n_long++] = h;
                hits[h]->count = 0;
            } else {
                int sc = mmo_diag_score_p(q, corr, qlen, P->ungapped_mat, P->alphabet, t, tlen, (int)(short)diag, qprof);
                hits[h]->count = (uint8_t)(sc > 255 ? 255 : sc);
            }
        }
        for (unsigned h = 0; h < n_long; h++) { /* sorted position h = the h-th long target; hits[h] = arrival position h */
            const uint8_t *t2 = P->tdata + P->toff[hits[h]->id];
            int tlen2 = (int)(P->toff[hits[h]->id + 1] - P->toff[hits[h]->id]);
            if (tlen2 >= 32768) {
                int sc = mmo_long_score_p(q, corr, qlen, P->ungapped_mat, P->alphabet, t2, tlen2, diag, qprof);
                hits[long_pos[h]]->count = (uint8_t)(sc > 255 ? 255 : sc);
            }
            mmo_long_stats[2]++;
            if (long_pos[h] != h) mmo_long_stats[3]++;
        }
        return;
    }
    for (unsigned h = 0; h < n; h++) {
        const uint8_t *t = P->tdata + P->toff[hits[h]->id];
        int tlen = (int)(P->toff[hits[h]->id + 1] - P->toff[hits[h]->id]);
        int sc = tlen >= 32768 ? mmo_long_score_p(q, corr, qlen, P->ungapped_mat, P->alphabet, t, tlen, diag, qprof)
                               : mmo_diag_score_p(q, corr, qlen, P->ungapped_mat, P->alphabet, t, tlen, (int)(short)diag, qprof);
        hits[h]->count = (uint8_t)(sc > 255 ? 255 : sc);
        if (tlen >= 32768) mmo_long_stats[1]++;
    }
}

/* UngappedAlignment::computeScores (UngappedAlignment.cpp:315-346): only elements without a score are scored; they are collected per
 * 16-bit diagonal in array order and scored in batches of DIAGONALBINSIZE, the rest of every diagonal at the end.  Without a
 * sequence of 32768 residues or more the batches do not show in the result and the elements are scored one by one. */
static void mmo_score_unscored(mmo_cr *fd, size_t n, const mmo_pf_params *P, const uint8_t *q, const int8_t *corr, int qlen,
                               const int8_t *qprof) {
    int any_long = qlen >= 32768;
    for (size_t z = 0; z < n && !any_long; z++)
        if (fd[z].count == 0 && P->toff[fd[z].id + 1] - P->toff[fd[z].id] >= 32768) any_long = 1;
    if (!any_long) {
        for (size_t z = 0; z < n; z++) {
            if (fd[z].count != 0) continue;
            const uint8_t *t = P->tdata + P->toff[fd[z].id];
            int tlen = (int)(P->toff[fd[z].id + 1] - P->toff[fd[z].id]);
            int sc = mmo_diag_score_p(q, corr, qlen, P->ungapped_mat, P->alphabet, t, tlen, (int)(short)fd[z].diagonal, qprof);
            fd[z].count = (uint8_t)(sc > 255 ? 255 : sc);
        }
        return;
    }
    mmo_cr **matches = (mmo_cr **)malloc((size_t)65536 * MMO_DIAGONALBINSIZE * sizeof(mmo_cr *));
    uint8_t *counter = (uint8_t *)calloc(65536, 1);
    for (size_t z = 0; z < n; z++) {
        if (fd[z].count != 0) continue;
        const uint16_t d = fd[z].diagonal;
        matches[(size_t)d * MMO_DIAGONALBINSIZE + counter[d]] = &fd[z];
        if (++counter[d] == MMO_DIAGONALBINSIZE) {
            mmo_score_batch(&matches[(size_t)d * MMO_DIAGONALBINSIZE], MMO_DIAGONALBINSIZE, P, q, corr, qlen, qprof);
            counter[d] = 0;
        }
    }
    for (size_t d = 0; d < 65536; d++)
        if (counter[d]) mmo_score_batch(&matches[d * MMO_DIAGONALBINSIZE], counter[d], P, q, corr, qlen, qprof);
    free(matches);
    free(counter);
}

/* radixSortByScoreSize (QueryMatcher.cpp:536-561) */
static size_t mmo_radix_by_score(const unsigned *sizes, mmo_cr *w, unsigned thr, const mmo_cr *r, size_t n) {
    mmo_cr *ptr[256];
    mmo_cr *prev = w + n;
    for (int i = 0; i < 256; i++) {
        ptr[i] = prev - sizes[i];
        prev = ptr[i];
    }
    size_t above = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned s = r[i].count;
        if (s >= thr) {
            above++;
            *ptr[s]++ = r[i];
        }
    }
    return above;
}

static int mmo_hit_cmp(const void *a, const void *b) { /* hit_t::compareHitsByScoreAndId QueryMatcher.h:38-49 */
    const mmo_pf_hit *x = (const mmo_pf_hit *)a, *y = (const mmo_pf_hit *)b;
    int ax = abs(x->score), ay = abs(y->score);
    if (ax != ay) return ax > ay ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

/* Stage dumps (all optional, for stage-by-stage comparison with the device pipeline):
 *   thr_out[qlen]        per window start: adjusted k-mer threshold, -1 = no window / contains X
 *   nsim_out[qlen]       number of similar k-mers per window
 *   arr_id/arr_diag      arrival-ordered index entries (databaseHits), capacity arr_cap
 *   dd_*                 foundDiagonals after findDuplicates (bin order) + ungapped count, capacity dd_cap */

static int match_query_impl(const mmo_pf_params *P, const uint8_t *q, int qlen, const float *comp_bias, const mmo_pf_profile *prof,
                            uint32_t identity_id, mmo_pf_hit *hits, uint64_t hit_cap, uint64_t *n_hits, mmo_pf_stats *st,
                            mmo_pf_dump *dump);

int mmo_pf_match_query(const mmo_pf_params *P, const uint8_t *q, int qlen, const float *comp_bias,
                       uint32_t identity_id, mmo_pf_hit *hits, uint64_t hit_cap, uint64_t *n_hits, mmo_pf_stats *st,
                       mmo_pf_dump *dump) {
    return match_query_impl(P, q, qlen, comp_bias, NULL, identity_id, hits, hit_cap, n_hits, st, dump);
}

/* Profile query (DBTYPE_HMM_PROFILE): q = Sequence::numSequence (the profile's query letters: only the X test of the
 * window reads them, QueryMatcher.cpp:264), no composition bias (:110-114), similar k-mers from the profile's own sorted
 * rows, ungapped scores from its alignment profile. */
int mmo_pf_match_query_profile(const mmo_pf_params *P, const uint8_t *q, int qlen, const mmo_pf_profile *prof,
                               uint32_t identity_id, mmo_pf_hit *hits, uint64_t hit_cap, uint64_t *n_hits, mmo_pf_stats *st,
                               mmo_pf_dump *dump) {
    return match_query_impl(P, q, qlen, NULL, prof, identity_id, hits, hit_cap, n_hits, st, dump);
}

static int match_query_impl(const mmo_pf_params *P, const uint8_t *q, int qlen, const float *comp_bias, const mmo_pf_profile *prof,
                            uint32_t identity_id, mmo_pf_hit *hits, uint64_t hit_cap, uint64_t *n_hits, mmo_pf_stats *st,
                            mmo_pf_dump *dump) {
    const int k = P->gen->k, kalph = P->gen->kalph, alphabet = P->alphabet;
    int8_t *qprof = NULL;
    if (prof) {
        qprof = (int8_t *)calloc((size_t)qlen * 21 + 1, 1);
        for (int p = 0; p < qlen; p++)
            for (int a = 0; a < 20; a++) qprof[(size_t)p * 21 + a] = prof->aln[(size_t)a * qlen + p];
    }
    uint8_t pat[16];
    const int plen = mmo_pf_pattern(k, P->spaced, pat);
    mmo_pf_stats S;
    memset(&S, 0, sizeof(S));
    const size_t db = P->n_targets;
    const size_t found_cap = db > 1000000 ? db : 1000000;      /* foundDiagonalsSize :44 */
    const size_t max_db_matches = found_cap * 2;               /* :45 */
    size_t max_hits = P->max_hits < db ? (size_t)P->max_hits : db; /* :47 */

    /* ---- match() (:243-376), including the databaseHits overflow path (:310-346) ---- */
    size_t acap = 1 << 16, an = 0;
    uint32_t *aid = (uint32_t *)malloc(acap * sizeof(uint32_t));
    uint16_t *adg = (uint16_t *)malloc(acap * sizeof(uint16_t));
    size_t simcap = 1 << 20;
    uint64_t *sim = (uint64_t *)malloc(simcap * sizeof(uint64_t));
    mmo_cr *fd = (mmo_cr *)malloc((2 * found_cap + 16) * sizeof(mmo_cr)); /* foundDiagonals (+ radix ping-pong half) */
    size_t overflow_hits = 0, overflow_matches = 0;
    int8_t *corr = (int8_t *)malloc((size_t)qlen + 1);
    mmo_pf_ungapped_corr(comp_bias, qlen, corr);
    if (dump && dump->thr_out)
        for (int i = 0; i < qlen; i++) dump->thr_out[i] = -1;
    if (dump && dump->nsim_out)
        for (int i = 0; i < qlen; i++) dump->nsim_out[i] = 0;
    int aborted = 0;
    for (int i = 0; i + plen <= qlen && !aborted; i++) {
        uint8_t w[16];
        float bc = 0;
        int hasx = 0;
        for (int p = 0; p < k; p++) {
            bc += comp_bias ? comp_bias[i + (short)pat[p]] : 0.0f; /* :261-263 */
            w[p] = q[i + pat[p]];
            if (w[p] >= kalph) hasx = 1;
        }
        if (hasx) continue; /* :264-268 */
        short bias = (short)((bc < 0.0) ? bc - 0.5 : bc + 0.5); /* :270 */
        int t0 = P->kmer_thr - bias;
        short kthr = (short)(t0 > 0 ? t0 : 0); /* :271 */
        size_t ns;
        if (P->exact_kmer) { /* takeOnlyBestKmer (:279-282): the window's own k-mer */
            /* (the index's own base: kalph, or the full alphabet where the targets are profiles, Prefiltering.cpp:560-563) */
            const uint64_t base = P->index_base > 0 ? (uint64_t)P->index_base : (uint64_t)kalph;
            uint64_t idx = 0, pw = 1;
            for (int p = 0; p < k; p++) {
                idx += (uint64_t)w[p] * pw;
                pw *= base;
            }
            sim[0] = idx;
            ns = 1;
        } else {
            ns = prof ? mmo_pf_kmer_list_profile(prof->score, prof->index, prof->row, k, pat, i, kalph, kthr, sim, simcap)
                      : mmo_pf_kmer_list(P->gen, w, kthr, sim, simcap);
        }
        if (ns > simcap) {
            simcap = ns + 16;
            sim = (uint64_t *)realloc(sim, simcap * sizeof(uint64_t));
            ns = prof ? mmo_pf_kmer_list_profile(prof->score, prof->index, prof->row, k, pat, i, kalph, kthr, sim, simcap)
                      : mmo_pf_kmer_list(P->gen, w, kthr, sim, simcap);
        }
        S.kmer_list_len += ns;
        if (dump && dump->thr_out) dump->thr_out[i] = kthr;
        if (dump && dump->nsim_out) dump->nsim_out[i] = (uint32_t)ns;
        for (size_t z = 0; z < ns; z++) { /* :296-350 */
            uint64_t o0 = P->offsets[sim[z]], o1 = P->offsets[sim[z] + 1];
            size_t len = (size_t)(o1 - o0);
            if (an + len >= max_db_matches) { /* :310 (sequenceHits + seqListSize) >= lastSequenceHit */
                S.overflow++;
                if (P->kmer_score) { /* the merge of the segments by score (:514-533) is not restated */
                    aborted = 1;
                    break;
                }
                /* everything gathered so far (incl. the lists of this position) is matched on its own */
                size_t hc = mmo_find_duplicates(aid, adg, an, P->bins, P->n_targets, fd + overflow_hits,
                                                found_cap - overflow_hits);
                if (overflow_hits != 0) { /* second overflow onwards (:320-328) */
                    overflow_hits = mmo_merge_keep_scored(fd, hc + overflow_hits, P->bins, P->n_targets);
                    mmo_score_unscored(fd, overflow_hits, P, q, corr, qlen, qprof);
                    overflow_hits = mmo_keep_max(fd, overflow_hits, P->bins, P->n_targets);
                } else {
                    overflow_hits = hc;
                }
                overflow_matches += an;
                an = 0;
                if (len >= max_db_matches) { /* :343 goto outer */
                    aborted = 1;
                    break;
                }
            }
            if (an + len > acap) {
                while (an + len > acap) acap *= 2;
                aid = (uint32_t *)realloc(aid, acap * sizeof(uint32_t));
                adg = (uint16_t *)realloc(adg, acap * sizeof(uint16_t));
            }
            for (size_t e = 0; e < len; e++) {
                aid[an] = P->ids[o0 + e];
                adg[an] = (uint16_t)((unsigned short)i - P->pos[o0 + e]); /* hashIndexEntry :346 */
                an++;
            }
        }
    }
    S.db_matches = overflow_matches + an;
    if (dump && dump->arr_id)
        for (size_t e = 0; e < an && e < dump->arr_cap; e++) {
            dump->arr_id[e] = aid[e];
            dump->arr_diag[e] = adg[e];
        }

    if (P->kmer_score) {
        /* ---- diagonalScoring == false: match() ends with findDuplicates(computeTotalScore) and the histogram of the counts
         * (:353-371); matchQuery cuts at max(minDiagScoreThr, computeScoreThreshold) in radix order (:215-220) and getResult
         * <KMER_SCORE> writes the counts as prefScore (self hit: UCHAR_MAX, :410-413) ---- */
        size_t rs = 0;
        if (an > 0 && !S.overflow) rs = mmo_find_duplicates_mode(aid, adg, an, P->bins, P->n_targets, fd, found_cap, 1);
        S.double_hits = rs;
        S.after_keepmax = rs;
        unsigned sizes[256];
        memset(sizes, 0, sizeof(sizes));
        for (size_t z = 0; z < rs; z++) sizes[fd[z].count]++;
        size_t foundh = 0, thr = 0;
        for (thr = 255; thr > 0; thr--) {
            foundh += sizes[thr];
            if (foundh >= max_hits) break;
        }
        unsigned cut = (unsigned)thr > P->min_diag_score ? (unsigned)thr : P->min_diag_score;
        S.diag_thr = cut;
        if (rs >= found_cap / 2) S.big_list = 1;
        mmo_cr *wr = fd + rs;
        size_t above = mmo_radix_by_score(sizes, wr, cut, fd, rs);
        size_t cur = 0;
        if (identity_id != UINT32_MAX && cur < hit_cap) {
            hits[cur].id = identity_id;
            hits[cur].score = UCHAR_MAX;
            hits[cur].diagonal = 0;
            cur++;
        }
        for (size_t z = 0; z < above && cur < max_hits; z++) {
            if (wr[z].count >= (cut & 0xFFFFu) && wr[z].id != identity_id) {
                if (cur >= hit_cap) break;
                hits[cur].id = wr[z].id;
                hits[cur].score = (int)wr[z].count;
                hits[cur].diagonal = wr[z].diagonal;
                cur++;
            }
        }
        if (cur > 1) {
            if (identity_id != UINT32_MAX)
                qsort(hits + 1, cur - 1, sizeof(mmo_pf_hit), mmo_hit_cmp);
            else
                qsort(hits, cur, sizeof(mmo_pf_hit), mmo_hit_cmp);
        }
        *n_hits = cur;
        free(corr);
        free(fd);
        free(qprof);
    } else {
        /* ---- last segment (:353-362): findDuplicates, and after an overflow the merge with the earlier hits ---- */
        size_t rs = 0;
        if (an > 0) {
            rs = mmo_find_duplicates(aid, adg, an, P->bins, P->n_targets, fd + overflow_hits, found_cap - overflow_hits);
            if (overflow_hits != 0) rs = mmo_merge_diag_dup(fd, overflow_hits + rs, P->bins, P->n_targets);
        }
        S.double_hits = rs;
        /* ---- ungappedAlignment->align (:131): count = min(255, best ungapped score), unscored elements only ---- */
        mmo_score_unscored(fd, rs, P, q, corr, qlen, qprof);
        if (dump && dump->dd_id)
            for (size_t z = 0; z < rs && z < dump->dd_cap; z++) {
                dump->dd_id[z] = fd[z].id;
                dump->dd_diag[z] = fd[z].diagonal;
                dump->dd_count[z] = fd[z].count;
            }
        /* ---- :147-177 nucleotide searches: radix sort by score at the minimum diagonal score, then among the saturated elements
         * (count == 255) of one target the diagonal with the best exact score is written into the target's first element
         * (the elements are brought together by SORT_SERIAL(.., sortById) :154 - an unstable sort; which element is
         * "first" only matters when two diagonals tie on the exact score, reported as sat_tie), then keepMax.  The
         * restatement sorts the saturated prefix stably by id. ---- */
        if (P->nucleotide && rs < found_cap / 2) {
            unsigned sz[256];
            memset(sz, 0, sizeof(sz));
            for (size_t z = 0; z < rs; z++) sz[fd[z].count]++;
            mmo_cr *wr0 = fd + rs;
            size_t above = mmo_radix_by_score(sz, wr0, P->min_diag_score, fd, rs);
            size_t len = 0;
            while (len < above && wr0[len].count >= 255) len++;
            S.sat_len = (int)len;
            /* stable sort of the saturated prefix by id (insertion into a temporary by counting would do; the prefix is short) */
            for (size_t a = 1; a < len; a++) {
                mmo_cr t = wr0[a];
                size_t b = a;
                while (b > 0 && wr0[b - 1].id > t.id) {
                    wr0[b] = wr0[b - 1];
                    b--;
                }
                wr0[b] = t;
            }
            for (size_t a = 0; a < len;) {
                size_t e = a + 1;
                while (e < len && wr0[e].id == wr0[a].id) e++;
                if (e - a > 1) {
                    const uint8_t *t = P->tdata + P->toff[wr0[a].id];
                    int tlen = (int)(P->toff[wr0[a].id + 1] - P->toff[wr0[a].id]);
                    unsigned best = 0;
                    uint16_t bd = wr0[a].diagonal;
                    int ties = 0;
                    for (size_t z = a; z < e; z++) {
                        unsigned sc = (unsigned)mmo_diag_score_p(q, corr, qlen, P->ungapped_mat, alphabet, t, tlen, (int)(short)wr0[z].diagonal, qprof);
                        if (z == a || sc > best) {
                            best = sc;
                            bd = wr0[z].diagonal;
                            ties = 0;
                        } else if (sc == best && wr0[z].diagonal != bd) {
                            ties = 1;
                        }
                    }
                    if (ties) S.sat_tie = 1;
                    wr0[a].diagonal = bd;
                }
                a = e;
            }
            memmove(fd, wr0, above * sizeof(mmo_cr));
            rs = above;
        }
        rs = mmo_keep_max(fd, rs, P->bins, P->n_targets);
        S.after_keepmax = rs;
        mmo_cr *rd = fd, *wr = fd + rs;
        unsigned sizes[256];
        memset(sizes, 0, sizeof(sizes));
        for (size_t z = 0; z < rs; z++) sizes[rd[z].count]++; /* updateScoreBins :183 */
        /* computeScoreThreshold QueryMatcher.h:211-221 */
        size_t foundh = 0, thr = 0;
        for (thr = 255; thr > 0; thr--) {
            foundh += sizes[thr];
            if (foundh >= max_hits) break;
        }
        unsigned diag_thr = (unsigned)thr > P->min_diag_score ? (unsigned)thr : P->min_diag_score;
        S.diag_thr = diag_thr;
        size_t cur = 0;
        /* Restated: the sorted branch, rs < foundDiagonalsSize/2 (:188-203).  A query that leaves max(1M, dbSize)/2 elements or more
         * takes :204-214 (filter, unstable std::sort, no rescoring) in the reference - not restated here; S.after_keepmax tells a
         * caller which branch the reference would take, and the device hands such queries to the host (MMGPU_PF_SAT_TIE). */
        const unsigned max_diag_thr = 255; /* UCHAR_MAX - getQueryBias() (=0) */
        int truncated = diag_thr >= max_diag_thr;
        S.truncated = truncated;
        size_t above = mmo_radix_by_score(sizes, wr, diag_thr, rd, rs);
        {
            mmo_cr *t = rd;
            rd = wr;
            wr = t;
        }
        unsigned thr_final = diag_thr;
        int rescale = 0;
        int8_t *corr2 = corr;
        if (truncated) {
            /* rescoreHits (:563-586) */
            memset(sizes, 0, sizeof(sizes));
            int self = mmo_diag_score_p(q, corr2, qlen, P->ungapped_mat, alphabet, q, qlen, 0, qprof);
            int ms = self - (int)max_diag_thr;
            ms = ms > 1 ? ms : 1;
            ms = ms < USHRT_MAX ? ms : USHRT_MAX;
            float fms = (float)ms;
            size_t el = 0;
            for (size_t z = 0; z < above && rd[z].count >= max_diag_thr; z++) {
                const uint8_t *t = P->tdata + P->toff[rd[z].id];
                int tlen = (int)(P->toff[rd[z].id + 1] - P->toff[rd[z].id]);
                unsigned ns = (unsigned)mmo_single_score_p(q, corr2, qlen, P->ungapped_mat, alphabet, t, tlen, rd[z].diagonal, qprof);
                ns -= max_diag_thr;
                float sc = (float)(ns < (unsigned)USHRT_MAX ? ns : (unsigned)USHRT_MAX);
                rd[z].count = (unsigned char)((sc / fms) * (float)UCHAR_MAX + 0.5);
                sizes[rd[z].count] += 1;
                el++;
            }
            above = mmo_radix_by_score(sizes, wr, 0, rd, el);
            mmo_cr *t = rd;
            rd = wr;
            wr = t;
            thr_final = 0;
            rescale = ms;
        }
        /* getResult<UNGAPPED_DIAGONAL_SCORE> (:401-458) */
        if (identity_id != UINT32_MAX && cur < hit_cap) {
            hits[cur].id = identity_id;
            hits[cur].score = USHRT_MAX;
            hits[cur].diagonal = 0;
            cur++;
        }
        for (size_t z = 0; z < above && cur < max_hits; z++) {
            unsigned sc = rd[z].count;
            if (sc >= (thr_final & 0xFFFFu) && rd[z].id != identity_id) {
                if (cur >= hit_cap) break;
                hits[cur].id = rd[z].id;
                hits[cur].score = (int)sc;
                hits[cur].diagonal = rd[z].diagonal;
                if (rescale != 0) {
                    unsigned nsx = 255u;
                    nsx += (sc * (unsigned)rescale / 255u);
                    hits[cur].score = (int)nsx;
                } else if ((int)sc >= 255) {
                    const uint8_t *t = P->tdata + P->toff[rd[z].id];
                    int tlen = (int)(P->toff[rd[z].id + 1] - P->toff[rd[z].id]);
                    hits[cur].score = mmo_single_score_p(q, corr2, qlen, P->ungapped_mat, alphabet, t, tlen, rd[z].diagonal, qprof);
                }
                cur++;
            }
        }
        /* final sort (:233-239) */
        if (cur > 1) {
            if (identity_id != UINT32_MAX)
                qsort(hits + 1, cur - 1, sizeof(mmo_pf_hit), mmo_hit_cmp);
            else
                qsort(hits, cur, sizeof(mmo_pf_hit), mmo_hit_cmp);
        }
        *n_hits = cur;
        free(corr);
        free(fd);
        free(qprof);
    }
    free(aid);
    free(adg);
    free(sim);
    if (st) *st = S;
    return 0;
}
