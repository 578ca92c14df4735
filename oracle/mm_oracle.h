/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 * Declarations of the plain-C oracle (oracle/ .c files).  See each .c file for the reference
 * file:line every function restates. */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t score;   /* s_align.score1 */
    int32_t q_start, q_end, t_start, t_end;
    int32_t word;    /* 1 when the reference re-runs in int16 (score + bias >= 255) */
    uint32_t ident;  /* identicalAACnt */
    int32_t bt_len;
} mmo_sw_res;

/* sw_oracle.c */
void mmo_round_comp_bias(const float *bias_f, int qlen, int8_t *out);
int mmo_sw_bias(const int8_t *mat, int alphabet, const int8_t *comp_bias, int qlen);
int mmo_sw_check_params(const int8_t *mat, int alphabet, const int8_t *comp_bias, int qlen, int gap_open,
                        int gap_extend);
void mmo_sw_profile(const uint8_t *q, int qlen, const int8_t *comp_bias, const int8_t *mat, int alphabet,
                    int16_t *prof);
void mmo_sw_score_end(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *t, int tlen,
                      const int8_t *mat, int alphabet, int gap_open, int gap_extend, mmo_sw_res *r);
int mmo_sw_start(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *t, const int8_t *mat,
                 int alphabet, int gap_open, int gap_extend, mmo_sw_res *r);
float mmo_sw_cov(unsigned int start, unsigned int end, unsigned int len);
int mmo_sw_banded_backtrace(const uint8_t *t, const uint8_t *q, const int8_t *comp_bias, int tlen, int qlen,
                            int score, int gap_open, int gap_extend, const int8_t *mat, int alphabet, char *bt,
                            int bt_cap);
int mmo_sw_align(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *t, int tlen,
                 const int8_t *mat, int alphabet, int gap_open, int gap_extend, int need_start, int need_bt,
                 mmo_sw_res *r, char *bt, int bt_cap);
/* profile query (DBTYPE_HMM_PROFILE): profile = int8 [profile_letters][qlen], cons = consensus sequence */
int mmo_sw_align_profile(const int8_t *profile, int profile_letters, const uint8_t *cons, int qlen, const uint8_t *t,
                         int tlen, int alphabet, int gap_open, int gap_extend, int need_start, int need_bt,
                         mmo_sw_res *r, char *bt, int bt_cap);
int mmo_sw_score_identical(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *t,
                           const int8_t *mat, int alphabet);
/* batch driver used by tests and bench.py's cpu_baseline ("port" kind): one query, n targets from a
 * flat residue array with offsets, score/end only. */
void mmo_sw_batch_score(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *tdata,
                        const uint64_t *toff, const uint32_t *ids, int n, const int8_t *mat, int alphabet,
                        int gap_open, int gap_extend, int32_t *score, int32_t *q_end, int32_t *t_end,
                        int32_t *word);

/* prefilter_oracle.c */
typedef struct {
    uint32_t id;       /* hit_t::seqId   (QueryMatcher.h:33-49) */
    int32_t score;     /* hit_t::prefScore */
    uint16_t diagonal; /* hit_t::diagonal */
} mmo_pf_hit;

/* similar-k-mer generator inputs: mmo_pf_score_matrix outputs for span 3 and 2 (row stride = element count) */
typedef struct {
    int k, kalph;
    const int16_t *s3;
    const uint32_t *i3;
    const int16_t *s2;
    const uint32_t *i2;
} mmo_pf_gen;

typedef struct {
    /* inputs */
    const mmo_pf_gen *gen;
    int alphabet;
    int spaced;
    int kmer_thr;
    const uint64_t *offsets;
    const uint32_t *ids;
    const uint16_t *pos;
    const uint8_t *tdata; /* SequenceLookup */
    const uint64_t *toff;
    uint32_t n_targets;
    const int8_t *ungapped_mat; /* alphabet x alphabet */
    uint32_t bins;              /* CacheFriendlyOperations<BINS>, QueryMatcher.cpp:460-488 */
    uint64_t max_hits;          /* maxHitsPerQuery before min(., dbSize) */
    uint32_t min_diag_score;
    int exact_kmer;             /* takeOnlyBestKmer (--exact-kmer-matching; always on in nucleotide searches, Search.cpp:186) */
    int nucleotide;             /* matchQuery's isNucleotide branch (QueryMatcher.cpp:147-177) */
    int kmer_score;             /* --diag-score 0 (diagonalScoring == false): the prefilter score of a target is the number of its
                                   double k-mer matches, no ungapped scoring (QueryMatcher.cpp:215-232, CacheFriendlyOperations.cpp:218-239) */
    int index_base;             /* exact_kmer: base of the k-mer index if it is not alphabet - 1 (0 = alphabet - 1): the full alphabet
                                   for an index over profile targets (Prefiltering.cpp:560-563) */
} mmo_pf_params;

typedef struct {
    uint64_t db_matches;   /* statistics_t::dbMatches */
    uint64_t kmer_list_len; /* sum of similar k-mers over positions */
    uint64_t double_hits;   /* foundDiagonals after findDuplicates */
    uint64_t after_keepmax;
    uint32_t diag_thr;
    int truncated;
    int overflow; /* 1 = the databaseHits overflow path would trigger (not restated in kmer_score mode) */
    int big_list; /* kmer_score mode: the element list reached foundDiagonalsSize / 2, where the reference sorts with an
                     unstable std::sort (QueryMatcher.cpp:221-231): not restated */
    int sat_tie;  /* nucleotide branch: a target has two saturated (>= 255) elements on different diagonals with the same exact
                     score - the reference's choice then depends on the element order its std::sort left (:154) */
    int sat_len;  /* nucleotide branch: number of saturated elements of the query = length of the range that std::sort sorts.  Up to 16
                     libstdc++ sorts by insertion (stable), and the restatement's stable choice IS the reference's; beyond, introsort */
} mmo_pf_stats;

typedef struct {
    int32_t *thr_out;
    uint32_t *nsim_out;
    uint32_t *arr_id;
    uint16_t *arr_diag;
    uint64_t arr_cap;
    uint32_t *dd_id;
    uint16_t *dd_diag;
    uint8_t *dd_count;
    uint64_t dd_cap;
} mmo_pf_dump;

void mmo_pf_score_matrix(const int16_t *submat, int alphabet, int kalph, int span, int16_t *score, uint32_t *index);
/* profile query (Sequence with DBTYPE_HMM_PROFILE and kmerSize != 0, Sequence.cpp:301-352) */
typedef struct {
    const int16_t *score;  /* Sequence::profile_score: [qlen][row], 20 scores per position sorted descending */
    const uint32_t *index; /* Sequence::profile_index: [qlen][row], the letters in that order */
    int row;               /* Sequence::profile_row_size */
    const int8_t *aln;     /* Sequence::getAlignmentProfile(): [20][qlen] */
} mmo_pf_profile;
size_t mmo_pf_kmer_list_profile(const int16_t *pscore, const uint32_t *pindex, int row, int k, const uint8_t *pat, int pos,
                                int kalph, int threshold, uint64_t *out, size_t cap);
int mmo_pf_match_query_profile(const mmo_pf_params *P, const uint8_t *q, int qlen, const mmo_pf_profile *prof,
                               uint32_t identity_id, mmo_pf_hit *hits, uint64_t hit_cap, uint64_t *n_hits, mmo_pf_stats *st,
                               mmo_pf_dump *dump);
size_t mmo_pf_kmer_list(const mmo_pf_gen *g, const uint8_t *kmer, int threshold, uint64_t *out, size_t cap);
int mmo_pf_pattern(int k, int spaced, uint8_t *pos_in_pattern);
uint64_t mmo_pf_index_build(const uint8_t *tdata, const uint64_t *toff, uint32_t n, const int16_t *kmer_submat,
                            int alphabet, int k, int spaced, int kmer_thr, uint64_t *offsets, uint32_t *ids,
                            uint16_t *pos);
void mmo_pf_ungapped_corr(const float *bias, int qlen, int8_t *corr);
int mmo_pf_ungapped_score(const uint8_t *q, const int8_t *corr, int qlen, const int8_t *mat, int alphabet,
                          const uint8_t *t, int tlen, uint16_t diagonal);
int mmo_pf_match_query(const mmo_pf_params *P, const uint8_t *q, int qlen, const float *comp_bias,
                       uint32_t identity_id, mmo_pf_hit *hits, uint64_t hit_cap, uint64_t *n_hits, mmo_pf_stats *st,
                       mmo_pf_dump *dump);
/* coverage of the long-sequence paths of the ungapped scoring since the last call (tests): elements of queries of 32768 residues or
 * more, long targets in batches that were not full, long targets in full batches, of those the ones scored with another element's target */
void mmo_pf_long_stats(uint64_t out[4]);


/* compbias_oracle.c */
void mmo_comp_bias(const int16_t *submat /*alphabet^2, row-major short matrix*/, const double *pback, int alphabet,
                   const uint8_t *seq, int n, float scale, float *out);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* ---- nucleotide alignment step (oracle/nucl_oracle.c) ---- */
typedef struct {
    int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar;
} mmo_ksw_ez;
typedef struct {
    int32_t score, q_start, q_end, t_start, t_end;
    uint32_t ident;
    int32_t bt_len, cigar_len;
} mmo_nucl_result;
int mmo_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int q, int e,
                  int w, int zdrop, int flag, mmo_ksw_ez *ez, uint32_t *cigar, int cigar_cap);
int mmo_nucl_align(const uint8_t *q_num, int qlen, const uint8_t *t_num, int tlen, const int8_t *mat, int alph,
                   const uint8_t *rev_lookup, int gapo, int gape, int zdrop, unsigned diagonal16, int reverse,
                   int past_end_q, int past_end_t, mmo_nucl_result *res, char *bt, int bt_cap);
/* the same with the caller's --wrapped-scoring: q_num is the query written twice (qlen = the doubled length) */
int mmo_nucl_align_wrapped(const uint8_t *q_num, int qlen, const uint8_t *t_num, int tlen, const int8_t *mat, int alph,
                   const uint8_t *rev_lookup, int gapo, int gape, int zdrop, unsigned diagonal16, int reverse,
                   int past_end_q, int past_end_t, int wrapped, mmo_nucl_result *res, char *bt, int bt_cap);

/* ---- block aligner (block_oracle.c): lib/block-aligner 0.4.0, AVX2 configuration, as the reference calls it ---- */
typedef struct { int32_t score; uint32_t query_idx, reference_idx; } mmo_block_res;
int mmo_block_align(const uint8_t *q, const int16_t *qbias, int qlen, const uint8_t *r, const int16_t *rbias, int rlen, const int8_t *mat,
                    int alphabet, int gap_open /* < 0 */, int gap_extend /* < 0 */, int min_size, int max_size, int x_drop, mmo_block_res *res,
                    uint8_t *ops /* walk order, end -> origin: 1 = M, 4 = I, 5 = D; may be NULL */, uint32_t ops_cap, uint32_t *n_ops);
int mmo_block_align_table(const uint8_t *q, const int16_t *qbias, int qlen, const uint8_t *r, const int16_t *rbias, int rlen,
                          const int8_t *scores27x32, int gap_open, int gap_extend, int min_size, int max_size, int x_drop, mmo_block_res *res,
                          uint8_t *ops, uint32_t ops_cap, uint32_t *n_ops);
/* any Block<trace, xdrop>::align the crate's unit tests use; kind 0 AAMatrix [27 * 32] (bytes = letter - 'A'), 1 NucMatrix [8 * 16]
 * (upper-case ASCII), 2 ByteMatrix {match, mismatch} (raw bytes); ops: cigar.rs Operation codes (M 1, = 2, X 3, I 4, D 5), end -> origin */
int mmo_block_align_generic(const uint8_t *q, int qlen, const uint8_t *r, int rlen, int kind, const int8_t *table, int gap_open, int gap_extend,
                            int min_size, int max_size, int x_drop, int trace, int xdrop, int eq, mmo_block_res *res, uint8_t *ops,
                            uint32_t ops_cap, uint32_t *n_ops);
/* Block<trace, xdrop>::align_profile (scan_block.rs:919-944) of query bytes q (letter - 'A') against an AAProfile of plen positions:
 * pos_aa_rows [plen][32] int8, gap costs for the indices 0 .. gap_n - 1 (index 0 = the padding position in front), everything else
 * as AAProfile::new leaves it (i8::MIN); pad_block = the block_size the profile was created with */
int mmo_block_align_profile(const uint8_t *q, int qlen, int plen, int pad_block, const int8_t *pos_aa_rows, const int16_t *gap_open_C,
                            const int16_t *gap_close_C, const int16_t *gap_open_R, int gap_n, int gap_extend, int min_size, int max_size,
                            int x_drop, int trace, int xdrop, mmo_block_res *res, uint8_t *ops, uint32_t ops_cap, uint32_t *n_ops);
/* SmithWaterman::alignStartPosBacktraceBlock<PROFILE_SEQ> (StripedSmithWaterman.cpp:943-1127): prof = int8 [alphabet][qlen] score rows
 * of the profile query, q = its consensus sequence */
int mmo_sw_block_backtrace_profile(const int8_t *prof, const uint8_t *q, int qlen, const uint8_t *t, int tlen, int alphabet, int gap_open,
                                   int gap_extend, int score, int q_end, int t_end, int *q_start, int *t_start, uint32_t *ident, char *bt,
                                   int bt_cap, int *bt_len, int *block_size_used);
void mmo_block_prefix_scan(const int16_t *v16, int gap, int16_t *out16);
/* test aid: the block list (i, j, height << 16 | width, right per block) of the last alignment run on the calling thread */
void mmo_block_growth_capture(uint32_t *buf, uint32_t cap);
uint32_t mmo_block_growth_count(void);
int mmo_sw_block_backtrace(const uint8_t *q, const int8_t *comp_bias, int qlen, const uint8_t *t, int tlen, const int8_t *mat, int alphabet,
                           int gap_open /* > 0, as the reference's */, int gap_extend, int score, int q_end, int t_end, int *q_start, int *t_start,
                           uint32_t *ident, char *bt, int bt_cap, int *bt_len, int *block_size_used);

/* ---- tantan repeat masking (tantan_oracle.c): lib/tantan as the reference's AVX2 + FMA build computes it ---- */
void mmo_tantan_b2f(double repeatProb, double decay, int maxRepeatOffset, double *b2f);
void mmo_tantan_probs(const uint8_t *seq, int len, const double *lr, int alph, double repeatProb, double repeatEndProb, double decay,
                      int maxRepeatOffset, float *probs);
int mmo_tantan_mask(uint8_t *seq, int len, const double *lr, int alph, double min_mask_prob, uint8_t mask_letter, float *probs_out);

#ifdef __cplusplus
}
#endif

#endif
