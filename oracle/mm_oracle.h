/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 * Declarations of the plain-C oracle (oracle/ .c files).  See each .c file for the reference
 * file:line every function restates. */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H
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
int mmo_sw_score_identical(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *t,
                           const int8_t *mat, int alphabet);
/* batch driver used by tests and bench.py's cpu_baseline ("port" kind): one query, n targets from a
 * flat residue array with offsets, score/end only. */
void mmo_sw_batch_score(const uint8_t *q, int qlen, const int8_t *comp_bias, const uint8_t *tdata,
                        const uint64_t *toff, const uint32_t *ids, int n, const int8_t *mat, int alphabet,
                        int gap_open, int gap_extend, int32_t *score, int32_t *q_end, int32_t *t_end,
                        int32_t *word);

/* compbias_oracle.c */
void mmo_comp_bias(const int16_t *submat /*alphabet^2, row-major short matrix*/, const double *pback, int alphabet,
                   const uint8_t *seq, int n, float scale, float *out);

#ifdef __cplusplus
}
#endif
#endif
