/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the block aligner as the reference uses it for int16-range hits:
 *
 *   SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ>   src/alignment/StripedSmithWaterman.cpp:943-1127
 *     -> block_align_aa_trace_xdrop_posbias / block_res_aa_trace_xdrop / block_cigar_aa_trace_xdrop
 *        = Block<TRACE = true, X_DROP = true>::align_aa   lib/block-aligner/src/scan_block.rs:1016-1052 (crate 0.4.0, vendored
 *          and modified by the reference: positional bias, numeric residues, the arg-max tie rule of :374-444)
 *
 * The crate is Rust and cannot be built in this image (no rustc); what is restated is its AVX2 configuration, the one a
 * -DHAVE_AVX2 build of the reference links (lib/block-aligner/src/avx2.rs: L = 16 lanes of int16, ZERO = 1 << 14, MIN = 0).
 * The result depends on the lane structure (zero fill of the in-lane byte shifts in the prefix scan, the blend masks of the
 * arg-max, the 16-row trace words), so every vector operation below is the lane-by-lane meaning of the intrinsic the crate
 * calls, each citing its line.
 *
 * PARITY: unpinned against the Rust-linked binary in this image.  scripts/make_block_goldens.sh is the one-command recipe a
 * Rust-equipped box runs to record tests/golden/block_vectors.npz; tests/test_block_oracle.py compares with it when the file
 * exists and says so loudly when it does not.  What IS checked here: the invariants the reference itself relies on (it only
 * accepts a block alignment whose score equals the striped SW score, :1058; the CIGAR re-scores to that score; start
 * positions consistent with the CIGAR), over thousands of pairs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

#define BL 16                 /* avx2.rs:10  L */
#define B_ZERO 16384          /* avx2.rs:15  ZERO = 1 << 14 */
#define B_MIN 0               /* avx2.rs:16  MIN */
#define B_STEP 8              /* scan_block.rs:813 */
#define B_X_DROP_ITER 2       /* :814 */
#define B_SHRINK 1            /* :815 */
#define B_SHRINK_SUFFIX_LEN 2 /* :816  STEP / 4 */
#define B_NULL 26             /* scores.rs:88 AAMatrix::NULL - 'A' (convert_char, :142-146) */

typedef struct { int16_t v[BL]; } bvec;

static int16_t sat16(int x) { return (int16_t)(x > 32767 ? 32767 : (x < -32768 ? -32768 : x)); }
static bvec v_set1(int16_t x) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = x; return r; }
static bvec v_load(const int16_t *p) { bvec r; memcpy(r.v, p, sizeof(r.v)); return r; }
static void v_store(int16_t *p, bvec a) { memcpy(p, a.v, sizeof(a.v)); }
static bvec v_adds(bvec a, bvec b) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = sat16((int)a.v[k] + b.v[k]); return r; }   /* _mm256_adds_epi16 */
static bvec v_subs(bvec a, bvec b) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = sat16((int)a.v[k] - b.v[k]); return r; }   /* _mm256_subs_epi16 */
static bvec v_max(bvec a, bvec b) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = a.v[k] > b.v[k] ? a.v[k] : b.v[k]; return r; }
static bvec v_cmpeq(bvec a, bvec b) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = a.v[k] == b.v[k] ? (int16_t)-1 : 0; return r; }
/* _mm256_blendv_epi8(a, b, mask) with masks that are whole int16 lanes of 0 / -1 (arg-max update, :1583-1585) */
static bvec v_blend16(bvec a, bvec b, bvec mask) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = mask.v[k] ? b.v[k] : a.v[k]; return r; }
/* simd_sl_i16!(a, b, 1), avx2.rs:101-117: [b[15], a[0], .., a[14]] */
static bvec v_sl1(bvec a, bvec b) { bvec r; r.v[0] = b.v[BL - 1]; for (int k = 1; k < BL; k++) r.v[k] = a.v[k - 1]; return r; }
/* simd_step(a, b) = _mm256_permute2x128_si256(a, b, 0x03), avx2.rs:139-143: [b[8..15], a[0..7]] */
static bvec v_step(bvec a, bvec b) { bvec r; for (int k = 0; k < 8; k++) { r.v[k] = b.v[8 + k]; r.v[8 + k] = a.v[k]; } return r; }
/* simd_sllz_i16!(a, n) = _mm256_slli_si256(a, 2n), avx2.rs:152-164: byte shift INSIDE each 128-bit half, zeros shifted in */
static bvec v_sllz(bvec a, int n) {
    bvec r;
    for (int k = 0; k < BL; k++) r.v[k] = (k % 8) >= n ? a.v[k - n] : 0;
    return r;
}
/* _mm256_slli_epi16(gap, s): per-lane shift, not saturating (avx2.rs:322,325) */
static bvec v_slli16(bvec a, int s) { bvec r; for (int k = 0; k < BL; k++) r.v[k] = (int16_t)((uint16_t)a.v[k] << s); return r; }
static int16_t v_hmax(bvec a) { int16_t m = a.v[0]; for (int k = 1; k < BL; k++) if (a.v[k] > m) m = a.v[k]; return m; }   /* simd_hmax_i16, avx2.rs:188-195 (no zero reaches lane 0) */
static bvec v_broadcasthi(bvec a) { return v_set1(a.v[BL - 1]); }      /* avx2.rs:167-172 */

/* get_prefix_scan_consts, avx2.rs:294-309 -> (gap_extend_all, prefix_scan_consts) */
static void prefix_scan_consts(bvec gap, bvec *gap_all, bvec *consts) {
    bvec shift1 = v_adds(v_sllz(gap, 1), gap);
    bvec shift2 = v_adds(v_sllz(shift1, 2), shift1);
    bvec shift4 = v_adds(v_sllz(shift2, 4), shift2);
    /* correct1 = srli_si256(shufflehi(shift4, 0xFF), 8) -> per half [w7 w7 w7 w7 0 0 0 0]; permute4x64(.., 0b00000101) ->
     * quad words (q1, q1, q0, q0) of that = [0 x 8, low half's w7 x 4, low half's w7 x 4] */
    bvec correct1;
    for (int k = 0; k < 8; k++) correct1.v[k] = 0;
    for (int k = 8; k < BL; k++) correct1.v[k] = shift4.v[7];
    *gap_all = v_adds(correct1, shift4);
    *consts = shift4;
}

/* simd_prefix_scan_i16, avx2.rs:311-337 */
static bvec prefix_scan(bvec R_max, bvec gap_cost, bvec gap_cost_lane) {
    bvec shift1 = v_max(R_max, v_adds(v_sllz(R_max, 1), gap_cost));
    bvec shift2 = v_max(shift1, v_adds(v_sllz(shift1, 2), v_slli16(gap_cost, 1)));
    bvec shift4 = v_max(shift2, v_adds(v_sllz(shift2, 4), v_slli16(gap_cost, 2)));
    /* shufflehi(shift4, 0xFF): words 4-7 of each half = word 7 of the half; permute4x64(.., 0b01010000): (q0, q0, q1, q1) */
    bvec correct1;
    for (int k = 0; k < 4; k++) { correct1.v[k] = shift4.v[k]; correct1.v[4 + k] = shift4.v[k]; }
    for (int k = 8; k < BL; k++) correct1.v[k] = shift4.v[7];
    correct1 = v_adds(correct1, gap_cost_lane);
    return v_max(shift4, correct1);
}

/* test hook: the crate's own unit test of the scan (avx2.rs test_prefix_scan) runs against this */
void mmo_block_prefix_scan(const int16_t *v, int gap, int16_t *out) {
    bvec all, consts;
    prefix_scan_consts(v_set1((int16_t)gap), &all, &consts);
    v_store(out, prefix_scan(v_load(v), v_set1((int16_t)gap), consts));
}

/* simd_movemask_i8(simd_blend_i8(lo, hi, 0xFF00 per lane)): bit 2k = lo lane k, bit 2k + 1 = hi lane k (:1570-1574) */
static uint32_t trace_word(bvec lo, bvec hi) {
    uint32_t w = 0;
    for (int k = 0; k < BL; k++) w |= ((lo.v[k] ? 1u : 0u) | (hi.v[k] ? 2u : 0u)) << (2 * k);
    return w;
}

/* ---- data the FFI objects hold ---- */
typedef struct {
    const uint8_t *s;      /* PaddedBytes::s: [NULL, bytes.., NULL x block_size] (scan_block.rs:2174-2180) */
    const int16_t *bias;   /* PosBias::bias: [0, b.., 0 x block_size] (scores.rs:721-725) */
    int len;
} bseq;

typedef struct {
    uint32_t *trace, *trace2;      /* Trace::trace / trace2 (TraceType = i32, avx2.rs:9) */
    uint8_t *right;                /* one flag per block (Trace::right bit set) */
    uint32_t *block_start;         /* [2 * blocks] i, j */
    uint16_t *block_size;          /* [2 * blocks] height, width */
    size_t trace_idx, block_idx, ckpt_trace_idx, ckpt_block_idx;
    size_t trace_cap, block_cap;
} btrace;

static void trace_add_block(btrace *t, size_t i, size_t j, size_t width, size_t height, int right) {   /* :1790-1805 */
    if (t->block_idx >= t->block_cap) abort();
    t->block_start[t->block_idx * 2] = (uint32_t)i;
    t->block_start[t->block_idx * 2 + 1] = (uint32_t)j;
    t->block_size[t->block_idx * 2] = (uint16_t)height;
    t->block_size[t->block_idx * 2 + 1] = (uint16_t)width;
    t->right[t->block_idx] = (uint8_t)right;
    t->block_idx++;
}

enum { BK_AA = 0, BK_NUC = 1, BK_BYTES = 2 };
/* AAProfile (scores.rs:475-489): position-specific scores both ways round and position-specific gap costs; index 0 is the
 * padding position in front (set(i + 1, ..) stores position i), curr_len = str_len + block_size + 1 entries per array, every
 * entry that was never set is i8::MIN (:494-506) */
typedef struct {
    const int8_t *pos_aa;                                   /* [curr_len][32] */
    const int16_t *aa_pos;                                  /* [32][curr_len] = the transpose */
    const int16_t *gap_open_C, *gap_close_C, *gap_open_R;   /* [curr_len] */
    int gap_extend;
    int len;                                                /* str_len */
    size_t curr_len;
    /* The vendored crate carries two edits by the reference's authors (scan_block.rs:724-725 and :731-732: the upstream lines are
     * left as comments above them): opening a gap costs gap_open in the tree, gap_open + gap_extend upstream.  The crate's
     * test_profile assertions were NOT updated with the edit - they hold for the upstream form only.  upstream_gaps = 1 computes
     * that form (to replay the crate's vectors); the reference links the edited form (0). */
    int upstream_gaps;
} bprofile;
typedef struct {
    const int8_t *scores;   /* AAMatrix::scores [27 * 32] (scores.rs:47-49) | NucMatrix::scores [8 * 16] (:157) | ByteMatrix {match, mismatch} (:238-241) */
    int gap_open, gap_extend;
    int kind;               /* which Matrix impl `scores` belongs to */
    int trace, xdrop;       /* Block<TRACE, X_DROP> (scan_block.rs:115); the reference instantiates <true, true> only */
    const bprofile *prof;   /* align_profile (scan_block.rs:919-944): the "reference" is this profile, `scores` / gap_open are unused */
} bparams;

/* Matrix::get_scores for one lane: the score of reference byte c against query byte v */
static int16_t lookup_score(const bparams *P, uint8_t c, uint8_t v) {
    if (P->kind == BK_NUC)      /* scores.rs:216-221 -> halfsimd_lookup1_i16 (avx2.rs:352-356): pshufb, row c & 7, entry v & 15, 0 if bit 7 of v */
        return (v & 0x80) ? 0 : P->scores[(size_t)(c & 7) * 16 + (v & 15)];
    if (P->kind == BK_BYTES)    /* scores.rs:276-280 -> halfsimd_lookup_bytes_i16 (avx2.rs:358-364) */
        return c == v ? P->scores[0] : P->scores[1];
    /* AAMatrix::get_scores (scores.rs:133-139) -> halfsimd_lookup2_i16 (avx2.rs:340-350): row c of the matrix, entry = low 5 bits
     * of the query byte (pshufb on the low 4 bits, blendv on bit 4), sign extended */
    return (v & 0x80) ? 0 : P->scores[(size_t)c * 32 + (v & 31)];
}

/* place_block_aa, scan_block.rs:1449-1613; with zero biases it is place_block (:1145-1277), whose only other difference is the
 * Matrix behind get_scores.  `query` is what the rows run over, `reference` what the columns run over; a down shift calls it with
 * the roles exchanged (:237-252). */
static void place_block_mat(const bparams *P, bseq query, bseq reference, btrace *tr, size_t start_i, size_t start_j, size_t width,
                            size_t height, int16_t *D_col, int16_t *C_col, int16_t *D_row, int16_t *R_row, bvec D_corner,
                            bvec *oD_max, bvec *oD_argmax_i, bvec *oD_argmax_j) {
    const bvec gap_open = v_set1((int16_t)P->gap_open), gap_extend = v_set1((int16_t)P->gap_extend);
    bvec gap_extend_all, consts;
    prefix_scan_consts(gap_extend, &gap_extend_all, &consts);
    bvec D_max = v_set1(B_MIN), D_argmax_i = v_set1(0), D_argmax_j = v_set1(0);
    if (width == 0 || height == 0) { *oD_max = D_max; *oD_argmax_i = D_argmax_i; *oD_argmax_j = D_argmax_j; return; }
    for (size_t j = 0; j < width; j++) {
        bvec R01 = v_set1(B_MIN), D11 = v_set1(B_MIN), R11 = v_set1(B_MIN), prev_trace_R = v_set1(0);
        const uint8_t c = reference.s[start_j + j];
        const bvec reference_bias = v_set1(reference.bias[start_j + j]);
        for (size_t i = 0; i < height; i += BL) {
            const bvec D10 = v_load(D_col + i), C10 = v_load(C_col + i);
            const bvec D00 = v_sl1(D10, D_corner);
            D_corner = D10;
            bvec scores, query_bias;
            for (int k = 0; k < BL; k++) {
                scores.v[k] = lookup_score(P, c, query.s[start_i + i + k]);
                query_bias.v[k] = query.bias[start_i + i + k];      /* PosBias::get_biases, scores.rs:737-739 */
            }
            const bvec pos_bias = v_adds(reference_bias, query_bias);
            D11 = v_adds(D00, v_adds(scores, pos_bias));
            if (start_i + i == 0 && start_j + j == 0) D11.v[0] = B_ZERO;       /* :1498-1500 */
            const bvec C11_open = v_adds(D10, gap_open);
            const bvec C11 = v_max(v_adds(C10, gap_extend), C11_open);
            D11 = v_max(D11, C11);
            const bvec D11_open = v_adds(D11, v_subs(gap_open, gap_extend));
            R11 = prefix_scan(D11_open, gap_extend, consts);
            R11 = v_max(R11, v_adds(v_broadcasthi(R01), gap_extend_all));
            D11 = v_max(D11, R11);
            R01 = R11;
            if (P->trace) {   /* TRACE, :1559-1577 */
                const bvec trace_D_C = v_cmpeq(D11, C11), trace_D_R = v_cmpeq(D11, R11);
                const uint32_t trace_data = trace_word(trace_D_C, trace_D_R);
                const bvec temp_trace_R = v_cmpeq(R11, D11_open);
                const bvec trace_R = v_sl1(temp_trace_R, prev_trace_R);
                const uint32_t trace_data2 = trace_word(v_cmpeq(C11, C11_open), trace_R);
                prev_trace_R = temp_trace_R;
                if (tr->trace_idx >= tr->trace_cap) abort();
                tr->trace[tr->trace_idx] = trace_data;
                tr->trace2[tr->trace_idx] = trace_data2;
                tr->trace_idx++;
            }
            D_max = v_max(D_max, D11);
            if (P->xdrop) {   /* X_DROP, :1581-1586 */
                const bvec mask = v_cmpeq(D_max, D11);
                D_argmax_i = v_blend16(D_argmax_i, v_set1((int16_t)i), mask);
                D_argmax_j = v_blend16(D_argmax_j, v_set1((int16_t)j), mask);
            }
            v_store(D_col + i, D11);
            v_store(C_col + i, C11);
        }
        D_corner = v_set1(B_MIN);
        D_row[j] = D11.v[BL - 1];
        R_row[j] = R11.v[BL - 1];
        if (!P->xdrop && start_i + height > (size_t)query.len && start_j + j >= (size_t)reference.len) {   /* :1601-1609 */
            if (P->trace) tr->trace_idx += (width - 1 - j) * (height / BL);
            break;
        }
    }
    *oD_max = D_max;
    *oD_argmax_i = D_argmax_i;
    *oD_argmax_j = D_argmax_j;
}

/* place_block_profile_gen!, scan_block.rs:649-815, instantiated twice (:1279-1280): `right` = rows run over the sequence `seq`,
 * columns over profile positions (place_block_profile_right); !right = rows run over profile positions - 16 consecutive ones per
 * vector - and columns over the letters of `seq` (place_block_profile_down, called with the roles and start_i / start_j
 * exchanged, :237-252).  The gap costs are the profile's: per column when shifting right, per vector lane when shifting down, with
 * the C and R roles exchanged there (:709-714); a closing cost exists for the C gap only (right: added to C, down: added to R). */
static void place_block_profile(const bparams *P, int right, bseq seq, btrace *tr, size_t start_i, size_t start_j, size_t width,
                                size_t height, int16_t *D_col, int16_t *C_col, int16_t *D_row, int16_t *R_row, bvec D_corner,
                                bvec *oD_max, bvec *oD_argmax_i, bvec *oD_argmax_j) {
    const bprofile *pr = P->prof;
    const bvec gap_extend = v_set1((int16_t)pr->gap_extend);
    bvec gap_extend_all, consts;
    prefix_scan_consts(gap_extend, &gap_extend_all, &consts);
    bvec D_max = v_set1(B_MIN), D_argmax_i = v_set1(0), D_argmax_j = v_set1(0);
    bvec gap_open_C = v_set1(B_MIN), gap_close_C = v_set1(B_MIN), gap_open_R = v_set1(B_MIN), gap_close_R = v_set1(B_MIN);
    if (width == 0 || height == 0) { *oD_max = D_max; *oD_argmax_i = D_argmax_i; *oD_argmax_j = D_argmax_j; return; }
    /* `$query.len()` / `$reference.len()` of the macro are the lengths of what the rows / the columns run over */
    const size_t rows_len = right ? (size_t)seq.len : (size_t)pr->len, cols_len = right ? (size_t)pr->len : (size_t)seq.len;
    for (size_t j = 0; j < width; j++) {
        bvec R01 = v_set1(B_MIN), D11 = v_set1(B_MIN), R11 = v_set1(B_MIN), prev_trace_R = v_set1(0);
        size_t idx = 0;
        if (right) {                                                    /* :699-704 */
            idx = start_j + j;
            gap_open_C = v_set1(pr->gap_open_C[idx]);
            gap_close_C = v_set1(pr->gap_close_C[idx]);
            gap_open_R = v_set1(pr->gap_open_R[idx]);
        }
        for (size_t i = 0; i < height; i += BL) {
            const bvec D10 = v_load(D_col + i), C10 = v_load(C_col + i);
            const bvec D00 = v_sl1(D10, D_corner);
            D_corner = D10;
            bvec scores;
            if (!right) {                                               /* :709-714 */
                idx = start_i + i;
                gap_open_C = v_load(pr->gap_open_R + idx);
                gap_open_R = v_load(pr->gap_open_C + idx);
                gap_close_R = v_load(pr->gap_close_C + idx);
                const uint8_t c = seq.s[start_j + j];                   /* get_scores_aa, scores.rs:634-637 */
                for (int k = 0; k < BL; k++) scores.v[k] = pr->aa_pos[(size_t)c * pr->curr_len + idx + k];
            } else {                                                    /* get_scores_pos, scores.rs:621-627 -> halfsimd_lookup2_i16 */
                for (int k = 0; k < BL; k++) {
                    const uint8_t v = seq.s[start_i + i + k];
                    scores.v[k] = (v & 0x80) ? 0 : pr->pos_aa[idx * 32 + (v & 31)];
                }
            }
            D11 = v_adds(D00, scores);
            if (start_i + i == 0 && start_j + j == 0) D11.v[0] = B_ZERO;       /* :722-724 */
            const bvec C11_open = pr->upstream_gaps ? v_adds(D10, v_adds(gap_open_C, gap_extend)) : v_adds(D10, gap_open_C);
            const bvec C11 = v_max(v_adds(C10, gap_extend), C11_open);
            const bvec C11_end = right ? v_adds(C11, gap_close_C) : C11;
            D11 = v_max(D11, C11_end);
            const bvec D11_open = pr->upstream_gaps ? v_adds(D11, gap_open_R) : v_adds(D11, v_subs(gap_open_R, gap_extend));
            R11 = prefix_scan(D11_open, gap_extend, consts);
            R11 = v_max(R11, v_adds(v_broadcasthi(R01), gap_extend_all));
            const bvec R11_end = right ? R11 : v_adds(R11, gap_close_R);
            D11 = v_max(D11, R11_end);
            R01 = R11;
            if (P->trace) {   /* :758-776 */
                const bvec trace_D_C = v_cmpeq(D11, C11_end), trace_D_R = v_cmpeq(D11, R11_end);
                const uint32_t trace_data = trace_word(trace_D_C, trace_D_R);
                const bvec temp_trace_R = v_cmpeq(R11, D11_open);
                const bvec trace_R = v_sl1(temp_trace_R, prev_trace_R);
                const uint32_t trace_data2 = trace_word(v_cmpeq(C11, C11_open), trace_R);
                prev_trace_R = temp_trace_R;
                if (tr->trace_idx >= tr->trace_cap) abort();
                tr->trace[tr->trace_idx] = trace_data;
                tr->trace2[tr->trace_idx] = trace_data2;
                tr->trace_idx++;
            }
            D_max = v_max(D_max, D11);
            if (P->xdrop) {   /* :780-785 */
                const bvec mask = v_cmpeq(D_max, D11);
                D_argmax_i = v_blend16(D_argmax_i, v_set1((int16_t)i), mask);
                D_argmax_j = v_blend16(D_argmax_j, v_set1((int16_t)j), mask);
            }
            v_store(D_col + i, D11);
            v_store(C_col + i, C11);
        }
        D_corner = v_set1(B_MIN);
        D_row[j] = D11.v[BL - 1];
        R_row[j] = R11.v[BL - 1];
        if (!P->xdrop && start_i + height > rows_len && start_j + j >= cols_len) {   /* :797-805 */
            if (P->trace) tr->trace_idx += (width - 1 - j) * (height / BL);
            break;
        }
    }
    *oD_max = D_max;
    *oD_argmax_i = D_argmax_i;
    *oD_argmax_j = D_argmax_j;
}

/* what align_core_gen! is instantiated with (:1054-1057): the matrix forms take (rows, columns) as two sequences; the profile
 * forms take the sequence and read the profile from the parameters.  `right`: the caller's first sequence runs down the rows. */
static void place_block_aa(const bparams *P, int right, bseq rows, bseq cols, btrace *tr, size_t start_i, size_t start_j, size_t width,
                           size_t height, int16_t *D_col, int16_t *C_col, int16_t *D_row, int16_t *R_row, bvec D_corner,
                           bvec *oD_max, bvec *oD_argmax_i, bvec *oD_argmax_j) {
    if (P->prof) place_block_profile(P, right, right ? rows : cols, tr, start_i, start_j, width, height, D_col, C_col, D_row, R_row, D_corner, oD_max, oD_argmax_i, oD_argmax_j);
    else place_block_mat(P, rows, cols, tr, start_i, start_j, width, height, D_col, C_col, D_row, R_row, D_corner, oD_max, oD_argmax_i, oD_argmax_j);
}

static void just_offset(size_t block_size, int16_t *buf1, int16_t *buf2, bvec off_add) {   /* :1065-1074 */
    for (size_t i = 0; i < block_size; i += BL) {
        v_store(buf1 + i, v_adds(v_load(buf1 + i), off_add));
        v_store(buf2 + i, v_adds(v_load(buf2 + i), off_add));
    }
}
static int16_t prefix_max(const int16_t *buf) {   /* :1082-1084, simd_prefix_hmax_i16!(v, STEP = 8): max of the first 8 lanes */
    int16_t m = buf[0];
    for (int k = 1; k < B_STEP; k++) if (buf[k] > m) m = buf[k];
    return m;
}
static int16_t suffix_max(const int16_t *buf, size_t len) {   /* :1092-1094, simd_suffix_hmax_i16!(v, 2): max of the last 2 lanes */
    return buf[len - 1] > buf[len - 2] ? buf[len - 1] : buf[len - 2];
}
static bvec shift_and_offset(size_t block_size, int16_t *buf1, int16_t *buf2, const int16_t *temp1, const int16_t *temp2, bvec off_add) {   /* :1102-1123 */
    bvec curr1 = v_adds(v_load(buf1), off_add);
    const bvec D_corner = v_set1(curr1.v[B_STEP - 1]);
    bvec curr2 = v_adds(v_load(buf2), off_add);
    size_t i = 0;
    for (; i < block_size - BL; i += BL) {
        const bvec next1 = v_adds(v_load(buf1 + i + BL), off_add), next2 = v_adds(v_load(buf2 + i + BL), off_add);
        v_store(buf1 + i, v_step(next1, curr1));
        v_store(buf2 + i, v_step(next2, curr2));
        curr1 = next1;
        curr2 = next2;
    }
    v_store(buf1 + block_size - BL, v_step(v_load(temp1), curr1));
    v_store(buf2 + block_size - BL, v_step(v_load(temp2), curr2));
    return D_corner;
}

static int16_t clamp16(int32_t x) { return (int16_t)(x < -32768 ? -32768 : (x > 32767 ? 32767 : x)); }   /* :2038-2040 */

enum { DIR_RIGHT, DIR_DOWN, DIR_GROW };

typedef struct {
    int16_t *D_col, *C_col, *D_row, *R_row, *D_col_ckpt, *C_col_ckpt, *D_row_ckpt, *R_row_ckpt, *temp1, *temp2;
} balloc;

/* align_core_gen!(align_aa_core, ..), scan_block.rs:120-632 with TRACE = X_DROP = true */
static void align_aa_core(const bparams *P, bseq query, bseq reference, size_t min_size, size_t max_size, int32_t x_drop, balloc *A,
                          btrace *tr, int32_t *res_score, size_t *res_i, size_t *res_j) {
    int32_t best_max = 0;
    size_t best_argmax_i = 0, best_argmax_j = 0;
    int prev_dir = DIR_GROW, dir = DIR_GROW;
    size_t prev_size = 0, block_size = min_size;
    int32_t off = 0, prev_off, off_max = 0;
    size_t y_drop_iter = 0, x_drop_iter = 0;
    size_t st_i = 0, st_j = 0, i_ckpt = 0, j_ckpt = 0;
    int32_t off_ckpt = 0;
    bvec D_corner = v_set1(B_MIN);
    const size_t qlen = (size_t)query.len, rlen = (size_t)reference.len;
    for (;;) {
        prev_off = off;
        bvec grow_D_max = v_set1(B_MIN), grow_D_argmax_i = v_set1(0), grow_D_argmax_j = v_set1(0);
        bvec D_max, D_argmax_i, D_argmax_j;
        int16_t right_max, down_max;
        if (dir == DIR_RIGHT) {
            off = off_max;
            const bvec off_add = v_set1(clamp16(prev_off - off));
            if (P->trace) trace_add_block(tr, st_i, st_j + block_size - B_STEP, B_STEP, block_size, 1);
            just_offset(block_size, A->D_col, A->C_col, off_add);
            place_block_aa(P, 1, query, reference, tr, st_i, st_j + block_size - B_STEP, B_STEP, block_size, A->D_col, A->C_col, A->temp1, A->temp2,
                           prev_dir == DIR_DOWN ? v_adds(D_corner, off_add) : v_set1(B_MIN), &D_max, &D_argmax_i, &D_argmax_j);
            right_max = prefix_max(A->D_col);
            D_corner = shift_and_offset(block_size, A->D_row, A->R_row, A->temp1, A->temp2, off_add);
            down_max = prefix_max(A->D_row);
        } else if (dir == DIR_DOWN) {
            off = off_max;
            const bvec off_add = v_set1(clamp16(prev_off - off));
            if (P->trace) trace_add_block(tr, st_i + block_size - B_STEP, st_j, block_size, B_STEP, 0);
            just_offset(block_size, A->D_row, A->R_row, off_add);
            place_block_aa(P, 0, reference, query, tr, st_j, st_i + block_size - B_STEP, B_STEP, block_size, A->D_row, A->R_row, A->temp1, A->temp2,
                           prev_dir == DIR_RIGHT ? v_adds(D_corner, off_add) : v_set1(B_MIN), &D_max, &D_argmax_i, &D_argmax_j);
            down_max = prefix_max(A->D_row);
            D_corner = shift_and_offset(block_size, A->D_col, A->C_col, A->temp1, A->temp2, off_add);
            right_max = prefix_max(A->D_col);
        } else {
            D_corner = v_set1(B_MIN);
            const size_t grow_step = block_size - prev_size;
            if (P->trace) trace_add_block(tr, st_i + prev_size, st_j, prev_size, grow_step, 0);
            bvec D_max1, D_ai1, D_aj1;
            place_block_aa(P, 0, reference, query, tr, st_j, st_i + prev_size, grow_step, prev_size, A->D_row, A->R_row, A->D_col + prev_size,
                           A->C_col + prev_size, v_set1(B_MIN), &D_max1, &D_ai1, &D_aj1);
            if (P->trace) trace_add_block(tr, st_i, st_j + prev_size, grow_step, block_size, 1);
            place_block_aa(P, 1, query, reference, tr, st_i, st_j + prev_size, grow_step, block_size, A->D_col, A->C_col, A->D_row + prev_size,
                           A->R_row + prev_size, v_set1(B_MIN), &D_max, &D_argmax_i, &D_argmax_j);
            right_max = prefix_max(A->D_col);
            down_max = prefix_max(A->D_row);
            grow_D_max = D_max1;
            grow_D_argmax_i = D_ai1;
            grow_D_argmax_j = D_aj1;
            memcpy(A->D_col_ckpt, A->D_col, block_size * 2);      /* :337-344 */
            memcpy(A->C_col_ckpt, A->C_col, block_size * 2);
            memcpy(A->D_row_ckpt, A->D_row, block_size * 2);
            memcpy(A->R_row_ckpt, A->R_row, block_size * 2);
            tr->ckpt_trace_idx = tr->trace_idx;
            tr->ckpt_block_idx = tr->block_idx;
        }
        prev_dir = dir;
        const int16_t D_max_max = v_hmax(D_max), grow_max = v_hmax(grow_D_max);
        const int16_t max = D_max_max > grow_max ? D_max_max : grow_max;
        off_max = off + (int32_t)max - (int32_t)B_ZERO;
        y_drop_iter++;
        int grow_no_max = dir == DIR_GROW;
        if (off_max > best_max) {
            if (P->xdrop) {   /* X_DROP: location of the maximum, ties to the larger column, then the larger row (:374-444) */
                size_t best_i = 0, best_j = 0;
                const int grow = dir == DIR_GROW && D_max_max < grow_max;
                const int16_t curr_max = grow ? grow_max : D_max_max;
                const bvec cm = grow ? grow_D_max : D_max, ci = grow ? grow_D_argmax_i : D_argmax_i, cj = grow ? grow_D_argmax_j : D_argmax_j;
                for (int lane = 0; lane < BL; lane++) {
                    if (cm.v[lane] != curr_max) continue;
                    const size_t idx_i = (size_t)(int64_t)ci.v[lane], idx_j = (size_t)(int64_t)cj.v[lane];      /* `as usize` */
                    const size_t r = idx_i + (size_t)lane, c = (block_size - B_STEP) + idx_j;
                    size_t gi, gj;
                    if (grow) { gi = st_i + prev_size + idx_j; gj = st_j + idx_i + (size_t)lane; }
                    else if (dir == DIR_RIGHT) { gi = st_i + r; gj = st_j + c; }
                    else if (dir == DIR_DOWN) { gi = st_i + c; gj = st_j + r; }
                    else { gi = st_i + idx_i + (size_t)lane; gj = st_j + prev_size + idx_j; }
                    if (gj != best_j ? gj > best_j : gi > best_i) { best_i = gi; best_j = gj; }
                }
                best_argmax_i = best_i;
                best_argmax_j = best_j;
            }
            if (block_size < max_size) {
                i_ckpt = st_i;
                j_ckpt = st_j;
                off_ckpt = off;
                memcpy(A->D_col_ckpt, A->D_col, block_size * 2);
                memcpy(A->C_col_ckpt, A->C_col, block_size * 2);
                memcpy(A->D_row_ckpt, A->D_row, block_size * 2);
                memcpy(A->R_row_ckpt, A->R_row, block_size * 2);
                tr->ckpt_trace_idx = tr->trace_idx;
                tr->ckpt_block_idx = tr->block_idx;
                grow_no_max = 0;
            }
            best_max = off_max;
            y_drop_iter = 0;
        }
        if (P->xdrop) {
            if (off_max < best_max - x_drop) {      /* :477-488 */
                if (x_drop_iter < B_X_DROP_ITER - 1) x_drop_iter++;
                else break;
            } else {
                x_drop_iter = 0;
            }
        }
        if (st_i + block_size > qlen && st_j + block_size > rlen) break;
        if (st_j + block_size > rlen) { st_i += B_STEP; dir = DIR_DOWN; continue; }
        if (st_i + block_size > qlen) { st_j += B_STEP; dir = DIR_RIGHT; continue; }
        const size_t next_size = block_size * 2;
        if (next_size <= max_size) {
            if (y_drop_iter > (block_size / B_STEP) - 1 || grow_no_max) {
                prev_size = block_size;
                block_size = next_size;
                dir = DIR_GROW;
                st_i = i_ckpt;
                st_j = j_ckpt;
                off = off_ckpt;
                memcpy(A->D_col, A->D_col_ckpt, prev_size * 2);
                memcpy(A->C_col, A->C_col_ckpt, prev_size * 2);
                memcpy(A->D_row, A->D_row_ckpt, prev_size * 2);
                memcpy(A->R_row, A->R_row_ckpt, prev_size * 2);
                tr->trace_idx = tr->ckpt_trace_idx;
                tr->block_idx = tr->ckpt_block_idx;
                y_drop_iter = 0;
                continue;
            }
        }
        if (B_SHRINK && block_size > min_size && y_drop_iter == 0) {
            const int16_t s1 = suffix_max(A->D_row, block_size), s2 = suffix_max(A->D_col, block_size);
            const int16_t shrink_max = s1 > s2 ? s1 : s2;
            if (shrink_max >= max) {
                prev_dir = DIR_GROW;
                block_size /= 2;
                memmove(A->D_col, A->D_col + block_size, block_size * 2);     /* copy_vec(i, i + block_size), ascending i (:552-559) */
                memmove(A->C_col, A->C_col + block_size, block_size * 2);
                memmove(A->D_row, A->D_row + block_size, block_size * 2);
                memmove(A->R_row, A->R_row + block_size, block_size * 2);
                st_i += block_size;
                st_j += block_size;
                i_ckpt = st_i;
                j_ckpt = st_j;
                off_ckpt = off;
                memcpy(A->D_col_ckpt, A->D_col, block_size * 2);
                memcpy(A->C_col_ckpt, A->C_col, block_size * 2);
                memcpy(A->D_row_ckpt, A->D_row, block_size * 2);
                memcpy(A->R_row_ckpt, A->R_row, block_size * 2);
                right_max = prefix_max(A->D_col);
                down_max = prefix_max(A->D_row);
                tr->ckpt_trace_idx = tr->trace_idx;
                tr->ckpt_block_idx = tr->block_idx;
                y_drop_iter = 0;
            }
        }
        if (down_max > right_max) { st_i += B_STEP; dir = DIR_DOWN; }
        else { st_j += B_STEP; dir = DIR_RIGHT; }
    }
    if (P->xdrop) {      /* :604-631 */
        *res_score = best_max;
        *res_i = best_argmax_i;
        *res_j = best_argmax_j;
    } else {             /* global alignment: the cell (|q|, |r|), read from the border the last shift left it on */
        if (dir == DIR_DOWN) *res_score = off + (int32_t)A->D_row[rlen - st_j] - (int32_t)B_ZERO;
        else *res_score = off + (int32_t)A->D_col[qlen - st_i] - (int32_t)B_ZERO;
        *res_i = qlen;
        *res_j = rlen;
    }
}

/* Trace::cigar_core<false>, scan_block.rs:1844-2006; ops (cigar.rs:10-31: M = 1, I = 4, D = 5) are appended to `ops` in the
 * order the walk produces them, i.e. from the end position back to the origin.  Returns their number. */
static size_t trace_cigar(const btrace *t, size_t i, size_t j, uint8_t *ops, size_t cap, const uint8_t *eq_q, const uint8_t *eq_r) {
    enum { T_D = 0, T_C = 1, T_R = 2 };
    size_t block_idx = t->block_idx, trace_idx = t->trace_idx, n = 0;
    int table = T_D;
    while (i > 0 || j > 0) {
        size_t block_i, block_j, block_h, block_w;
        int right;
        for (;;) {
            if (block_idx == 0) abort();
            block_idx--;
            block_i = t->block_start[block_idx * 2];
            block_j = t->block_start[block_idx * 2 + 1];
            block_h = t->block_size[block_idx * 2];
            block_w = t->block_size[block_idx * 2 + 1];
            trace_idx -= block_w * block_h / BL;
            if (i >= block_i && j >= block_j) { right = t->right[block_idx]; break; }
        }
        while (i >= block_i && j >= block_j && (i > 0 || j > 0)) {
            const size_t ci = i - block_i, cj = j - block_j;
            size_t idx;
            unsigned sh;
            if (right) { idx = trace_idx + ci / BL + cj * (block_h / BL); sh = (unsigned)(ci % BL) * 2; }
            else { idx = trace_idx + cj / BL + ci * (block_w / BL); sh = (unsigned)(cj % BL) * 2; }
            const unsigned tt = (t->trace[idx] >> sh) & 3u, t2 = (t->trace2[idx] >> sh) & 3u;
            int op, di, dj, nt;
            /* OP_LUT, :1870-1933 */
            if (right) {
                if (table == T_C) { op = 5; di = 0; dj = 1; nt = (t2 & 1u) ? T_D : T_C; }
                else if (table == T_R) { op = 4; di = 1; dj = 0; nt = (t2 & 2u) ? T_D : T_R; }
                else if (tt == 0) { op = 1; di = 1; dj = 1; nt = T_D; }
                else if (tt & 1u) { op = 5; di = 0; dj = 1; nt = (t2 & 1u) ? T_D : T_C; }
                else { op = 4; di = 1; dj = 0; nt = (t2 & 2u) ? T_D : T_R; }
            } else {
                if (table == T_R) { op = 4; di = 1; dj = 0; nt = (t2 & 1u) ? T_D : T_R; }
                else if (table == T_C) { op = 5; di = 0; dj = 1; nt = (t2 & 2u) ? T_D : T_C; }
                else if (tt == 0) { op = 1; di = 1; dj = 1; nt = T_D; }
                else if (tt & 1u) { op = 4; di = 1; dj = 0; nt = (t2 & 1u) ? T_D : T_R; }
                else { op = 5; di = 0; dj = 1; nt = (t2 & 2u) ? T_D : T_C; }
            }
            if ((size_t)di > i || (size_t)dj > j) abort();      /* would be an out-of-range walk in the crate as well */
            if (eq_q && op == 1) op = eq_q[i] == eq_r[j] ? 2 : 3;      /* cigar_eq: Operation::Eq / X (:1957-1965), PaddedBytes::get(i) = s[i] */
            i -= (size_t)di;
            j -= (size_t)dj;
            table = nt;
            if (n >= cap) abort();
            ops[n++] = (uint8_t)op;
        }
    }
    return n;
}

/* One block_align_aa_trace_xdrop_posbias call + block_res + (optionally) block_cigar for the end position it reports.
 * q / r: numeric residues (< 26), qbias / rbias: int16 per position (NULL = zeros); mat: alphabet x alphabet, installed over
 * AAMatrix::new_simple(1, -1) exactly as ssw_init does (:708, :1469-1474).  ops (may be NULL): cap >= qlen + rlen + 5. */
int mmo_block_align(const uint8_t *q, const int16_t *qbias, int qlen, const uint8_t *r, const int16_t *rbias, int rlen, const int8_t *mat,
                    int alphabet, int gap_open, int gap_extend, int min_size, int max_size, int x_drop, mmo_block_res *res, uint8_t *ops,
                    uint32_t ops_cap, uint32_t *n_ops) {
    if (alphabet < 1 || alphabet > 26) return -1;
    int8_t scores[27 * 32];
    memset(scores, -128, 27 * 32);                                   /* AAMatrix::new_simple, scores.rs:53-66 */
    for (int a = 0; a < 26; a++)
        for (int b = 0; b < 26; b++) scores[a * 32 + b] = a == b ? 1 : -1;
    for (int a = 0; a < alphabet; a++)
        for (int b = 0; b < alphabet; b++) { scores[a * 32 + b] = mat[a * alphabet + b]; scores[b * 32 + a] = mat[a * alphabet + b]; }   /* set_num, :105-110, in ssw_init's loop order */
    return mmo_block_align_table(q, qbias, qlen, r, rbias, rlen, scores, gap_open, gap_extend, min_size, max_size, x_drop, res, ops, ops_cap, n_ops);
}

/* Test aid: the block list of the LAST alignment run on this thread (Trace::block_start / block_size / right as they stand when
 * align_core returns - the sequence of grow / right / down steps with their sizes, after every x-drop restore), four words per
 * block: i, j, height << 16 | width, right.  mmo_block_growth_capture(buf, cap) arms it, mmo_block_growth_count() = blocks of
 * the last run (may exceed cap: the list is cut, not the count). */
static __thread uint32_t *growth_buf = NULL;
static __thread uint32_t growth_cap = 0, growth_n = 0;
void mmo_block_growth_capture(uint32_t *buf, uint32_t cap) { growth_buf = buf; growth_cap = buf ? cap : 0; growth_n = 0; }
uint32_t mmo_block_growth_count(void) { return growth_n; }

/* Block<TRACE, X_DROP>::align / align_aa (scan_block.rs:862-892, :1016-1052) on PaddedBytes built from q / r (bytes AFTER
 * Matrix::convert_char), padded with `null_byte` (Matrix::NULL after convert_char).  eq: walk with cigar_eq instead of cigar. */
static int block_run(const uint8_t *q, const int16_t *qbias, int qlen, const uint8_t *r, const int16_t *rbias, int rlen, const bparams *P,
                     uint8_t null_byte, int min_size, int max_size, int x_drop, mmo_block_res *res, uint8_t *ops, uint32_t ops_cap,
                     uint32_t *n_ops, int eq) {
    if (P->prof) {
        if (qlen < 0 || rlen < 0 || P->prof->gap_extend >= 0) return -1;                                                  /* :921 */
    } else if (qlen < 0 || rlen < 0 || P->gap_open >= 0 || P->gap_extend >= 0 || P->gap_open >= P->gap_extend) return -1;      /* :864-867 */
    size_t mn = (size_t)(min_size < BL ? BL : min_size), mx = (size_t)(max_size < BL ? BL : max_size);      /* :868-869, :1024-1025 */
    if ((mn & (mn - 1)) || (mx & (mx - 1)) || mn > mx || mx >= 65535) return -1;
    uint8_t *qs = (uint8_t *)malloc((size_t)qlen + mx + 1 + BL), *rs = (uint8_t *)malloc((size_t)rlen + mx + 1 + BL);
    int16_t *qb = (int16_t *)calloc((size_t)qlen + mx + 1 + BL, 2), *rb = (int16_t *)calloc((size_t)rlen + mx + 1 + BL, 2);
    memset(qs, null_byte, (size_t)qlen + mx + 1 + BL);
    memset(rs, null_byte, (size_t)rlen + mx + 1 + BL);
    if (qlen) memcpy(qs + 1, q, (size_t)qlen);
    if (rlen && r) memcpy(rs + 1, r, (size_t)rlen);      /* (profile alignments have no reference bytes: place_block_profile never reads them) */
    for (int k = 0; k < qlen && qbias; k++) qb[1 + k] = qbias[k];
    for (int k = 0; k < rlen && rbias; k++) rb[1 + k] = rbias[k];
    bseq Q = {qs, qb, qlen}, R = {rs, rb, rlen};
    const size_t len = (size_t)qlen + (size_t)rlen;
    btrace T;
    T.trace_cap = (mx / BL) * (len + mx * 2);          /* Trace::new, :1742-1748 */
    T.block_cap = len + 8;                              /* block_start has 2 * len entries: len blocks */
    T.trace = (uint32_t *)malloc(T.trace_cap * 4 + 4);
    T.trace2 = (uint32_t *)malloc(T.trace_cap * 4 + 4);
    T.right = (uint8_t *)calloc(T.block_cap, 1);
    T.block_start = (uint32_t *)calloc(T.block_cap * 2, 4);
    T.block_size = (uint16_t *)calloc(T.block_cap * 2, 2);
    T.trace_idx = T.block_idx = T.ckpt_trace_idx = T.ckpt_block_idx = 0;
    balloc A;
    int16_t **bufs[10] = {&A.D_col, &A.C_col, &A.D_row, &A.R_row, &A.D_col_ckpt, &A.C_col_ckpt, &A.D_row_ckpt, &A.R_row_ckpt, &A.temp1, &A.temp2};
    for (int k = 0; k < 10; k++) {
        const size_t n = k < 8 ? mx : BL;
        *bufs[k] = (int16_t *)malloc(n * 2);
        for (size_t z = 0; z < n; z++) (*bufs[k])[z] = B_MIN;      /* Allocated::clear, :1704-1721 */
    }
    int32_t score = 0;
    size_t ri = 0, rj = 0;
    align_aa_core(P, Q, R, mn, mx, x_drop, &A, &T, &score, &ri, &rj);
    res->score = score;
    res->query_idx = (uint32_t)ri;
    res->reference_idx = (uint32_t)rj;
    if (n_ops) *n_ops = 0;
    if (ops && n_ops && P->trace) *n_ops = (uint32_t)trace_cigar(&T, ri, rj, ops, ops_cap, eq ? qs : NULL, eq ? rs : NULL);
    if (growth_buf && P->trace) {
        growth_n = (uint32_t)T.block_idx;
        for (size_t k = 0; k < T.block_idx && k < growth_cap; k++) {
            growth_buf[4 * k] = T.block_start[2 * k];
            growth_buf[4 * k + 1] = T.block_start[2 * k + 1];
            growth_buf[4 * k + 2] = (uint32_t)T.block_size[2 * k] << 16 | T.block_size[2 * k + 1];
            growth_buf[4 * k + 3] = T.right[k];
        }
    }
    for (int k = 0; k < 10; k++) free(*bufs[k]);
    free(T.trace); free(T.trace2); free(T.right); free(T.block_start); free(T.block_size);
    free(qs); free(rs); free(qb); free(rb);
    return 0;
}

/* the same with the AAMatrix table itself: scores27x32[a * 32 + b] (what the C API objects of oracle/ref_block_capi.cpp hold) */
int mmo_block_align_table(const uint8_t *q, const int16_t *qbias, int qlen, const uint8_t *r, const int16_t *rbias, int rlen,
                          const int8_t *scores, int gap_open, int gap_extend, int min_size, int max_size, int x_drop, mmo_block_res *res,
                          uint8_t *ops, uint32_t ops_cap, uint32_t *n_ops) {
    bparams P;
    P.scores = scores;
    P.gap_open = gap_open;
    P.gap_extend = gap_extend;
    P.kind = BK_AA;
    P.trace = P.xdrop = 1;
    P.prof = NULL;
    return block_run(q, qbias, qlen, r, rbias, rlen, &P, B_NULL, min_size, max_size, x_drop, res, ops, ops_cap, n_ops, 0);
}

/* Every instantiation the crate's own unit tests use (scan_block.rs:2267-2432): Block<trace, xdrop>::align over an AAMatrix
 * (kind 0: table [27 * 32], bytes = letter - 'A', NULL = 26), a NucMatrix (kind 1: table [8 * 16], bytes = upper-case ASCII,
 * NULL = 'Z') or a ByteMatrix (kind 2: table = {match, mismatch}, raw bytes, NULL = 0).  ops: Operation codes of cigar.rs:10-31
 * (M 1, = 2, X 3, I 4, D 5) from the end position back to the origin; eq selects cigar_eq. */
int mmo_block_align_generic(const uint8_t *q, int qlen, const uint8_t *r, int rlen, int kind, const int8_t *table, int gap_open, int gap_extend,
                            int min_size, int max_size, int x_drop, int trace, int xdrop, int eq, mmo_block_res *res, uint8_t *ops,
                            uint32_t ops_cap, uint32_t *n_ops) {
    if (kind < BK_AA || kind > BK_BYTES || (xdrop && x_drop < 0)) return -1;      /* :872-874 */
    bparams P;
    P.scores = table;
    P.gap_open = gap_open;
    P.gap_extend = gap_extend;
    P.kind = kind;
    P.trace = trace != 0;
    P.xdrop = xdrop != 0;
    P.prof = NULL;
    const uint8_t null_byte = kind == BK_AA ? B_NULL : (kind == BK_NUC ? (uint8_t)'Z' : 0);
    return block_run(q, NULL, qlen, r, NULL, rlen, &P, null_byte, min_size, max_size, x_drop, res, ops, ops_cap, n_ops, eq);
}

/* SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ> (StripedSmithWaterman.cpp:943-1127) for one pair whose forward scan
 * ended at (q_end, t_end) with `score`: reversed prefixes, block sizes 32, 64, .. 4096 until the block aligner reaches the
 * score, accepted only if it equals it (:1058).  comp_bias = ssw_init's int8 rounding of the composition bias (may be NULL).
 * Returns 1 and fills start positions / identities / backtrace (forward order, 'M' 'I' 'D'), or 0 = "Block alignment failed"
 * (the caller falls back to the reverse scan + banded traceback, :873-882). */
int mmo_sw_block_backtrace(const uint8_t *q, const int8_t *comp_bias, int qlen, const uint8_t *t, int tlen, const int8_t *mat, int alphabet,
                           int gap_open, int gap_extend, int score, int q_end, int t_end, int *q_start, int *t_start, uint32_t *ident,
                           char *bt, int bt_cap, int *bt_len, int *block_size_used) {
    (void)tlen;
    const int qa = q_end + 1, ta = t_end + 1;
    uint8_t *qr = (uint8_t *)malloc((size_t)qa + 1), *trv = (uint8_t *)malloc((size_t)ta + 1);
    int16_t *qb = (int16_t *)calloc((size_t)qa + 1, 2);
    for (int k = 0; k < qa; k++) { qr[k] = q[q_end - k]; qb[k] = comp_bias ? comp_bias[q_end - k] : 0; }      /* :996-997, :1466 */
    for (int k = 0; k < ta; k++) trv[k] = t[t_end - k];                                                       /* :1011-1013 */
    uint8_t *ops = (uint8_t *)malloc((size_t)qa + ta + 8);
    mmo_block_res res;
    res.score = -1000000000;
    res.query_idx = res.reference_idx = 0;
    uint32_t n_ops = 0;
    int used = 0;
    for (int min_size = 32; min_size <= 4096 && res.score < score; min_size *= 2) {                            /* :1021-1038 */
        const int x_drop = -(min_size * (-gap_extend) + (-gap_open));
        mmo_block_align(qr, qb, qa, trv, NULL, ta, mat, alphabet, -gap_open, -gap_extend, min_size, 4096, x_drop, &res, ops, (uint32_t)(qa + ta + 8), &n_ops);
        used = min_size;
    }
    int ok = 0;
    if (block_size_used) *block_size_used = used;
    if (!(res.score != score && !(score == 32767 && res.score >= score))) {                                    /* :1058 */
        /* Cigar::add appends while cigar_core walks back from the end position (cigar.rs:72-81), and block_get_cigar(cigar, i)
         * = Cigar::s[idx - 1 - i] (:88-90) counts from the LAST run added: the reference's loop (:1071-1105) therefore sees the
         * runs from the origin of the reversed sequences towards the end position, queryPos = targetPos = 0 at the origin. */
        uint32_t ids = 0, qp = 0, tp = 0;
        int n = 0;
        for (uint32_t k = n_ops; k-- > 0;) {
            char ch;
            if (ops[k] == 1) { ids += qr[qp] == trv[tp]; qp++; tp++; ch = 'M'; }      /* :1073-1082 */
            else if (ops[k] == 4) { qp++; ch = 'I'; }                                 /* :1083-1087 */
            else { tp++; ch = 'D'; }                                                  /* :1094-1098 */
            if (bt && n < bt_cap) bt[n] = ch;
            n++;
        }
        if (bt && n <= bt_cap) {                       /* std::reverse(backtrace), :1110 */
            for (int a = 0, b = n - 1; a < b; a++, b--) { const char c = bt[a]; bt[a] = bt[b]; bt[b] = c; }
        }
        if (bt_len) *bt_len = n;
        if (ident) *ident = ids;
        if (q_start) *q_start = (q_end + 1) - (int)qp;    /* :1111-1112 */
        if (t_start) *t_start = (t_end + 1) - (int)tp;
        ok = 1;
    }
    free(qr); free(trv); free(qb); free(ops);
    return ok;
}

/* ---- profile alignments: Block<TRACE, X_DROP>::align_profile (scan_block.rs:919-944) ---------------------------------------------
 * An AAProfile as AAProfile::new leaves it (scores.rs:494-507: every score and gap cost i8::MIN) with `pos_aa_rows` installed for
 * the positions 1 .. plen (row p = 32 int8 scores of profile position p - 1, indexed by the converted query byte; set / set_all /
 * the reference's memcpy all write these rows, :544-551, StripedSmithWaterman.cpp:970-989) and the three gap-cost arrays for the
 * indices 0 .. gap_n - 1 (from_bytes sets 0 .. plen, :519-523; set_all_gap_* fill all curr_len entries, :575-587).  pad_block =
 * the block_size the profile was created with (curr_len = plen + pad_block + 1). */
typedef struct {
    bprofile p;
    int8_t *pos_aa;
    int16_t *aa_pos, *goc, *gcc, *gor;
} bprofile_own;

static int profile_build(bprofile_own *o, int plen, int pad_block, const int8_t *pos_aa_rows, const int16_t *gap_open_C, const int16_t *gap_close_C,
                         const int16_t *gap_open_R, int gap_n, int gap_extend) {
    const size_t cl = (size_t)plen + (size_t)pad_block + 1;
    if (plen < 0 || pad_block < BL || gap_n < 0 || (size_t)gap_n > cl) return -1;
    o->pos_aa = (int8_t *)malloc((cl + BL) * 32);
    o->aa_pos = (int16_t *)malloc(32 * (cl + BL) * 2);
    o->goc = (int16_t *)malloc((cl + BL) * 2);
    o->gcc = (int16_t *)malloc((cl + BL) * 2);
    o->gor = (int16_t *)malloc((cl + BL) * 2);
    memset(o->pos_aa, -128, (cl + BL) * 32);
    for (size_t k = 0; k < cl + BL; k++) o->goc[k] = o->gcc[k] = o->gor[k] = -128;
    for (int p = 0; p < plen; p++) memcpy(o->pos_aa + (size_t)(p + 1) * 32, pos_aa_rows + (size_t)p * 32, 32);
    for (size_t k = 0; k < 32 * (cl + BL); k++) o->aa_pos[k] = -128;
    for (size_t i = 0; i < cl; i++)
        for (int b = 0; b < 32; b++) o->aa_pos[(size_t)b * cl + i] = o->pos_aa[i * 32 + b];      /* both arrays are written together, :544-551 */
    for (int k = 0; k < gap_n; k++) { o->goc[k] = gap_open_C[k]; o->gcc[k] = gap_close_C[k]; o->gor[k] = gap_open_R[k]; }
    o->p.pos_aa = o->pos_aa; o->p.aa_pos = o->aa_pos; o->p.gap_open_C = o->goc; o->p.gap_close_C = o->gcc; o->p.gap_open_R = o->gor;
    o->p.gap_extend = gap_extend; o->p.len = plen; o->p.curr_len = cl;
    o->p.upstream_gaps = 0;
    return 0;
}
static void profile_free(bprofile_own *o) { free(o->pos_aa); free(o->aa_pos); free(o->goc); free(o->gcc); free(o->gor); }

/* Block<trace, xdrop>::align_profile(query, profile, min_size ..= max_size, x_drop) + res (+ cigar): q = query bytes after
 * AAMatrix::convert_char (letter - 'A'; padded with 26); the rest as profile_build.  The crate's test_profile vectors
 * (scan_block.rs:2432-2477) run through this (tests/test_block_oracle.py). */
int mmo_block_align_profile(const uint8_t *q, int qlen, int plen, int pad_block, const int8_t *pos_aa_rows, const int16_t *gap_open_C,
                            const int16_t *gap_close_C, const int16_t *gap_open_R, int gap_n, int gap_extend, int min_size, int max_size,
                            int x_drop, int trace, int xdrop, mmo_block_res *res, uint8_t *ops, uint32_t ops_cap, uint32_t *n_ops) {
    if (xdrop && x_drop < 0) return -1;      /* :927-929 */
    bprofile_own o;
    /* trace / xdrop bit 1: the upstream crate's gap-open form (bprofile::upstream_gaps) - test hook */
    const int upstream = (trace & 2) != 0;
    trace &= 1;
    if (profile_build(&o, plen, pad_block, pos_aa_rows, gap_open_C, gap_close_C, gap_open_R, gap_n, gap_extend) != 0) return -1;
    o.p.upstream_gaps = upstream;
    bparams P;
    P.scores = NULL;
    P.gap_open = P.gap_extend = 0;
    P.kind = BK_AA;
    P.trace = trace != 0;
    P.xdrop = xdrop != 0;
    P.prof = &o.p;
    const int rc = block_run(q, NULL, qlen, NULL, NULL, plen, &P, B_NULL, min_size, max_size, x_drop, res, ops, ops_cap, n_ops, 0);
    profile_free(&o);
    return rc;
}

/* SmithWaterman::alignStartPosBacktraceBlock<PROFILE_SEQ> (StripedSmithWaterman.cpp:943-1127) for one (profile query, target) pair
 * whose forward scan ended at (q_end, t_end) with `score`.  prof = the query's int8 score rows [alphabet][qlen] (ssw_init's
 * profile->mat for a profile query, what mmgpu_sw_query.profile carries), q = its consensus sequence (identities are counted
 * against it, :1076).  The profile handed to the crate is the reversed prefix 0 .. q_end (pos_aa_rev rows, :970-980), every gap
 * opening costs gap_open, closing costs nothing, extension gap_extend (:988-990); the TARGET is the crate's `query` (rows of its DP
 * matrix), so its I / D operations change sides on the way back (:1083-1101).  Same return contract as mmo_sw_block_backtrace. */
int mmo_sw_block_backtrace_profile(const int8_t *prof, const uint8_t *q, int qlen, const uint8_t *t, int tlen, int alphabet, int gap_open,
                                   int gap_extend, int score, int q_end, int t_end, int *q_start, int *t_start, uint32_t *ident, char *bt,
                                   int bt_cap, int *bt_len, int *block_size_used) {
    (void)tlen;
    if (alphabet < 1 || alphabet > 32) return 0;
    const int qa = q_end + 1, ta = t_end + 1;
    int8_t *rows = (int8_t *)malloc((size_t)qa * 32 + 32);
    memset(rows, -128, (size_t)qa * 32 + 32);                       /* memset(pos_aa_rev, 0x80, ..), :1452; only alphabetSize entries are copied, :975-980 */
    uint8_t *qr = (uint8_t *)malloc((size_t)qa + 1), *trv = (uint8_t *)malloc((size_t)ta + 1);
    for (int k = 0; k < qa; k++) {
        qr[k] = q[q_end - k];
        for (int a = 0; a < alphabet; a++) rows[(size_t)k * 32 + a] = prof[(size_t)a * qlen + (q_end - k)];
    }
    for (int k = 0; k < ta; k++) trv[k] = t[t_end - k];
    const int pad_block = 4096;                                      /* block_new_aaprofile(queryAlnLen, MAX_SIZE, gaps.extend), :965 */
    const size_t cl = (size_t)qa + pad_block + 1;
    int16_t *goc = (int16_t *)malloc(cl * 2), *gcc = (int16_t *)malloc(cl * 2), *gor = (int16_t *)malloc(cl * 2);
    for (size_t k = 0; k < cl; k++) { goc[k] = (int16_t)(-gap_open); gcc[k] = 0; gor[k] = (int16_t)(-gap_open); }   /* set_all_gap_*, :988-990 */
    uint8_t *ops = (uint8_t *)malloc((size_t)qa + ta + 8);
    mmo_block_res res;
    res.score = -1000000000;
    res.query_idx = res.reference_idx = 0;
    uint32_t n_ops = 0;
    int used = 0;
    for (int min_size = 32; min_size <= 4096 && res.score < score; min_size *= 2) {                            /* :1039-1049 */
        const int x_drop = -(min_size * (-gap_extend) + (-gap_open));
        mmo_block_align_profile(trv, ta, qa, pad_block, rows, goc, gcc, gor, (int)cl, -gap_extend, min_size, 4096, x_drop, 1, 1, &res, ops,
                                (uint32_t)(qa + ta + 8), &n_ops);
        used = min_size;
    }
    int ok = 0;
    if (block_size_used) *block_size_used = used;
    if (!(res.score != score && !(score == 32767 && res.score >= score))) {                                    /* :1058 */
        uint32_t ids = 0, qp = 0, tp = 0;
        int n = 0;
        for (uint32_t k = n_ops; k-- > 0;) {
            char ch;
            if (ops[k] == 1) { ids += qr[qp] == trv[tp]; qp++; tp++; ch = 'M'; }      /* :1073-1082 */
            else if (ops[k] == 4) { tp++; ch = 'D'; }                                 /* PROFILE_SEQ: I of the crate = the target advances, :1088-1091 */
            else { qp++; ch = 'I'; }                                                  /* :1099-1102 */
            if (bt && n < bt_cap) bt[n] = ch;
            n++;
        }
        if (bt && n <= bt_cap) {
            for (int a = 0, b = n - 1; a < b; a++, b--) { const char c = bt[a]; bt[a] = bt[b]; bt[b] = c; }
        }
        if (bt_len) *bt_len = n;
        if (ident) *ident = ids;
        if (q_start) *q_start = (q_end + 1) - (int)qp;
        if (t_start) *t_start = (t_end + 1) - (int)tp;
        ok = 1;
    }
    free(rows); free(qr); free(trv); free(goc); free(gcc); free(gor); free(ops);
    return ok;
}
