/* libmmgpu - C-ABI of the MI355X-native prefilter -> align hot path of MMseqs2.
 *
 * Drop-in boundary.  The reference has no plugin API on this path; the two
 * seams are C++ member functions whose per-query loop bodies this library replaces:
 *
 *   Alignment::run        src/alignment/Alignment.cpp:248-542   (Matcher::initQuery :340 +
 *                                                                Matcher::getSWResult :379)
 *   Prefiltering::runSplit src/prefiltering/Prefiltering.cpp:755-982 (QueryMatcher::matchQuery :873)
 *
 * The in-tree precedent for such a seam is `class Marv` (lib/libmarv/src/marv.h:6-57, driven from
 * src/prefiltering/ungappedprefilter.cpp:144-158,205-207).  Conventions follow the reference:
 *  - the caller owns every host buffer; the library copies into HBM at *_load / *_upload time and never
 *    frees or retains host pointers beyond the call;
 *  - no exceptions (the reference is -fno-exceptions): every entry point returns 0 on success and a
 *    negative code on failure, with a message available from mmgpu_last_error();
 *  - sequences are *numeric* residues exactly as Sequence::numSequence / SequenceLookup hold them
 *    (aa2num order of the matrix file, X = alphabet-1), matrices are the int8 tables the reference
 *    builds (Matcher::setSubstitutionMatrix, Matcher.cpp:29-36);
 *  - a context may be used from any host thread, one call at a time per context.
 *
 * Nothing in this header names a torch type; PyTorch (tests, bench.py) talks to it through ctypes.
 */
#ifndef MMGPU_H
#define MMGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmgpu_ctx mmgpu_ctx;

#define MMGPU_OK 0
#define MMGPU_ERR_ARG (-1)     /* invalid argument */
#define MMGPU_ERR_HIP (-2)     /* HIP runtime failure (message has hipGetErrorString) */
#define MMGPU_ERR_STATE (-3)   /* call order violated, e.g. no targets loaded */
#define MMGPU_ERR_UNSUPPORTED (-4)

/* ---- context ------------------------------------------------------------------------------- */
/* One context per GPU (one process per GPU in multi-GPU runs).  Replaces `new Marv(...)`
 * (ungappedprefilter.cpp:144). */
int mmgpu_init(mmgpu_ctx **ctx, int device_id);
void mmgpu_destroy(mmgpu_ctx *ctx);
const char *mmgpu_last_error(void);
/* Work is issued on this HIP stream (a hipStream_t; NULL = the legacy default stream).  bench.py passes
 * torch's current stream so that torch.cuda events bracket the kernels. */
int mmgpu_set_stream(mmgpu_ctx *ctx, void *hip_stream);
int mmgpu_synchronize(mmgpu_ctx *ctx);
/* number of compute units / device name, for reports */
int mmgpu_device_info(mmgpu_ctx *ctx, int *compute_units, char *name, int name_cap);

/* ---- host-side helpers (run on the CPU in the reference too) ------------------
 * SubstitutionMatrix::calcLocalAaBiasCorrection (src/commons/SubstitutionMatrix.cpp:79-112): float composition
 * bias, window +-20.  submat is BaseMatrix::subMatrix (short, alphabet x alphabet row-major), pback is
 * BaseMatrix::pBack.  Bit-identical to the reference's float/double evaluation order. */
int mmgpu_host_comp_bias(const int16_t *submat, const double *pback, int alphabet, const uint8_t *seq, uint32_t len,
                         float scale, float *out);
/* ssw_init's rounding of that bias to int8 (StripedSmithWaterman.cpp:1378-1380) */
int mmgpu_host_round_comp_bias(const float *bias, uint32_t len, int8_t *out);

/* ---- target database ---------------------------------------------------------------------------
 * residues/offsets are SequenceLookup's `data` / `offsets[n+1]` (src/prefiltering/SequenceLookup.h:44-48),
 * or equivalently Sequence::numSequence of every DBReader entry concatenated.  Replaces Marv::loadDb
 * (ungappedprefilter.cpp:153-158).  The copy in HBM is re-laid out (4-byte aligned starts, lengths,
 * length-sorted id list); ids in every later call are indices into this array (shard-local ids,
 * the reference's dbFrom convention, Prefiltering.cpp:879-881). */
int mmgpu_load_targets(mmgpu_ctx *ctx, const uint8_t *residues, const uint64_t *offsets, uint32_t n_targets,
                       int alphabet);

/* ---- gapped alignment (behind Alignment::run) -------------------------------------------------- */
typedef struct {
    const int8_t *mat; /* alphabet*alphabet, row-major: Matcher::tinySubMat (Matcher.cpp:29-36) */
    int alphabet;      /* 21 for amino acids */
    int gap_open;      /* cost of the first gap residue, par.gapOpen (11) */
    int gap_extend;    /* par.gapExtend (1) */
} mmgpu_sw_params;

typedef struct {
    const uint8_t *q;         /* Sequence::numSequence of the query */
    uint32_t qlen;
    const int8_t *comp_bias;  /* s_profile::composition_bias, already rounded as ssw_init does
                                 (StripedSmithWaterman.cpp:1378-1380); NULL = no correction */
    const uint32_t *target_ids; /* the query's prefilter list, in list order */
    uint32_t n_targets;
    int32_t min_start_score;  /* MMGPU_SW_START only: run the reverse scan for pairs with score >= this.
                                 The host derives it from the E-value threshold (the smallest raw score whose
                                 EvalueComputation::computeEvalue passes -e; ssw_align_private returns early
                                 otherwise, StripedSmithWaterman.cpp:857-863).  <= 1 means every scoring pair. */
} mmgpu_sw_query;

/* s_align (StripedSmithWaterman.h:52-67) restricted to what the kernels produce. */
typedef struct {
    int32_t score;   /* score1, saturating at 32767 like sw_sse2_word */
    int32_t q_end;   /* qEndPos1 */
    int32_t t_end;   /* dbEndPos1; -1 when score == 0 */
    int32_t q_start; /* qStartPos1, -1 unless mode >= 1 */
    int32_t t_start; /* dbStartPos1, -1 unless mode >= 1 */
    int32_t word;    /* 1 when the reference would have left its uint8 pass (score + bias >= 255) */
} mmgpu_sw_hit;

#define MMGPU_SW_SCORE_END 0 /* Matcher::SCORE_ONLY: score + end positions */
#define MMGPU_SW_START 1     /* + start positions (alignStartPosBacktrace's reverse scan) */

/* One call = the hit loop of Alignment::run (:346-397) for nq queries: for every (query, target) pair
 * the forward Gotoh scan (alignScoreEndPos, StripedSmithWaterman.cpp:892-941) and, with MMGPU_SW_START,
 * the reverse scan for the start position (:1129-1204) for the pairs reaching min_start_score.
 * out has sum(n_targets) entries, query-major, list order. */
int mmgpu_sw_batch(mmgpu_ctx *ctx, const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t n_queries,
                   int mode, mmgpu_sw_hit *out);

/* Split form of the same call, for callers that keep batches resident in HBM (and for bench.py, whose
 * timed region must start with inputs already on the device):
 *   prepare  = host-side scheduling + H2D copies, returns a batch handle
 *   run      = kernel launches only (asynchronous on the context's stream)
 *   fetch    = D2H of the results (synchronises) */
typedef struct mmgpu_sw_batch_t mmgpu_sw_batch_t;
int mmgpu_sw_prepare(mmgpu_ctx *ctx, const mmgpu_sw_params *params, const mmgpu_sw_query *queries,
                     uint32_t n_queries, int mode, mmgpu_sw_batch_t **batch);
int mmgpu_sw_run(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch);
int mmgpu_sw_fetch(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, mmgpu_sw_hit *out);
/* forward-DP cells of the batch = sum over pairs of qlen*tlen ("alignments calculated" x lengths,
 * Alignment.cpp:380,530) and the number of pairs */
int mmgpu_sw_batch_stats(mmgpu_sw_batch_t *batch, uint64_t *cells, uint64_t *pairs);
/* milliseconds spent in the kernels of the last mmgpu_sw_run of this batch, measured with HIP events on the
 * context's stream (synchronises) */
int mmgpu_sw_last_kernel_ms(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, float *ms);
/* mean over the last `last_n` runs (0 = all runs since prepare, at most 256 are recorded) */
int mmgpu_sw_kernel_ms_mean(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, uint32_t last_n, float *ms, uint32_t *n_used);
void mmgpu_sw_free(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch);

#ifdef __cplusplus
}
#endif
#endif /* MMGPU_H */
