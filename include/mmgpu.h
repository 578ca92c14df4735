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
/* free / total HBM of the context's device in bytes (hipMemGetInfo): what the caller sizes its query blocks against -
 * the scratch arrays of a context grow to the largest batch it has seen and are kept until mmgpu_destroy */
int mmgpu_device_memory(mmgpu_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- host-side helpers (run on the CPU in the reference too) ------------------
 * SubstitutionMatrix::calcLocalAaBiasCorrection (src/commons/SubstitutionMatrix.cpp:79-112): float composition
 * bias, window +-20.  submat is BaseMatrix::subMatrix (short, alphabet x alphabet row-major), pback is
 * BaseMatrix::pBack.  Bit-identical to the reference's float/double evaluation order. */
int mmgpu_host_comp_bias(const int16_t *submat, const double *pback, int alphabet, const uint8_t *seq, uint32_t len,
                         float scale, float *out);
/* ssw_init's rounding of that bias to int8 (StripedSmithWaterman.cpp:1378-1380) */
int mmgpu_host_round_comp_bias(const float *bias, uint32_t len, int8_t *out);

/* The two helpers above over a block of sequences (residues / offsets[n + 1] as in SequenceLookup), dealt to n_threads host
 * threads - what the reference computes per query inside the OpenMP loops of Prefiltering::runSplit (QueryMatcher.cpp:108-116)
 * and Alignment::run (ssw_init, StripedSmithWaterman.cpp:1371-1381).  out_float / out_round may each be NULL; indexed like residues. */
int mmgpu_host_comp_bias_batch(const int16_t *submat, const double *pback, int alphabet, const uint8_t *residues,
                               const uint64_t *offsets, uint32_t n, float scale, float *out_float, int8_t *out_round, int n_threads);

/* Length-bucket sharding of the target database over n_shards devices (multi-GPU runs, one process per device): shard_of[i]
 * and local_id[i] for every target (local ids ascend with the global ids inside a shard), shard_sizes[n_shards], optional
 * shard_residues[n_shards].  Every shard gets the same length distribution, so residues, index entries and alignment work
 * balance (the reference balances residues over its target splits, src/commons/DBReader.cpp:1108-1150). */
int mmgpu_host_partition_targets(const uint64_t *offsets, uint32_t n_targets, uint32_t n_shards, uint32_t *shard_of,
                                 uint32_t *local_id, uint32_t *shard_sizes, uint64_t *shard_residues);

/* ---- target database ---------------------------------------------------------------------------
 * residues/offsets are SequenceLookup's `data` / `offsets[n+1]` (src/prefiltering/SequenceLookup.h:44-48),
 * or equivalently Sequence::numSequence of every DBReader entry concatenated.  Replaces Marv::loadDb
 * (ungappedprefilter.cpp:153-158).  The copy in HBM is re-laid out (4-byte aligned starts, lengths,
 * length-sorted id list); ids in every later call are indices into this array (shard-local ids,
 * the reference's dbFrom convention, Prefiltering.cpp:879-881). */
int mmgpu_load_targets(mmgpu_ctx *ctx, const uint8_t *residues, const uint64_t *offsets, uint32_t n_targets,
                       int alphabet);
/* Loading a database drops a resident prefilter index (it indexes the previous database): call mmgpu_pf_load_index /
 * mmgpu_pf_build_index again, and free prefilter batches prepared against the old index first. */

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
    const int8_t *profile;    /* profile query (Parameters::DBTYPE_HMM_PROFILE): Sequence::getAlignmentProfile(), int8
                                 [PROFILE_AA_SIZE = 20 or more letters][qlen] letter-major, as ssw_init receives it
                                 (StripedSmithWaterman.cpp:1364-1420); q = the consensus sequence (numSequence, used for the
                                 identity count of the backtrace only), comp_bias is ignored (:1375-1384).  Letters from
                                 profile_letters on (the X state) score 0 (:1389-1390).  NULL = sequence query. */
    uint32_t profile_letters; /* letters the profile holds rows for (Sequence::PROFILE_AA_SIZE = 20); 0 with profile == NULL */
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
#define MMGPU_SW_START_NOT_WORD 2 /* + start positions for the hits of the reference's uint8 pass only (word == 0).  ssw_align_private
                                     takes the start of an int16-range hit (word == 1) from the block aligner and scans backwards
                                     only when that declines (StripedSmithWaterman.cpp:865-882): such pairs keep q_start = t_start
                                     = -1 here, mmgpu_sw_block_backtrace supplies their start positions, and the ones it answers
                                     MMGPU_BLOCK_DECLINED get their reverse scan from mmgpu_sw_reverse_pairs */

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
/* Fused hand-over: the alignment batch for the hit lists of a prefilter batch that has been run,
 * without the lists leaving the device.  queries[i] supplies q / qlen / comp_bias / min_start_score (target_ids and
 * n_targets are ignored); the result array has n_queries * min(max_hits, dbSize) slots, slot q * stride + k = hit k of
 * query q's list (zeroed beyond the list's length).  mmgpu_pf_hit is declared further down.  The alignment batch copies what
 * it needs (list lengths, target id of every slot) during this call: afterwards the prefilter batch may be re-run or freed. */
struct mmgpu_pf_batch_t;
int mmgpu_sw_prepare_from_pf(mmgpu_ctx *ctx, const mmgpu_sw_params *params, const mmgpu_sw_query *queries,
                             uint32_t n_queries, int mode, struct mmgpu_pf_batch_t *pf_batch, mmgpu_sw_batch_t **batch);
/* The same for hit lists in DEVICE memory that did not come from a prefilter batch of this context (multi-GPU runs: the
 * merged lists, localised by mmgpu_pf_localize_lists): d_hits [n_queries][stride] mmgpu_pf_hit with ids of the resident
 * targets, d_counts [n_queries] uint32.  The lists are copied during the call. */
int mmgpu_sw_prepare_from_lists(mmgpu_ctx *ctx, const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t n_queries,
                                int mode, const void *d_hits, const void *d_counts, uint32_t stride, mmgpu_sw_batch_t **batch);
int mmgpu_sw_run(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch);
int mmgpu_sw_fetch(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, mmgpu_sw_hit *out);
/* The reverse scan (StripedSmithWaterman.cpp:1129-1204) after the fact, for exactly the pairs named (indices into the batch's result
 * array): the fall-back of :873-882 for int16-range hits of a MMGPU_SW_START_NOT_WORD batch the block aligner declined.  The forward
 * results stay; q_start / t_start of the named pairs are filled in the batch (later mmgpu_sw_fetch / mmgpu_sw_traceback calls see
 * them) and, when out != NULL, the n records are copied to out[0..n) in pair_index order.  Synchronises. */
int mmgpu_sw_reverse_pairs(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, const uint32_t *pair_index, uint32_t n_pairs, mmgpu_sw_hit *out);
/* Same results, copied device -> device into caller-owned device memory (the multi-GPU result exchange works on
 * device tensors); asynchronous on the context's stream. */
int mmgpu_sw_fetch_device(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, void *d_out);
/* forward-DP cells of the batch = sum over pairs of qlen*tlen ("alignments calculated" x lengths,
 * Alignment.cpp:380,530) and the number of pairs */
int mmgpu_sw_batch_stats(mmgpu_sw_batch_t *batch, uint64_t *cells, uint64_t *pairs);
/* milliseconds spent in the kernels of the last mmgpu_sw_run of this batch, measured with HIP events on the
 * context's stream (synchronises) */
int mmgpu_sw_last_kernel_ms(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, float *ms);
/* mean over the last `last_n` runs (0 = all runs since prepare, at most 256 are recorded) */
int mmgpu_sw_kernel_ms_mean(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, uint32_t last_n, float *ms, uint32_t *n_used);
void mmgpu_sw_free(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch);

/* ---- backtrace / CIGAR (behind alignStartPosBacktrace's banded_sw, StripedSmithWaterman.cpp:1220-1247,1478-1693) ----
 * For the pairs the host selects after its E-value / coverage checks (Matcher::SCORE_COV_SEQID or -a): the
 * reference's banded scalar Gotoh over [q_start..q_end] x [t_start..t_end] (band |dlen|+1, doubled until the score
 * is reached) with its tie rules, and the walk back from the bottom-right corner.  Output per pair: the backtrace
 * string of computerBacktrace (:1280-1308) - 'M', 'I' (consumes query), 'D' (consumes target) - and
 * identicalAACnt. */
typedef struct {
    uint64_t bt_off;  /* offset of this pair's string in the caller's bt buffer */
    uint32_t bt_len;  /* its length (no terminating NUL); 0 unless status == MMGPU_BT_OK */
    uint32_t ident;   /* identicalAACnt */
    int32_t status;
    int32_t reserved;
} mmgpu_sw_bt;
#define MMGPU_BT_OK 0
#define MMGPU_BT_TOO_LARGE 1 /* band/direction storage above the device scratch budget: host runs banded_sw itself */
#define MMGPU_BT_FAILED 2    /* the walk left the band (the reference reads unrelated memory there) */
#define MMGPU_BT_NO_START 3  /* the pair has no start position (mode 0, score below min_start_score, score 0) */
/* pair_index: indices into the batch's result array (the order of mmgpu_sw_fetch); the batch must have been run with
 * MMGPU_SW_START.  Strings are written back to back in pair_index order (each pair reserves
 * (q_end-q_start+1)+(t_end-t_start+1)+1 bytes); *bt_used receives the bytes reserved.  If bt_cap is too small the
 * call fails with MMGPU_ERR_ARG and *bt_used holds the size needed. */
int mmgpu_sw_traceback(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, const uint32_t *pair_index, uint32_t n_pairs,
                       mmgpu_sw_bt *info, char *bt, size_t bt_cap, size_t *bt_used);

/* Int16-range hits (mmgpu_sw_hit::word == 1) whose start position the reference takes from the block aligner, not from the
 * reverse scan: SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ> and, since round 5, <PROFILE_SEQ> (StripedSmithWaterman.cpp:
 * 943-1127 -> lib/block-aligner 0.4.0, AVX2 configuration; a profile query's score rows take the place of matrix + bias and the
 * query sits on the crate's reference side, :963-990) on the device (block_kernel.hip).  One call serves what the reference gets
 * from one block-aligner run: start positions (modes >= 1), identities and the backtrace string (mode 3 / -a).
 * Restated from the crate's source; this image has no Rust toolchain, so equality with a Rust-linked build is pinned only as
 * far as tests/test_block_oracle.py goes (the crate's own test vectors, the invariants the reference checks, the recorded
 * vectors of scripts/make_block_goldens.sh once a Rust-equipped box has produced them). */
typedef struct {
    int32_t q_start, t_start;   /* qStartPos1 / dbStartPos1 (:1111-1112) */
    uint32_t ident;             /* identicalAACnt */
    uint32_t bt_len;
    uint64_t bt_off;            /* the pair's backtrace in the caller's bt buffer ('M', 'I', 'D'; forward order) */
    int32_t status;
    int32_t reserved;
} mmgpu_sw_block;
#define MMGPU_BLOCK_OK 0
#define MMGPU_BLOCK_DECLINED 1   /* "Block alignment failed" (:1058,873-882): the block aligner's score differs from the pair's -
                                    the reference falls back to the reverse scan + banded traceback (q_start / t_start of
                                    mmgpu_sw_fetch, mmgpu_sw_traceback) */
#define MMGPU_BLOCK_TOO_LARGE 2  /* not decided on the device: the pair's scratch (4096-row blocks, as the crate's) could not be
                                    allocated.  Blocks grow to the crate's own 4096 rows (second launch for the pairs that
                                    leave the 512-row LDS form), so this no longer depends on the pair */
#define MMGPU_BLOCK_NOT_WORD 3   /* not an int16-range hit with a positive score */
/* pair_index as for mmgpu_sw_traceback (any mode: only score / q_end / t_end of the forward scan are read); every pair reserves
 * (q_end + 1) + (t_end + 1) + 1 bytes of bt, rounded up to a multiple of four (*bt_used of a call with bt == NULL is the size to
 * bring).  bt == NULL with bt_cap == MMGPU_BLOCK_NO_STRINGS: the caller wants start positions, identities and bt_len only (the
 * strings stay on the device, no download).  bt == NULL with bt_cap == MMGPU_BLOCK_STARTS_ONLY: start positions and status only -
 * what `mmseqs search` consumes in alignment mode 2 without -a (Matcher.cpp:107-127: neither identicalAACnt nor the backtrace
 * reach the record there); ident and bt_len come back 0, the device keeps no trace and walks nothing back. */
#define MMGPU_BLOCK_NO_STRINGS ((size_t)-1)
#define MMGPU_BLOCK_STARTS_ONLY ((size_t)-2)
int mmgpu_sw_block_backtrace(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, const uint32_t *pair_index, uint32_t n_pairs,
                             mmgpu_sw_block *out, char *bt, size_t bt_cap, size_t *bt_used);
/* Search semantics in one call, for a batch run with MMGPU_SW_START_NOT_WORD: the device itself picks every int16-range hit whose score
 * reaches its query's min_start_score (what passes ssw_align_private's E-value gate and then asks the block aligner,
 * StripedSmithWaterman.cpp:857-882), runs the block aligner for start positions only, writes q_start / t_start into the batch's result
 * records, and runs the reverse scan for the pairs it declined (:873-882).  mmgpu_sw_fetch afterwards returns, for every pair that
 * passes the gate, the start position the reference reports.  n_too_large: pairs left undecided (start positions -1; 0 unless the
 * scratch pool could not be allocated).  Callers with gates of their own beyond the score (coverage pre-checks, Alignment.cpp) use
 * mmgpu_sw_block_backtrace with their own pair list instead.  Synchronises. */
int mmgpu_sw_block_starts(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, uint32_t *n_selected, uint32_t *n_declined, uint32_t *n_too_large);
/* Test aid (the growth-sequence test of tests/test_sw_gpu.py): the same run without strings, plus every pair's block list as it
 * stands when the crate's align_core returns - Trace::block_start / block_size / right, i.e. the sequence of grow / shift-right /
 * shift-down steps with their sizes after every x-drop restore.  growth: n_pairs x (1 + 4 * growth_cap) words, per pair the number
 * of blocks (it may exceed growth_cap: the list is cut, not the count; 0 for MMGPU_BLOCK_TOO_LARGE) and (i, j, height << 16 | width,
 * right) per block. */
int mmgpu_sw_block_growth(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, const uint32_t *pair_index, uint32_t n_pairs, mmgpu_sw_block *out,
                          uint32_t *growth, uint32_t growth_cap);
/* reporting: how many pairs of the batch's LAST mmgpu_sw_block_backtrace call were decided with blocks up to 512 rows (two pairs
 * per wavefront with blocks up to 128 rows, then one pair per wavefront with the borders in LDS) and how many needed the later
 * launches with the crate's full 4096-row blocks (borders in HBM) */
int mmgpu_sw_block_tiers(const mmgpu_sw_batch_t *batch, uint32_t *first_tier, uint32_t *second_tier);

/* ---- nucleotide alignment step (behind Alignment::run for nucleotide databases) --------------------------------
 * BandedNucleotideAligner::align (src/alignment/BandedNucleotideAligner.cpp:76-263; Matcher::getSWResult calls it
 * instead of the Smith-Waterman for DBTYPE_NUCLEOTIDES, Matcher.cpp:76-79): ungapped seed on the prefilter diagonal,
 * left extension on the reversed sequences, right extension with CIGAR by ksw_extz2_sse (band 64, z-drop), backtrace.
 * Targets are the resident database (mmgpu_load_targets with alphabet 5, numeric codes of NucleotideMatrix:
 * A C T G X = 0..4).
 *
 * past_end_query / past_end_target: the reference reads ONE residue past the end of both sequences
 * (SmithWaterman::seq_reverse is called with L where it expects L - 1, BandedNucleotideAligner.cpp:61,68,93), i.e.
 * whatever Sequence::numSequence[L] holds from earlier sequences.  That letter is an explicit input here (4 = X
 * scores 0 in the extension); a host that wants its own run reproduced passes what its buffers hold. */
typedef struct {
    const int8_t *mat;        /* 5 x 5, NucleotideMatrix::subMatrix as int8 (nucleotide.out: match 2, mismatch -3) */
    const uint8_t *reverse;   /* [5] NucleotideMatrix::reverseResidue */
    int gap_open, gap_extend; /* 5, 2 (Parameters: gapOpen / gapExtend for nucleotides) */
    int zdrop;                /* par.zdrop (40) */
    int past_end_query, past_end_target;
    int wrapped;              /* --wrapped-scoring (circular sequences): every query is its sequence written twice (Alignment.cpp:332-337);
                               * the seed may wrap around, the extensions run over the original length at most (:98-113,171-174,189-191) */
} mmgpu_nucl_params;
typedef struct {
    const uint8_t *q;  /* Sequence::numSequence of the query (forward strand) */
    uint32_t qlen;
} mmgpu_nucl_query;
typedef struct {
    uint32_t query;    /* index into the queries of the call */
    uint32_t target;   /* id in the resident database */
    uint16_t diagonal; /* hit_t::diagonal of the prefilter hit */
    uint8_t reverse;   /* 1: align the reverse complement of the query (Matcher::getSWResult's isReverse) */
    uint8_t past_end;  /* 0: the letters of mmgpu_nucl_params; MMGPU_NUCL_PAST_END(q, t): this pair's own letters one residue
                          past the query's / the target's end - what a host that replays the reference's buffer history
                          (integration/MMGpuNuclAlignRun.cpp) knows per pair */
} mmgpu_nucl_pair;
#define MMGPU_NUCL_PAST_END(q, t) ((uint8_t)(0x80u | ((unsigned)(q) & 7u) | (((unsigned)(t) & 7u) << 3)))
typedef struct {
    int32_t score;              /* s_align::score1 */
    int32_t q_start, q_end;     /* on the aligned strand, like s_align::qStartPos1 / qEndPos1 */
    int32_t t_start, t_end;
    uint32_t ident;             /* identicalAACnt */
    uint32_t bt_len;            /* letters of the backtrace (M / I / D), 0-terminated in bt */
    int32_t status;             /* MMGPU_NUCL_* */
    uint64_t bt_off;            /* offset of this pair's string in bt */
} mmgpu_nucl_hit;
#define MMGPU_NUCL_OK 0
#define MMGPU_NUCL_BT_OVERFLOW 1 /* bt_cap was too small for this pair's string: positions and score are valid */
/* Synchronous.  out[i] belongs to pairs[i]; strings are packed into bt in completion order; *bt_used = bytes needed
 * (sum of bt_len + 1).  bt_cap = sum over pairs of (qlen + tlen + 2) always suffices. */
int mmgpu_nucl_align(mmgpu_ctx *ctx, const mmgpu_nucl_params *params, const mmgpu_nucl_query *queries, uint32_t n_queries,
                     const mmgpu_nucl_pair *pairs, uint32_t n_pairs, mmgpu_nucl_hit *out, char *bt, uint64_t bt_cap,
                     uint64_t *bt_used);

/* ---- k-mer prefilter (behind Prefiltering::runSplit) ---------------------------------------------------------
 * Host-side table builders.  In a drop-in build the reference's own objects supply these tables
 * (ExtendedSubstitutionMatrix::calcScoreMatrix, Prefiltering.cpp:220-225; IndexBuilder::fillDatabase,
 * Prefiltering.cpp:544-583) and the two functions below are not needed; they exist so that callers without the
 * reference (bench.py, tests) can build bit-identical tables. */
/* ScoreMatrix of all span-mers over the first kalph = alphabet-1 letters (ExtendedSubstitutionMatrix.cpp:20-71):
 * score/index are [n][n], n = kalph^span, rows sorted by descending score, ties in enumeration order. */
int mmgpu_host_score_matrix(const int16_t *submat, int alphabet, int span, int16_t *score, uint32_t *index);
/* the same into rows of row_stride elements (ScoreMatrix::rowSize, ExtendedSubstitutionMatrix.cpp:24-25); the padding elements
 * behind each row are left to the caller.  What the patched Prefiltering constructor calls in place of calcScoreMatrix: a stable
 * counting sort per row instead of std::stable_sort (0.45 s -> 0.05 s for the 8000 x 8000 3-mer table on 32 threads). */
int mmgpu_host_score_matrix_rows(const int16_t *submat, int alphabet, int span, size_t row_stride, int16_t *score, uint32_t *index);
/* IndexTable over numeric targets, masking off (IndexTable.h:135-191,350-403; IndexBuilder.cpp:118-166,226-270):
 * one entry (seqId, first position) per distinct k-mer of a target whose window has no X and whose self score
 * is >= kmer_thr.  offsets has (alphabet-1)^k + 1 elements.  Call with ids == pos == NULL to get the entry
 * count in *n_entries (offsets is filled), then again with arrays of that size. */
int mmgpu_host_index_build(const uint8_t *residues, const uint64_t *seq_offsets, uint32_t n_targets,
                           const int16_t *kmer_submat, int alphabet, int kmer_size, int spaced, int kmer_thr,
                           uint64_t *offsets, uint32_t *ids, uint16_t *pos, uint64_t *n_entries);

/* Everything QueryMatcher's constructor receives that lives in memory (Prefiltering.cpp:826-842). */
typedef struct {
    int kmer_size;              /* 6 or 7 with the similar-k-mer tables; 4..15 for exact k-mer matching (score3 == NULL) */
    int alphabet;               /* subMat->alphabetSize, 21 */
    int spaced;                 /* spaced k-mer pattern of Sequence.h:24-27 */
    const int16_t *score3;      /* ScoreMatrix::score of _3merSubMatrix */
    const uint32_t *index3;     /* ScoreMatrix::index */
    size_t row3;                /* ScoreMatrix::rowSize (elements per row incl. SIMD padding) */
    const int16_t *score2;      /* _2merSubMatrix (needed for kmer_size 7 only, may be NULL otherwise) */
    const uint32_t *index2;
    size_t row2;
    const uint64_t *offsets;    /* IndexTable::getOffsets(), (alphabet-1)^k + 1 */
    const uint32_t *entry_ids;  /* IndexEntryLocal::seqId    } either these two arrays ...               */
    const uint16_t *entry_pos;  /* IndexEntryLocal::position_j }                                          */
    const void *entries6;       /* ... or IndexTable::getEntries(): packed 6-byte IndexEntryLocal records */
    uint64_t n_entries;
    const int8_t *ungapped_mat; /* ungappedSubMat as int8, alphabet x alphabet (blosum62, bit factor 2) */
    int kmer_alphabet;          /* mmgpu_pf_load_index only: IndexTable::getAlphabetSize() if it is not alphabet - 1 (0 = alphabet - 1).  A
                                 * profile TARGET database is indexed over the full alphabet (Prefiltering.cpp:560-563: X is a letter of
                                 * the k-mer index, 21^k offsets), its queries match exactly; windows containing X are skipped as ever */
} mmgpu_pf_index;
/* Copies the tables into HBM.  The SequenceLookup the ungapped scorer reads is the database given to
 * mmgpu_load_targets (call that first, with the - possibly masked - SequenceLookup residues). */
int mmgpu_pf_load_index(mmgpu_ctx *ctx, const mmgpu_pf_index *index);

/* Device-side index construction (GPU-resident DB): IndexBuilder::fillDatabase with masking
 * off (IndexBuilder.cpp:118-166,226-270) over the targets made resident by mmgpu_load_targets - the same index
 * mmgpu_host_index_build returns, built in HBM.  `tables` supplies the score matrices and the ungapped matrix as for
 * mmgpu_pf_load_index (its offsets / entries fields are ignored); kmer_submat is BaseMatrix::subMatrix of the k-mer
 * matrix (short, alphabet x alphabet), kmer_thr the k-mer threshold that also gates which target k-mers are indexed
 * (IndexTable.h:146-154). */
int mmgpu_pf_build_index(mmgpu_ctx *ctx, const mmgpu_pf_index *tables, const int16_t *kmer_submat, int kmer_thr);
/* optional: loads the kernels' code objects now instead of at their first launch (0.1 - 0.2 s otherwise paid inside the first
 * prefilter block / alignment batch of a process); thread-safe with respect to other calls on the context */
int mmgpu_warmup(mmgpu_ctx *ctx);

/* tantan repeat masking of the resident targets, for the prefilter only (the masking step of IndexBuilder::fillDatabase,
 * IndexBuilder.cpp:148 -> Masker::maskSequence with maskTantan, Masker.cpp:14-57 -> tantan::maskSequences, lib/tantan/tantan.cpp
 * :469-487, maxRepeatOffset 50, repeatProb 0.005, repeatEndProb 0.05, decay 0.9, no gaps): call it after mmgpu_load_targets with
 * the UNMASKED residues and before mmgpu_pf_build_index; the k-mer index and the ungapped scorer then see the masked letters
 * (replaced by mask_letter = X), the alignment kernels the original ones - one resident database serves both stages.
 * likelihood_ratios: alphabet x alphabet doubles, probMatrix[i][j] / (pBack[i] * pBack[j]) of the k-mer matrix (ProbabilityMatrix,
 * BaseMatrix.h:83-101); min_mask_prob: --mask-prob as the reference passes it (its float, widened).  The repeat probabilities
 * are computed as the reference's AVX2 + FMA build computes them (tantan_kernel.hip; tests/test_tantan.py).
 * likelihood_ratios == NULL takes the masked view back: the prefilter reads the residues as loaded again, the index over the
 * masked ones is dropped (a resident database that serves a --mask 0 run after a --mask 1 run). */
int mmgpu_pf_mask_targets(mmgpu_ctx *ctx, const double *likelihood_ratios, int alphabet, double min_mask_prob, int mask_letter,
                          uint64_t *n_masked /* may be NULL */);
/* test hook: the prefilter's view of the resident targets (masked if mmgpu_pf_mask_targets ran) in the caller's layout */
int mmgpu_pf_debug_masked_targets(mmgpu_ctx *ctx, const uint64_t *offsets, uint32_t n_targets, uint8_t *residues);
/* test hook: copies the resident index back in the host builder's layout (any pointer may be NULL) */
int mmgpu_pf_debug_index(mmgpu_ctx *ctx, uint64_t *offsets, uint32_t *ids, uint16_t *pos, uint64_t *n_entries);

/* ---- persisted device layout -------------------------------------------------------------------------
 * The reference prepares a database for its GPU path ahead of time (`makepaddedseqdb`, src/util/makepaddedseqdb.cpp) and keeps
 * precomputed prefilter indexes on disk (`createindex`; src/prefiltering/PrefilteringIndexReader.cpp reads them back).  The
 * counterpart: ONE file with what a context has resident, in the layout it has on the device - the targets (4-byte aligned
 * residues, offsets, lengths), the prefilter's masked view when mmgpu_pf_mask_targets ran, and the k-mer index (offsets + entries)
 * when index_fingerprint != 0.  mmgpu_db_load brings a context to the same state without the host touching a sequence (no
 * SequenceLookup fill, no masking, no index build); the similar-k-mer score tables and the ungapped matrix are the caller's, as
 * for mmgpu_pf_build_index (`tables`: offsets / entries fields ignored).  The two fingerprints are the caller's: of the source
 * database (the drop-in hashes the keys and lengths of the DBReader), and of everything else the index depends on (k, pattern,
 * k-mer threshold, matrices, mask parameters).  A file that does not match - or is not there - is answered MMGPU_ERR_STATE and
 * the caller builds as ever (and may save).  index_fingerprint == 0: save / load the targets only (the alignment module's view). */
typedef struct mmgpu_db_info {
    uint64_t source_fingerprint, index_fingerprint;
    uint32_t n_targets, alphabet;
    uint64_t total_residues;
    int32_t has_masked_view, has_index, kmer_size, spaced;
    uint64_t n_entries, file_bytes;
} mmgpu_db_info;
int mmgpu_db_save(mmgpu_ctx *ctx, const char *path, uint64_t source_fingerprint, uint64_t index_fingerprint);
int mmgpu_db_probe(const char *path, mmgpu_db_info *info);      /* the header alone: no device needed */
int mmgpu_db_load(mmgpu_ctx *ctx, const char *path, uint64_t source_fingerprint, uint64_t index_fingerprint,
                  const mmgpu_pf_index *tables /* may be NULL when index_fingerprint == 0 */);

/* limits of the prefilter entry points: calls beyond them return MMGPU_ERR_UNSUPPORTED and the host keeps its CPU path */
#define MMGPU_PF_MAX_HITS 131072    /* max_hits (--max-seqs); above 4096 the final sort of a list runs in HBM instead of LDS */
#define MMGPU_PF_MAX_FUSED_HITS 4096 /* ... except sharded (exchange) batches, which stay at 4096; mmgpu_sw_prepare_from_pf takes lists up to 16384 */
#define MMGPU_SW_MAX_FUSED_LIST 16384
#define MMGPU_PF_MAX_TARGETS 8388608 /* resident targets a context's prefilter can index (2048 bins of 4096 ids); larger databases: several
                                      * contexts (mmgpu_init_multi - a repeated device id puts them on one device), merged lists = unsplit */
#define MMGPU_PF_MAX_SEQ_LEN 65536  /* Parameters.h:271; queries / candidates of 32768 residues or more: MMGPU_PF_LONG_SEQ */

typedef struct {
    int kmer_thr;            /* Prefiltering::getKmerThreshold */
    uint32_t max_hits;       /* maxResListLen (--max-seqs); min(., dbSize) is applied like QueryMatcher.cpp:47 */
    uint32_t min_diag_score; /* --min-ungapped-score (15); must be >= 1 (0 is accepted with kmer_score, where it equals 1) */
    uint32_t ref_bins;       /* the CacheFriendlyOperations<N> the CPU run would use (QueryMatcher.cpp:460-488);
                                only decides which of several equal-score hits survive the max_hits cut.
                                0 = derive from dbSize and this host's L2 size like the reference */
    uint32_t exact_kmer;     /* takeOnlyBestKmer (--exact-kmer-matching; always on in nucleotide searches, Search.cpp:186): every
                                window matches its own k-mer only, kmer_thr is not read (QueryMatcher.cpp:279-282) */
    uint32_t nucleotide;     /* the target database is nucleotide: matchQuery's isNucleotide branch (QueryMatcher.cpp:147-177) - of
                                several saturated (>= 255) diagonals of one target the one with the best exact score is kept */
    uint32_t kmer_score;     /* --diag-score 0 (Prefiltering::diagonalScoring == false): no ungapped scoring - the prefilter score of a
                                target is the number of its double k-mer matches, saturating at 255 (findDuplicates with
                                computeTotalScore, CacheFriendlyOperations.cpp:218-239; QueryMatcher.cpp:215-232, getResult<KMER_SCORE>);
                                the self hit scores 255.  Queries on the databaseHits overflow path, or with 500 000 or more
                                elements, come back MMGPU_PF_OVERFLOW / MMGPU_PF_SAT_TIE (the host runs the reference for them);
                                not with sharded databases, profile queries or nucleotide searches */
} mmgpu_pf_params;

typedef struct {
    const uint8_t *q;        /* Sequence::numSequence */
    uint32_t qlen;
    const float *comp_bias;  /* QueryMatcher::compositionBias (mmgpu_host_comp_bias over the k-mer matrix), NULL = 0 */
    uint32_t identity_id;    /* targetSeqId of Prefiltering.cpp:855-868, UINT32_MAX = none */
    /* profile query (DBTYPE_HMM_PROFILE; Prefiltering.cpp:832-834 hands Sequence::profile_matrix to the matcher): q = numSequence
     * (only the X test of the k-mer window reads it), comp_bias is ignored (QueryMatcher.cpp:110-114), and the three
     * arrays below are what Sequence::mapProfile left in the Sequence (Sequence.cpp:301-352).  All NULL: sequence query. */
    const int16_t *profile_score;   /* Sequence::profile_score: [qlen][profile_row], 20 scores per position sorted descending */
    const uint32_t *profile_index;  /* Sequence::profile_index: the letters in that order */
    uint32_t profile_row;           /* Sequence::profile_row_size */
    const int8_t *profile;          /* Sequence::getAlignmentProfile(): [20][qlen] (UngappedAlignment::createProfile :405-411) */
} mmgpu_pf_query;

/* == hit_t (QueryMatcher.h:33-49) */
typedef struct {
    uint32_t id;
    int32_t score;
    uint16_t diagonal;
    uint16_t reserved;
} mmgpu_pf_hit;

#define MMGPU_PF_OK 0
#define MMGPU_PF_OVERFLOW 1 /* the query needs more than 62 flushes of the reference's databaseHits buffer
                               (QueryMatcher.cpp:310-346; up to 62 are emulated on the device): not computed here,
                               the host must run QueryMatcher::matchQuery for this query */
#define MMGPU_PF_SAT_TIE 3  /* nucleotide searches: two saturated diagonals of one target tie on the exact score in a query with more than
                               16 saturated elements - the reference's choice then depends on the element order its std::sort by id
                               left (QueryMatcher.cpp:154; up to 16 elements libstdc++ sorts by insertion, i.e. stably, and the device
                               makes the same choice); or the query took the databaseHits overflow path.  Any search: the query's double-diagonal candidates number
                               max(1M, dbSize) / 2 or more, where the reference may take its unsorted branch (unstable std::sort, no
                               rescoring, QueryMatcher.cpp:188,204-214).  Not decided on the device: the host runs
                               QueryMatcher::matchQuery for this query */
#define MMGPU_PF_LONG_SEQ 2 /* sequences of 32768 residues or more are scored on the device since round 6 (UngappedAlignment::
                               computeLongScore: every 65536-shift of the 16-bit diagonal, UngappedAlignment.cpp:295-312, and the
                               batches of eight elements of scoreDiagonalAndUpdateHits, :187-293) - this status is left for: such a
                               query in a sharded run's shard, a nucleotide search or --diag-score 0 (declined on the host); a query
                               with candidates on such targets that is on the databaseHits overflow path, or has more than 4096
                               candidates on the diagonals of those targets.  The host must run QueryMatcher::matchQuery for it
                               (a sharded run re-runs it against the unsplit database when it holds one) */

#define MMGPU_PF_SHARD_INEXACT 4 /* sharded runs (mmgpu_multi_pf_fetch): an element of the query's merged list took the reference's
                               overflow path on a shard, or was not scored on the device - the tie order at the cut is not the
                               unsplit run's; the caller re-runs the query unsplit (or on the host) */

typedef struct {
    uint64_t db_matches;     /* statistics_t::dbMatches */
    uint64_t kmer_list_len;  /* sum over positions of similar k-mers (kmersPerPos * L) */
    uint32_t double_hits;    /* elements after keepMaxScoreElementOnly with count >= min_diag_score; with kmer_score
                                statistics_t::doubleMatches, the sum of the match counts (QueryMatcher.cpp:366-385) */
    uint32_t diag_thr;       /* diagonalThr; bit 31 = scoreIsTruncated */
} mmgpu_pf_qstat;

/* One call = the query loop of Prefiltering::runSplit (:848-917) up to the hit_t list, for nq queries.
 * hits has nq * hit_stride entries (hit_stride >= min(max_hits, dbSize)); counts/status have nq entries. */
typedef struct mmgpu_pf_batch_t mmgpu_pf_batch_t;
int mmgpu_pf_batch(mmgpu_ctx *ctx, const mmgpu_pf_params *params, const mmgpu_pf_query *queries, uint32_t n_queries,
                   mmgpu_pf_hit *hits, uint32_t hit_stride, uint32_t *counts, int32_t *status);
/* split form: prepare = host prep + H2D; run = all kernels (contains two small D2H size read-backs);
 * fetch = D2H of the results */
int mmgpu_pf_prepare(mmgpu_ctx *ctx, const mmgpu_pf_params *params, const mmgpu_pf_query *queries,
                     uint32_t n_queries, mmgpu_pf_batch_t **batch);
int mmgpu_pf_run(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch);
int mmgpu_pf_fetch(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, mmgpu_pf_hit *hits, uint32_t hit_stride,
                   uint32_t *counts, int32_t *status, mmgpu_pf_qstat *stats /* may be NULL */);
/* Device-resident hand-over for multi-GPU runs: copies the batch's hit lists into caller-owned DEVICE memory
 * (e.g. a torch tensor that is then all-gathered over RCCL): d_hits [nq][hit_stride] mmgpu_pf_hit, d_counts [nq]. */
int mmgpu_pf_fetch_device(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, void *d_hits, uint32_t hit_stride, void *d_counts);
/* Device analogue of Prefiltering::mergeTargetSplits (Prefiltering.cpp:412-526) on gathered lists:
 * d_hits [n_splits][nq][stride], d_counts [n_splits][nq] (device pointers); ids of split s get id_offsets[s] added
 * (dbFrom, Prefiltering.cpp:879-881); every query's lists are concatenated and sorted by (|score| desc, id asc).
 * d_out_hits [nq][n_splits*stride], d_out_counts [nq].  No truncation, like the reference (which instead lowers
 * maxResListLen per split, Prefiltering.cpp:391-394 - pass that value as max_hits of each split's batch). */
int mmgpu_pf_merge_splits(mmgpu_ctx *ctx, const void *d_hits, const void *d_counts, uint32_t n_splits, uint32_t n_queries,
                          uint32_t stride, const uint32_t *id_offsets, void *d_out_hits, void *d_out_counts);
/* ---- multi-GPU runs whose merged result equals the UNSPLIT run (one process per device) ----------
 * The reference's own target-split mode shortens the per-split lists (Prefiltering.cpp:391-394), which changes the
 * result against a single-split run; mmgpu_pf_merge_splits above reproduces that mode.  The calls below instead
 * reproduce the single-split run exactly: a device that holds one shard selects its candidates by the unsplit run's
 * own total order - 8-bit diagonal score, CacheFriendlyOperations bin of the GLOBAL target id, arrival order (which
 * similar-k-mer list emitted the element: a function of the query alone, then target id) - and hands them over as
 * exchange records; after ONE all-gather of the records (RCCL) every device redoes the reference's threshold /
 * truncation / rescoring / final sort over the union.  Not covered: queries on the databaseHits overflow path
 * (QueryMatcher.cpp:310-346) - the flush points depend on the size of the database a process holds, so the reference's
 * own result differs between split and unsplit runs there; such queries are flagged. */
typedef struct {
    uint32_t n_shards, shard;         /* this device's shard */
    uint32_t global_db_size;          /* targets of the whole database */
    const uint32_t *global_ids;       /* [targets of this shard] local -> global id, ascending */
    const uint32_t *shard_of;         /* [global_db_size] mmgpu_host_partition_targets */
    const uint32_t *local_id;         /* [global_db_size] */
} mmgpu_pf_shard;
/* after mmgpu_load_targets (which resets it); NULL = the device holds the whole database.  While a shard is set,
 * mmgpu_pf_prepare batches are exchange batches: max_hits / ref_bins refer to the whole database, the results are read
 * with mmgpu_pf_fetch_exchange (mmgpu_pf_fetch / mmgpu_sw_prepare_from_pf refuse them). */
int mmgpu_pf_set_shard(mmgpu_ctx *ctx, const mmgpu_pf_shard *shard);
typedef struct {
    uint32_t id;        /* GLOBAL target id */
    uint32_t score;     /* exact ungapped score of the target's best diagonal */
    uint16_t diagonal;
    uint16_t flags;     /* MMGPU_PF_X_* */
    uint32_t order;     /* ordinal of the emitting similar-k-mer list in the query's list stream */
} mmgpu_pf_xhit;
#define MMGPU_PF_X_INEXACT_ORDER 1 /* query on the overflow path in this shard: `order` is not shard independent */
#define MMGPU_PF_X_IDENTITY 2      /* the element of the query's own target (takes part in the score histogram only) */
/* device -> device copy of an exchange batch's records: d_xhits [nq][stride] mmgpu_pf_xhit, d_counts [nq] uint32 */
int mmgpu_pf_fetch_exchange(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, void *d_xhits, uint32_t stride, void *d_counts);
/* d_xhits [n_shards][nq][stride], d_counts [n_shards][nq] (device, all-gathered); `batch` = this device's batch of the
 * same queries (supplies max_hits, min_diag_score, ref_bins, self scores); identity_global [nq] (host, may be NULL) =
 * global id of each query's own target (UINT32_MAX none).  d_out_hits [nq][out_stride] mmgpu_pf_hit with GLOBAL ids in
 * the reference's final order, d_out_counts [nq], d_out_flags [nq] (may be NULL; bit 0: inexact tie order, see above). */
int mmgpu_pf_merge_exchange(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, const void *d_xhits, const void *d_counts, uint32_t n_shards,
                            uint32_t stride, const uint32_t *identity_global, void *d_out_hits, uint32_t out_stride,
                            void *d_out_counts, void *d_out_flags);
/* Prefiltering::mergeTargetSplits (Prefiltering.cpp:412-526) never hands a query back, and neither does a sharded run here: the
 * queries whose merged list is flagged (MMGPU_PF_X_INEXACT_ORDER took part, or a shard declined the query) run once more against
 * `full`, a context that holds the WHOLE database (same ids as the global ids) with its own index - 3 GB for the million targets of
 * the headline, beside the shard - and that list replaces the merged one in the batch (hits, count, flag cleared) before the lists
 * are localized and aligned.  params / queries: what the batch was prepared with (identity_id = the GLOBAL id).  *n_redone = queries
 * re-run, *n_left = those of them the unsplit run itself leaves to the host (MMGPU_PF_OVERFLOW beyond 62 flushes, MMGPU_PF_LONG_SEQ ...:
 * their flag stays set).  Reads the flags back (synchronises the context's stream); without a flagged query nothing else happens. */
int mmgpu_pf_exchange_redo_unsplit(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, mmgpu_ctx *full, const mmgpu_pf_params *params,
                                   const mmgpu_pf_query *queries, uint32_t n_queries, uint32_t *n_redone, uint32_t *n_left);
/* The part of merged lists (global ids) this device owns, as lists of local ids in list order: d_local_hits [nq][stride],
 * d_local_counts [nq], d_local_slot [nq][stride] (position of each kept hit in its merged list).  Feeds
 * mmgpu_sw_prepare_from_lists: every (query, target) pair is aligned on the device that holds the target. */
int mmgpu_pf_localize_lists(mmgpu_ctx *ctx, const void *d_hits, const void *d_counts, uint32_t n_queries, uint32_t stride,
                            void *d_local_hits, void *d_local_counts, void *d_local_slot);

/* ---- the communicator of a multi-GPU run, owned by the library -------------------------------
 * The shards of ONE target database live in N contexts (one per GPU); the two exchange steps of a query batch - the
 * all-gather of the per-shard hit lists and the gather of the alignment records - are RCCL collectives over xGMI that the
 * library enqueues on the context's stream, with no host synchronisation.  This replaces what the reference does between
 * MPI ranks through files (Prefiltering::runMpiSplits + mergeTargetSplits, Prefiltering.cpp:605-689,412-526).
 *   one process per GPU : rank 0 calls mmgpu_comm_unique_id, the MMGPU_COMM_ID_BYTES bytes reach the other ranks by the
 *                         host's own means (MPI_Bcast in an MPI build of mmseqs, the launcher's store under torchrun),
 *                         every rank calls mmgpu_comm_init_rank on its context (collective: all ranks must call it);
 *   one process, N GPUs : mmgpu_init_multi below creates the contexts and the communicator together.
 * RCCL is loaded on first use (dlopen librccl.so.1, or $MMGPU_RCCL_LIB); single-GPU callers never load it. */
#define MMGPU_COMM_ID_BYTES 128
int mmgpu_comm_unique_id(uint8_t *id /* MMGPU_COMM_ID_BYTES */);
int mmgpu_comm_init_rank(mmgpu_ctx *ctx, const uint8_t *id, int rank, int n_ranks);
int mmgpu_comm_info(mmgpu_ctx *ctx, int *rank, int *n_ranks, char *transport, int transport_cap);   /* "rccl" | "copy" | "none" */
void mmgpu_comm_destroy(mmgpu_ctx *ctx);      /* also done by mmgpu_destroy */
/* Exchange step 1, for an exchange batch that has been run on every rank (same queries, same parameters): all-gather of the
 * records over the context's communicator (a context without one is its own single rank) and the merge kernel - threshold,
 * truncation and final order of the UNSPLIT run.  Everything is enqueued on the context's stream.  The merged lists stay
 * in device buffers owned by the batch (valid until it is run again or freed): *d_hits [nq][*stride] mmgpu_pf_hit with GLOBAL
 * ids, *d_counts [nq] uint32, *d_flags [nq] uint32 (bit 0: MMGPU_PF_X_INEXACT_ORDER took part); any of them may be NULL.
 * identity_global [nq] (host, may be NULL) = global id of each query's own target.  The communicator's rank count must equal
 * the shard description's n_shards. */
int mmgpu_pf_exchange_merge(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, const uint32_t *identity_global, const void **d_hits,
                            const void **d_counts, const void **d_flags, uint32_t *stride);
/* Exchange step 2.  mmgpu_sw_prepare_owned: the alignment batch of the pairs of the merged lists (left in `merged` by
 * mmgpu_pf_exchange_merge) whose target this context's shard holds; queries as for mmgpu_sw_prepare_from_pf.  After
 * mmgpu_sw_run, mmgpu_sw_gather_owned all-gathers the owned records (compacted, 32 bytes each) and scatters them into
 * *d_full: mmgpu_sw_hit [nq * stride] in merged-list order (slot q * stride + k = hit k of query q's merged list; zero where
 * the list ends), identical on every rank; *d_status: uint32 [2] = records gathered, ranks whose send buffer overflowed
 * (the deal by length bucket is even: the buffer holds 1.5 x the even share; MMGPU_SW_GATHER_DENSE=1 sizes it for the worst
 * case).  mmgpu_sw_fetch_owned copies the array to the host and fails on overflow. */
int mmgpu_sw_prepare_owned(mmgpu_ctx *ctx, const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t n_queries,
                           int mode, mmgpu_pf_batch_t *merged, mmgpu_sw_batch_t **batch);
int mmgpu_sw_gather_owned(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, const void **d_full, const void **d_status);
int mmgpu_sw_fetch_owned(mmgpu_ctx *ctx, mmgpu_sw_batch_t *batch, mmgpu_sw_hit *out, uint32_t *records);

/* ---- one process, several GPUs (the form the patched `mmseqs` binary uses) ----------------------------------------------
 * mmgpu_init_multi creates one context per device id, each with a stream of its own, and the communicator over them
 * (ncclCommInitAll).  A repeated device id, or MMGPU_MULTI_TRANSPORT=copy, selects the "copy" transport instead: the
 * all-gathers become device-to-device copies ordered by events (hipMemcpyPeerAsync between devices) - it lets several shards
 * share one GPU (tests on a 1-GPU box) and stands in where RCCL is missing.  The mmgpu_multi_* calls drive every step over
 * all contexts from the calling thread; each step only enqueues work (the collectives of a step inside one RCCL group). */
typedef struct mmgpu_multi mmgpu_multi;
typedef struct mmgpu_multi_pf_batch mmgpu_multi_pf_batch;
int mmgpu_init_multi(mmgpu_multi **multi, const int *device_ids, int n_devices);
void mmgpu_destroy_multi(mmgpu_multi *multi);
int mmgpu_multi_size(mmgpu_multi *multi);
mmgpu_ctx *mmgpu_multi_ctx(mmgpu_multi *multi, int i);      /* context of shard i (owned by `multi`) */
int mmgpu_multi_synchronize(mmgpu_multi *multi);
/* the database dealt to the contexts by length bucket (mmgpu_host_partition_targets), shard descriptions set */
int mmgpu_multi_load_targets(mmgpu_multi *multi, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int alphabet);
/* tantan masking of every shard's resident targets (mmgpu_pf_mask_targets per context; n_masked = the sum): between
 * mmgpu_multi_load_targets and mmgpu_multi_pf_build_index */
int mmgpu_multi_pf_mask_targets(mmgpu_multi *multi, const double *likelihood_ratios, int alphabet, double min_mask_prob,
                                int mask_letter, uint64_t *n_masked /* may be NULL */);
/* every shard's k-mer index built on its device (mmgpu_pf_build_index) */
int mmgpu_multi_pf_build_index(mmgpu_multi *multi, const mmgpu_pf_index *index, const int16_t *kmer_submat, int kmer_thr);
/* One prefilter batch over all shards whose merged lists EQUAL the unsplit run's.  queries[i].identity_id is the GLOBAL id.
 * run = per-shard prefilter -> all-gather of the exchange records -> merge kernel on every context (enqueues only);
 * fetch = merged lists from context 0 (global ids), status MMGPU_PF_OK, MMGPU_PF_SHARD_INEXACT (count 0: re-run unsplit) or what a
 * shard's own run decided for the query (MMGPU_PF_LONG_SEQ, MMGPU_PF_OVERFLOW: count 0, the host runs the reference for it). */
int mmgpu_multi_pf_prepare(mmgpu_multi *multi, const mmgpu_pf_params *params, const mmgpu_pf_query *queries, uint32_t n_queries,
                           mmgpu_multi_pf_batch **batch);
int mmgpu_multi_pf_run(mmgpu_multi *multi, mmgpu_multi_pf_batch *batch);
int mmgpu_multi_pf_fetch(mmgpu_multi *multi, mmgpu_multi_pf_batch *batch, mmgpu_pf_hit *hits, uint32_t hit_stride, uint32_t *counts,
                         int32_t *status);
/* When the first device has room for it beside its shard (three times 12 bytes per residue + 3 GB free), mmgpu_multi_load_targets /
 * _pf_mask_targets / _pf_build_index also keep the WHOLE database in a context of its own there (mmgpu_multi_has_unsplit), and the
 * queries a run flags MMGPU_PF_SHARD_INEXACT are run once more against it where the merged lists are first read (mmgpu_multi_pf_fetch,
 * mmgpu_multi_sw_from_pf): no query is handed back for the way it was sharded (Prefiltering::mergeTargetSplits hands none back,
 * Prefiltering.cpp:412-526).  mmgpu_multi_pf_redone: how many queries of the batch's last run took that path, and how many of them the
 * unsplit run itself leaves to the host. */
int mmgpu_multi_has_unsplit(mmgpu_multi *multi);
int mmgpu_multi_pf_redone(mmgpu_multi_pf_batch *batch, uint32_t *n_redone, uint32_t *n_left);
uint32_t mmgpu_multi_pf_stride(mmgpu_multi_pf_batch *batch);   /* slots per query of the merged lists, 0 before the first run */
void mmgpu_multi_pf_free(mmgpu_multi *multi, mmgpu_multi_pf_batch *batch);
/* Alignment of the merged lists of a batch that has been run: every context aligns the pairs whose target its shard holds
 * (mmgpu_sw_prepare_owned + mmgpu_sw_run), the records are gathered over the communicator, out [n_queries * stride] receives
 * them in merged-list order.  queries as for mmgpu_sw_prepare_from_pf.  cells / kernel_ms (may be NULL): forward cells of all
 * shards, alignment-kernel time of the slowest context. */
int mmgpu_multi_sw_from_pf(mmgpu_multi *multi, const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t n_queries,
                           int mode, mmgpu_multi_pf_batch *batch, mmgpu_sw_hit *out, uint64_t *cells, float *kernel_ms);

/* milliseconds per stage of the last run (HIP events on the context's stream; synchronises):
 * ms[0] similar k-mers + index lists, ms[1] gather + bin split, ms[2] double-diagonal replay + ungapped scoring + best
 * element per target (bins with <= 64 candidates, i.e. nearly all), ms[3] ungapped scoring of larger bins, ms[4] best
 * element per target of larger bins, ms[5] top-N select, ms[6] whole run (including the two host read-backs) */
int mmgpu_pf_stage_ms(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, float ms[7]);
/* ungapped diagonal cells scored by the last run (sum of overlap lengths of the double-diagonal candidates) */
int mmgpu_pf_last_cells(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, uint64_t *cells, uint64_t *candidates);
/* Stage dumps for the parity tests: copies one intermediate buffer of the last run to the host.
 * *bytes = size of the buffer; copies min(cap, *bytes).  Layouts are those of mmseqs2_amd/csrc/mmgpu_internal.h. */
#define MMGPU_PF_DBG_NSIM 0       /* uint32[n_pos]   similar k-mers per window                              */
#define MMGPU_PF_DBG_LIST_BASE 1  /* uint32[n_pos+1]                                                         */
#define MMGPU_PF_DBG_LISTS 2      /* {start,len,lprefix,pos} uint32 x 4 per similar k-mer                     */
#define MMGPU_PF_DBG_PEB 3        /* uint32[n_pos+1] arrival index of each window's first entry (per query)   */
#define MMGPU_PF_DBG_SPLIT 4      /* uint64[n_tiles][tile] entries grouped by bin                             */
#define MMGPU_PF_DBG_BIN_OFF 5    /* uint16[n_tiles][bins+1]                                                  */
#define MMGPU_PF_DBG_CAND_BASE 6  /* uint32[nq*bins+1]                                                        */
#define MMGPU_PF_DBG_SURV 7       /* {id,arr,score,diag|pad} 16 B records, query q at cand_base[q*bins]       */
#define MMGPU_PF_DBG_SURV_COUNT 8 /* uint32[nq]                                                               */
#define MMGPU_PF_DBG_BINS 9       /* uint32[3]: device bins, reference bins, entries per tile                 */
int mmgpu_pf_debug_fetch(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch, int what, void *dst, size_t cap, size_t *bytes);
void mmgpu_pf_free(mmgpu_ctx *ctx, mmgpu_pf_batch_t *batch);

#ifdef __cplusplus
}
#endif
#endif /* MMGPU_H */
