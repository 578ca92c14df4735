// Internal declarations shared by the HIP translation units of libmmgpu (not part of the C-ABI).
#ifndef MMGPU_INTERNAL_H
#define MMGPU_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <array>
#include <string>
#include <map>
#include <mutex>
#include <chrono>
#include <memory>
#include <vector>
#include <thread>
#include <atomic>

#include "../../include/mmgpu.h"
#include "nucl_core.h"

namespace mmgpu {

// Target database resident in HBM.  Every sequence starts on a 4-byte boundary so the group-head lane can
// fetch four residues per global_load_dword; `off4` is the start in dwords.
struct DeviceDb {
    uint8_t *res = nullptr;      // residues, 4-byte aligned starts, tail slack of max_len + 64 bytes
    uint32_t *off4 = nullptr;    // [n] start of sequence i in units of 4 bytes
    uint32_t *len = nullptr;     // [n]
    uint32_t n = 0;
    uint32_t max_len = 0;
    uint64_t total_residues = 0;
    int alphabet = 0;
    size_t res_bytes = 0;        // allocated size of `res`
};

// One workgroup's share of a batch: hits [hit_begin, hit_end) of one query (already sorted by target length).
struct SwJob {
    uint32_t query;
    uint32_t hit_begin;
    uint32_t hit_end;
    uint32_t shape;   // low 8 bits: tile shape (rows per lane - 1, + 32 for multi-tile); high 24: multi-tile scratch slot
};

// Everything a Smith-Waterman launch needs, device pointers only.
struct SwLaunch {
    const SwJob *jobs;
    uint32_t n_jobs;
    // queries of the batch
    const uint8_t *q_res;     // concatenated numeric residues
    const int8_t *q_cb;       // concatenated rounded composition bias (zeros when absent)
    const uint32_t *q_off;    // [nq+1]
    const int32_t *q_bias;    // [nq] ssw_init bias (StripedSmithWaterman.cpp:1397-1406), decides `word`
    const int32_t *q_minstart; // [nq] reverse scan only for pairs with score >= this
    // profile queries (ssw_init with DBTYPE_HMM_PROFILE): int8 [alphabet][qlen] per query, letter-major, concatenated;
    // q_prof_off[q] = byte offset of the query's block, 0xFFFFFFFF for a sequence query; q_prof_off null = none in the batch
    const int8_t *q_prof;
    const uint32_t *q_prof_off;
    // targets
    const uint8_t *t_res;
    const uint32_t *t_off4;
    const uint32_t *t_len;
    // hits (sorted order) and where each one's result goes
    const uint32_t *hit_target;
    const uint32_t *hit_out;  // index into out[]
    mmgpu_sw_hit *out;
    // reverse pass only: per-hit forward result lives in out[hit_out[h]]
    // scoring
    const int8_t *mat;        // alphabet*alphabet
    int alphabet;
    int gap_open, gap_extend;
    // multi-tile scratch (H/F boundary rows), [scratch slot][4 waves][4 groups][2][scratch_cols] x uint2.  Slots are a
    // pool sized by the workgroups that can be resident at once, not by the jobs of the batch: a workgroup claims a free
    // slot when it starts (scratch_busy, one flag per slot) and releases it when it ends.
    uint2 *scratch;
    uint32_t scratch_cols;
    uint32_t *scratch_busy;
    uint32_t scratch_slots;
    // fused hand-over from a prefilter batch: hits of query q live in slots [q * hit_stride, + q_hit_count[q])
    const uint32_t *q_hit_count;   // null for caller-supplied lists
    uint32_t hit_stride;
    // which pairs get a reverse scan (sw_rev_wanted, sw_kernel.hip): 0 = score >= q_minstart (MMGPU_SW_START); 1 = ... and word == 0
    // (MMGPU_SW_START_NOT_WORD: the start of an int16-range hit comes from the block aligner); 2 = the pairs flagged in rev_force,
    // indexed like out[] (mmgpu_sw_reverse_pairs).  rev_only: the forward results are in out[] already, only the reverse scan runs.
    int rev_mode = 0;
    int rev_only = 0;
    const uint8_t *rev_force = nullptr;
};

constexpr int SW_PF_MAX_LIST = 16384;  // longest list of the fused hand-over (sw_from_pf_kernel orders it in LDS, 4 bytes per slot; PF_MAX_HITS is the prefilter's own LDS limit)
// sw_from_pf_kernel's statistics (cells, pairs, longest target): one workgroup per query adds to slot blockIdx % SLOTS of
// [SLOTS][3] - 10 000 workgroups x 6 atomics on three words serialised at the L2 and were most of the kernel's 0.73 ms
constexpr int SW_FROM_PF_STAT_SLOTS = 64;

struct SwFromPfArgs {
    const mmgpu_pf_hit *pf_hits;   // [nq][pf_stride]
    uint32_t pf_stride;
    const uint32_t *hit_count;     // [nq]
    uint32_t stride;               // slots per query in the alignment batch
    const uint32_t *q_off;
    const uint32_t *t_len;
    uint32_t *hit_target, *hit_out;
    unsigned long long *cells, *pairs;      // slot 0 of [SW_FROM_PF_STAT_SLOTS][3] (cells, pairs, longest target)
    // copies owned by the alignment batch, so that it does not depend on the prefilter batch after this kernel:
    uint32_t *count_copy;          // [nq] list lengths (clipped to stride)
    uint32_t *slot_target;         // [nq * stride] target id of every result slot (list order), 0 beyond the list
};

constexpr int SW_REV_JOB_MAX = 1024;   // most hits of one reverse-scan job of a multi-tile query (sw_rev_multi_kernel)
hipError_t launch_sw_rev_multi(const SwLaunch &L, size_t lds_bytes, hipStream_t stream);
hipError_t launch_sw_from_pf(const SwFromPfArgs &A, uint32_t nq, hipStream_t stream);
// workgroups of the multi-tile kernel group that can be resident on the device at once (sizes the scratch pool)
uint32_t sw_multi_resident_blocks(size_t lds_bytes, bool both_passes, int compute_units);

// rows_per_lane in {8,16,24,32}: a 16-lane group covers 16*rows_per_lane query rows per tile.
// reverse = false: forward score/end scan; true: start-position scan over the reversed prefixes.
#ifndef MMGPU_SW_MAX_R
#define MMGPU_SW_MAX_R 28
#endif
constexpr int SW_MAX_R = MMGPU_SW_MAX_R;   // rows per lane of the largest tile (16 lanes x SW_MAX_R query rows); longer queries are cut into tiles
static_assert(SW_MAX_R >= 16 && SW_MAX_R <= 32, "multi-tile bodies exist for 8..32 rows per lane: a query cut into tiles gets at least SW_MAX_R / 2");
#ifndef MMGPU_SW_MIN_WAVES
#define MMGPU_SW_MIN_WAVES 2
#endif
constexpr int SW_MIN_WAVES = MMGPU_SW_MIN_WAVES;   // occupancy floor (waves per SIMD) the alignment kernels are compiled for
#ifndef MMGPU_SW_GROUPS
#define MMGPU_SW_GROUPS 4
#endif
// kernels per pass: tile shapes grouped by register need (sw_kernel.hip): R <= 12, R <= 24, single tiles of R >= 26, and - always
// the last group - the queries cut into several tiles (3: the last two in one kernel, rounds 1-2)
constexpr int SW_GROUPS = MMGPU_SW_GROUPS;
static_assert(SW_GROUPS == 3 || SW_GROUPS == 4, "three or four kernel groups");
int sw_shape_group(uint32_t shape);
hipError_t launch_sw(const SwLaunch &L, int group, size_t lds_bytes, bool both_passes, hipStream_t stream);
size_t sw_lds_bytes(int rows_per_lane, int alphabet);

// ---------------------------------------------------------------------------------------------------------
// prefilter (pf_kernels.hip)
constexpr int PF_T = 4096;             // arrival-ordered index entries per tile (2048: 22 % slower, more tiles)
constexpr int PF_IDS_PER_BIN = 4096;   // targets per replay bin (one 16 KB LDS state table per wavefront)
constexpr int PF_SAT_CAP = 1024;        // saturated elements per query exported in nucleotide mode (16 KB per query)
constexpr int PF_CAND0 = 32;            // candidates per (query, bin) kept in the dense array
constexpr int PF_QSTAGE = 2048;         // longest query whose residues the ungapped kernel stages in LDS
constexpr int PF_MAX_HITS = 4096;      // largest --max-seqs the select kernel sorts in LDS (larger lists: global scratch, PF_MAX_HITS_BIG)
constexpr int PF_MAX_HITS_BIG = 131072;   // == MMGPU_PF_MAX_HITS

struct PfList {        // index list of one similar k-mer of one query position
    uint32_t start;    // first entry in the index arrays
    uint32_t len;
    uint32_t lprefix;  // entries of earlier lists of the same position
    uint32_t pos;      // batch-global query position
};

struct PfCand {        // double-diagonal candidate / surviving element
    uint32_t id;
    uint32_t arr;      // arrival index of the emitting index entry within its query
    uint32_t score;    // exact ungapped score
    uint16_t diag;
    uint16_t pad;
};

// PfCand::score of an element whose count is NOT min(255, exact score): a target of 32768 residues or more in a full batch of
// UngappedAlignment::scoreDiagonalAndUpdateHits takes its count from another element's target (pf_long_kernel), while the rescoring of
// a saturated element (scoreSingleSequence) reads its own.  Bit 31 set: count in bits 30..23, exact score in bits 22..0.
__host__ __device__ inline uint32_t pf_el_count(uint32_t s) { return (s & 0x80000000u) ? ((s >> 23) & 0xFFu) : (s < 255u ? s : 255u); }
__host__ __device__ inline uint32_t pf_el_exact(uint32_t s) { return (s & 0x80000000u) ? (s & 0x7FFFFFu) : s; }

constexpr int PF_PROW = 32;            // bytes per position of the profile-query score rows (21 letters used)
constexpr int PF_PROF_LETTERS = 20;    // Sequence::PROFILE_AA_SIZE

struct PfKmerArgs {
    const uint8_t *q_res;      // batch residues, concatenated
    const int16_t *q_thr;      // per position: adjusted k-mer threshold, -1 = no window / X in window
    uint32_t n_pos;
    uint8_t pat[16];           // Sequence::aaPosInSpacedPattern
    uint32_t kalph, n3;
    uint32_t kbase;            // base of the k-mer index (pf_kmers_exact_kernel): kalph, or the full alphabet for profile targets
    const int16_t *s3;         // [n3][n3] ScoreMatrix::score of the 3-mer matrix (no padding columns)
    const uint32_t *i3;        // [n3][n3] ScoreMatrix::index
    const uint32_t *offsets;   // IndexTable::offsets, [kalph^k + 1]
    const uint32_t *nonempty;  // one bit per k-mer: list not empty (null: not used, the index is dense)
    const uint4 *cofs;         // compact offset table (pf_kernels.hip: 16-byte blocks of 12 k-mers), null = look-ups read `offsets`
    const uint16_t *cum3;      // [n3][cum_w]: cum3[row][k] = number of entries of the row with score >= score_min + k
    uint32_t cum_w;
    int32_t score_min;
    // k = 7 only: the 2-mer tables
    int k;
    const int16_t *s2;         // [n2][n2]
    const uint32_t *i2;
    const uint16_t *cum2;      // [n2][cum2_w]
    uint32_t cum2_w;
    int32_t score2_min;
    // profile queries (Sequence::profile_score / profile_index of the positions of profile queries, 20 per position,
    // sorted descending as Sequence::mapProfile leaves them): q_kind[gp] = 1 marks their positions (null: none in the batch)
    const uint8_t *q_kind;
    const int16_t *prof_score;    // [n_pos][20]
    const uint8_t *prof_letter;   // [n_pos][20]
    int exact;                 // takeOnlyBestKmer: every window matches its own k-mer only (QueryMatcher.cpp:279-282)
    const uint32_t *order;     // [n_pos] work order of the positions (launch_pf_order: grouped by the window's last 3-mer), null = as stored
    // count pass
    uint32_t *nsim;
    // emit pass
    const uint32_t *list_base; // [n_pos + 1]
    PfList *lists;
    uint32_t *pos_entries;
};

struct PfSplitArgs {
    const uint32_t *tile_q, *tile_idx;
    const uint32_t *q_off;            // [nq + 1]
    const uint32_t *q_entries;        // [nq]
    const uint32_t *pos_entry_base;   // [n_pos + 1], relative to the query
    const uint32_t *list_base;        // [n_pos + 1]
    const PfList *lists;
    const uint64_t *idx_entries;      // IndexEntryLocal re-packed to 8 bytes: seqId | position_j << 32 (one line per list)
    uint32_t bins;
    uint32_t *split;                  // [n_tiles][PF_T] (id >> log2 bins) | low diagonal byte << 12 | slot in tile << 20
    uint8_t *split_hi;                // [n_tiles][PF_T] high diagonal byte (same layout)
    uint16_t *bin_off;                // [n_tiles][bins + 1]
    uint32_t *bucket_count;           // [nq][bins]
};

struct PfDedupArgs {
    uint32_t n_queries, bins;         // queries of this launch: q_first .. q_first + n_queries - 1 of the batch
    uint32_t q_first;
    uint32_t cand_origin;             // cand/surv hold the entries cand_origin .. of the batch (stage chunks, pf_api.hip)
    const uint32_t *q_tile_base, *q_ntiles;
    const uint32_t *split;
    const uint8_t *split_hi;
    const uint16_t *bin_off;
    const uint32_t *cand_base;        // [nq * bins + 1]
    PfCand *cand, *surv;
    PfCand *cand_small;               // [nq * bins][PF_CAND0]
    uint32_t *surv_count;             // [nq]
    uint32_t *cand_count;             // [nq * bins]
    uint64_t *cell_counter;           // [nq] ungapped cells scored per query (statistics), may be null
    const uint32_t *q_off;
    const uint8_t *q_res;
    const int8_t *q_corr;             // UngappedAlignment::aaCorrectionScore per position
    const int8_t *mat;                // ungapped matrix, alphabet x alphabet
    int alphabet;
    // profile queries: UngappedAlignment::queryProfile rows (createProfile's profile branch, UngappedAlignment.cpp:405-411),
    // [n_pos][PF_PROW] int8, letter-indexed, filled for the positions of profile queries; q_isprof[q] picks the path
    const int8_t *q_rows;
    const uint8_t *q_isprof;
    // nucleotide searches (QueryMatcher.cpp:147-177): every bucket goes through pf_keepmax_nucl_kernel
    int nucl;
    uint32_t sort_cap;                // foundDiagonalsSize / 2: the branch is skipped for larger candidate sets (:146)
    uint32_t *q_ncand;                // [nq] double-diagonal candidates of the query (nucleotide mode only)
    // nucleotide mode: the saturated elements of every query, for the host's replay of the reference's std::sort (mmgpu_pf_fetch)
    PfCand *sat;                      // [nq][sat_cap], any order
    uint32_t *q_nsat;                 // [nq] their number (may exceed sat_cap: then the list is incomplete)
    uint32_t sat_cap;
    const uint8_t *t_res;
    const uint32_t *t_off4, *t_len;
    uint32_t min_diag_score;
    uint32_t *q_flags;                // [nq] bit 0: a candidate's target has >= 32768 residues (UngappedAlignment::computeLongScore,
                                      // UngappedAlignment.cpp:295-312): pf_long_kernel scores those and clears the bit; where it
                                      // does not run or gives up (sharded / overflow-path / nucleotide queries) the host runs the query
    uint32_t ref_bins;                // BINCOUNT of the reference's CacheFriendlyOperations: the order of its result array (pf_long_kernel)
    // work list of the buckets with more than 64 candidates (bucket index relative to the launch's first bucket), appended by the
    // replay kernel, walked by pf_ungapped_kernel / pf_keepmax_kernel; null: those kernels run one wavefront per bucket of the launch
    uint32_t *big_list, *big_count;
    // overflow emulation (null when the batch has no query on the overflow path)
    const uint32_t *q_nseg;           // [nq] databaseHits flushes of the query (0 = ordinary query)
    const uint32_t *seg_start;        // [nq][PF_MAX_SEG + 2] arrival index at which segment k starts
    // buckets the compact replay hands to the full-state one (more emitting targets than its table holds)
};

struct PfSelectArgs {
    uint32_t q_first, cand_origin;    // as in PfDedupArgs
    int kmer_score;                   // --diag-score 0: scores are match counts (no rescoring, self hit 255)
    const PfCand *surv;
    const uint32_t *cand_base;        // survivors of query q start at cand_base[q * bins]
    uint32_t bins;
    const uint32_t *surv_count;
    const uint32_t *q_identity;
    const int32_t *q_self_score;
    uint32_t max_hits, min_diag_score, ref_bins;
    mmgpu_pf_hit *hits;
    uint32_t hit_stride;
    uint32_t *hit_count, *q_diag_thr;
    // shard of a multi-GPU run (mmgpu_pf_set_shard): exchange records instead of hit lists
    mmgpu_pf_xhit *xhits;             // null = ordinary run
    const uint32_t *global_ids;       // [n_targets] local -> global id
    const uint32_t *q_nseg;           // overflow-path queries (their order key is not shard independent), may be null
    uint32_t *q_flags;                // long-sequence queries (scores not computed on the device), may be null; bit 1 is set here (below)
    // diagonal scoring of sequences: a query whose double-diagonal candidates number foundDiagonalsSize / 2 or more MAY take the
    // reference's unsorted branch (QueryMatcher.cpp:188,204-214: filter, unstable std::sort, exact scores, no rescoring) - the
    // candidate total bounds resultSize from above, such queries are flagged (bit 1 -> MMGPU_PF_SAT_TIE) and left to the host
    const uint32_t *cand_count;       // [nq * bins] candidates per (query, bin), null = no check (modes that count elsewhere)
    uint32_t cand_cap;
    int nucl;                         // nucleotide searches: saturated elements are ordered by target id (QueryMatcher.cpp:154)
    // max_hits above PF_MAX_HITS: the selected elements are sorted in global scratch, [nq][big_stride] keys + diagonals
    uint64_t *big_keys;
    uint16_t *big_diags;
    uint32_t big_stride;              // power of two >= max_hits
    const uint32_t *q_off, *peb, *list_base;
    const PfList *lists;
};

// ---- multi-GPU merge (pf_shard_kernels.hip)
constexpr int PF_XMERGE_CAP = 4096;    // exchange records per query (n_shards * stride) the merge kernel holds in LDS
struct PfXMergeArgs {
    const mmgpu_pf_xhit *xhits;       // [n_shards][nq][stride]
    const uint32_t *counts;           // [n_shards][nq]
    uint32_t n_shards, nq, stride;
    uint32_t max_hits, min_diag_score, ref_bins;
    const int32_t *q_self_score;      // [nq]
    const uint32_t *q_identity;       // [nq] GLOBAL id of the self hit, 0xFFFFFFFF = none
    mmgpu_pf_hit *out_hits;           // [nq][out_stride]
    uint32_t out_stride;
    uint32_t *out_counts;             // [nq]
    uint32_t *out_flags;              // [nq] bit 0: an overflow-path element took part (tie order at the cut not exact)
};
hipError_t launch_pf_xmerge(const PfXMergeArgs &A, hipStream_t s);

struct PfLocalizeArgs {
    const mmgpu_pf_hit *hits;         // [nq][stride] global ids
    const uint32_t *counts;
    uint32_t nq, stride, shard;
    const uint32_t *shard_of, *local_id;   // [global_db_size]
    mmgpu_pf_hit *local_hits;         // [nq][stride] local ids, list order kept
    uint32_t *local_counts;
    uint32_t *local_slot;             // [nq][stride] position of the hit in the query's merged list
};
hipError_t launch_pf_localize(const PfLocalizeArgs &A, hipStream_t s);

// ---- alignment records of the pairs a rank owns, gathered over the ranks (pf_shard_kernels.hip) ----
struct SwOwnedRec {                   // 32 bytes: one aligned pair and where it goes in the merged-list order
    mmgpu_sw_hit hit;
    uint32_t slot;                    // q * stride + position in query q's merged list
    uint32_t pad;
};
struct SwOwnedPackArgs {
    const mmgpu_sw_hit *res;          // [nq * stride] the batch's records, local list order
    const uint32_t *local_counts;     // [nq]
    const uint32_t *local_slot;       // [nq * stride]
    uint32_t nq, stride;
    SwOwnedRec *send;                 // [cap]
    uint32_t cap;
    uint32_t *counter;                // [2]: records packed, records that did not fit (overflow)
};
struct SwOwnedScatterArgs {
    const SwOwnedRec *recv;           // [n_ranks][cap]
    const uint32_t *counters;         // [n_ranks][2]
    uint32_t n_ranks, cap, n_slots;
    mmgpu_sw_hit *full;               // [n_slots], zeroed before
    uint32_t *status;                 // [2]: records scattered, ranks whose send buffer overflowed
};
hipError_t launch_sw_owned_pack(const SwOwnedPackArgs &A, hipStream_t s);
hipError_t launch_sw_owned_scatter(const SwOwnedScatterArgs &A, hipStream_t s);

constexpr int PF_MAX_SEG = 62;         // databaseHits flushes per query the device emulates (QueryMatcher.cpp:310-346)

struct PfSegArgs {
    const uint32_t *ovf_queries;      // queries that gather >= maxDbMatches entries
    uint32_t n_ovf;
    const uint32_t *q_off, *list_base, *pos_entry_base;
    const PfList *lists;
    uint64_t cap;                     // maxDbMatches
    uint32_t *seg_start;              // [nq][PF_MAX_SEG + 2]
    uint32_t *q_nseg;                 // [nq]; PF_MAX_SEG + 1 = more flushes than emulated
    uint32_t *q_final;                // [nq] entries of the last segment
    const uint32_t *q_entries;
};

struct PfOvfElem {                    // element of foundDiagonals on the overflow path
    uint32_t id;
    uint32_t score;                   // exact ungapped score, 0 = not scored yet (CounterResult::count == 0)
    long long ord;                    // position key: order inside a CPU bin through the merges (reversal = negation)
    uint16_t diag;
    uint16_t pad0;
    uint32_t pad1;
};

struct PfOvfArgs {
    PfDedupArgs D;
    const uint32_t *ovf_queries;
    uint32_t n_ovf;
    uint32_t step;                    // flush number 1..nseg, nseg + 1 = the final merge
    const uint32_t *q_final;
    const uint64_t *ovf_base;         // [n_ovf] first element of the query's region in buf_a / buf_b
    PfOvfElem *buf_a, *buf_b;
    uint32_t *o_count;                // [n_ovf * bins]
    uint32_t *totals;                 // [n_ovf][PF_MAX_SEG + 2] overflowHitCount after flush k
};

hipError_t launch_pf_segments(const PfSegArgs &A, hipStream_t s);
hipError_t launch_pf_overflow(const PfOvfArgs &A, hipStream_t s);

constexpr int PF_MERGE_CAP = 8192;     // hits per query the split-merge kernel sorts in LDS

struct PfMergeArgs {
    const mmgpu_pf_hit *hits;    // [n_splits][nq][stride]
    const uint32_t *counts;      // [n_splits][nq]
    uint32_t n_splits, nq, stride;
    uint32_t id_offset[64];
    mmgpu_pf_hit *out_hits;      // [nq][n_splits * stride]
    uint32_t *out_counts;        // [nq]
};

hipError_t launch_pf_merge(const PfMergeArgs &A, hipStream_t s);
hipError_t launch_pf_kmers(const PfKmerArgs &A, bool emit, hipStream_t s);
size_t pf_cofs_bytes(uint64_t table);
hipError_t launch_pf_cofs(const uint32_t *offsets, uint64_t table, void *cofs, hipStream_t s);
// persisted device layout (db_kernels.hip): checksum of a section where it lies on the device; the properties the kernels rely on
hipError_t db_section_checksum(const void *dev, size_t bytes, unsigned long long *scratch, uint64_t *sum, hipStream_t s);
hipError_t db_validate_layout(const uint32_t *off4, const uint32_t *len, uint32_t n, uint64_t res_bytes, uint32_t max_len, const uint32_t *offsets,
                              uint64_t table, const uint64_t *entries, uint64_t n_entries, uint32_t *scratch, uint32_t *bad, hipStream_t s);
hipError_t launch_pf_tiles(const uint32_t *q_tile_base, const uint32_t *q_ntiles, uint32_t nq, uint32_t *tile_q, uint32_t *tile_idx, hipStream_t s);
hipError_t launch_pf_bitmap(const uint32_t *offsets, uint64_t table, uint32_t *bitmap, unsigned long long *nonempty, hipStream_t s);
hipError_t launch_pf_scan(const uint32_t *in, const uint32_t *q_off, uint32_t nq, const uint64_t *base, uint32_t *out,
                          uint64_t *totals, hipStream_t s);
hipError_t launch_pf_split(const PfSplitArgs &A, uint32_t n_tiles, hipStream_t s);
hipError_t launch_pf_dedup(const PfDedupArgs &A, hipEvent_t after_replay, hipEvent_t after_ungapped, hipStream_t s);
hipError_t launch_pf_count(const PfDedupArgs &A, hipEvent_t after_a, hipEvent_t after_b, hipStream_t s);   // --diag-score 0
hipError_t launch_pf_long(const PfDedupArgs &A, bool long_queries, hipStream_t s);   // after launch_pf_dedup: sequences of >= 32768 residues
hipError_t launch_pf_select(const PfSelectArgs &A, uint32_t nq, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------
// index construction (ix_kernels.hip)
struct IxArgs {
    const uint8_t *t_res;
    const uint32_t *t_off4, *t_len;
    uint32_t n_targets;
    int k, pattern_len, kmer_thr;
    uint32_t kalph;
    uint8_t pat[16];
    int8_t self_score[32];      // (char) subMatrix[a][a], IndexBuilder.cpp:11-22
    uint32_t *counts;           // [table]: count pass = list lengths, fill pass = write cursors
    const uint32_t *offsets;    // [table + 1] (fill pass)
    uint64_t *entries;          // fill pass: seqId | position << 32, unordered inside a list
    uint32_t *scratch;          // one uint32 per residue slot, used by targets with more than 4096 windows
};

struct IxSortArgs {
    uint64_t table;
    const uint32_t *offsets;
    const uint64_t *src;
    uint64_t *dst;
    uint32_t *long_lists;       // k-mers whose list has more than 16 entries
    uint32_t *n_long;
    uint32_t long_cap;
};

hipError_t launch_ix_target(const IxArgs &A, bool fill, hipStream_t s);

// tantan masking of the resident targets (tantan_kernel.hip)
struct TantanArgs {
    const uint8_t *t_res;              // unmasked residues
    uint8_t *out_res;                  // the masked copy (same layout; only masked positions are written)
    const uint32_t *t_off4, *t_len;
    const uint32_t *order;             // [n] target ids, longest first: 64 consecutive ones share a wavefront
    uint32_t n;
    const double *lr;                  // [alphabet][alphabet] likelihood ratios (ProbabilityMatrix, BaseMatrix.h:83-101)
    const double *b2f;                 // [50] background -> repeat state probabilities (Tantan constructor, tantan.cpp:118-131)
    int alphabet;
    double repeat_prob, repeat_end_prob, min_mask_prob;
    uint8_t mask_letter;
    float *probs;                      // forward background probabilities, [wave][position][64 lanes]
    double *scales;                    // scale factors, [wave][position / 16][64 lanes]
    const uint64_t *wave_prob_base, *wave_scale_base;
    unsigned long long *n_masked;
};
hipError_t launch_tantan_mask(const TantanArgs &A, hipStream_t s);
// code-object loading ahead of the first launch (mmgpu_warmup)
void warm_sw();
void warm_block();
void warm_pf();
void warm_ix();
void warm_tantan();
void warm_bt();
hipError_t launch_ix_sort_short(const IxSortArgs &A, hipStream_t s);
hipError_t launch_ix_sort_long(const IxSortArgs &A, uint32_t n_long, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------
// banded traceback (bt_kernel.hip)
struct BtJob {
    uint32_t slot;      // where the caller wants the result (index into info[])
    uint32_t query, target;
    int32_t q_start, q_end, t_start, t_end, score;
    uint64_t bt_off;
};

struct BtLaunch {
    const BtJob *jobs;
    uint32_t n_jobs;
    const uint8_t *q_res;
    const int8_t *q_cb;
    const uint32_t *q_off;
    const int8_t *q_prof;       // profile queries, as in SwLaunch (banded_sw<PROFILE_SEQ>, StripedSmithWaterman.cpp:1565-1567)
    const uint32_t *q_prof_off;
    const uint8_t *t_res;
    const uint32_t *t_off4;
    const int8_t *mat;
    int alphabet, gap_open, gap_extend;
    uint32_t *scratch;          // [blocks][words_per_lane][64 lanes]
    uint32_t words_per_lane;
    uint32_t band_cap;          // words per band row (3 rows), the rest of words_per_lane holds direction bits
    mmgpu_sw_bt *info;          // indexed by BtJob::slot
    char *bt;
    // wave kernel (bt_wave_kernel.hip): one pool of direction bytes, carved up by a bump allocator
    uint8_t *dir_pool;
    unsigned long long dir_pool_bytes;
    unsigned long long *dir_cursor;
};

// ---- block aligner on the device (block_kernel.hip; row a15): int16-range hits ----
constexpr int BLOCK_MAX_SIZE = 512;        // largest block the first-tier kernel holds in LDS
constexpr int BLOCK_MID_SIZE = 2048;       // second tier: still in LDS (32 KB)
constexpr int BLOCK_REF_MAX_SIZE = 4096;   // MAX_SIZE of the reference (StripedSmithWaterman.cpp:37); third tier, borders in HBM
struct BlockJob {
    uint32_t query, target;
    int32_t score, q_end, t_end;
    uint32_t slot;                // index into out / bt_off
};
struct BlockLaunch {
    const BlockJob *jobs;
    uint32_t n_jobs;
    const uint8_t *q_res;
    const int8_t *q_cb;
    const uint32_t *q_off;
    const uint8_t *t_res;
    const uint32_t *t_off4;
    const int8_t *q_prof;         // profile queries, as in SwLaunch: int8 [alphabet][qlen] per query; null = none in the batch
    const uint32_t *q_prof_off;
    int alphabet;
    const int8_t *scores;         // AAMatrix::scores [27 * 32] as ssw_init leaves it (new_simple(1, -1) + set_num of the matrix)
    int gap_open, gap_extend;     // the crate's convention: negative
    mmgpu_sw_block *out;
    const uint64_t *bt_off;
    char *bt;
    uint8_t *pool;                // scratch slots: block list + trace of one pair
    uint64_t slot_bytes;
    uint32_t n_pool_slots;
    uint32_t *pool_busy;
    // test aid (mmgpu_sw_block_growth): per pair 1 + 4 * growth_cap words - the number of blocks of the pair's last alignment run and
    // (i, j, height << 16 | width, right) of each, i.e. Trace::block_start / block_size / right when align_core returns; null = off
    uint32_t *growth = nullptr;
    uint32_t growth_cap = 0;
};
// tier 0: blocks up to BLOCK_MAX_SIZE rows (LDS), 1: up to BLOCK_MID_SIZE (LDS), 2: up to BLOCK_REF_MAX_SIZE rows, border arrays
// in the first 8 * 4096 * 2 bytes of the pair's scratch slot
hipError_t launch_sw_block(const BlockLaunch &L, int tier, hipStream_t stream);
struct BkBlock { uint32_t i, j; uint16_t h, w; uint32_t right, tstart; };   // Trace::block_start / block_size / right + the block's first trace entry

// ---- the same aligner, four pairs per wavefront (block4_kernel.hip): sequence queries; what a launch answers MMGPU_BLOCK_TOO_LARGE
// goes to the next form ----
struct Block2Job {
    uint32_t query, target;
    int32_t score, q_end, t_end;
    uint32_t slot;                // index into out / bt_off
    uint64_t pool_off;            // the pair's scratch (block list + trace) in the pool; unused without a trace
    uint32_t pool_bytes, pad;     // pad: block4_kernel.hip - the first minimum block size to try (0 = 32)
};
struct Block2Launch {
    const Block2Job *jobs;
    uint32_t n_jobs;
    uint32_t *counter;            // the queue's head (zero before the launch)
    const uint8_t *q_res;
    const int8_t *q_cb;
    const uint32_t *q_off;
    const uint8_t *t_res;
    const uint32_t *t_off4;
    const int8_t *scores;         // AAMatrix::scores [27 * 32], as BlockLaunch::scores
    int gap_open, gap_extend;     // the crate's convention: negative
    mmgpu_sw_block *out;
    const uint64_t *bt_off;       // multiples of four
    char *bt;                     // null: no strings
    uint8_t *pool;
    uint32_t *growth = nullptr;   // test aid, as BlockLaunch::growth
    uint32_t growth_cap = 0;
    uint8_t *ck_pool = nullptr;   // block4_kernel.hip: checkpoint arrays, 8 * rows bytes per row of a wavefront (4 rows a wavefront; skewed form: one)
    uint32_t trace_bytes = 0;     // walk kernel: the trace is the skewed form's (a byte per cell)
};
// trace = false: start positions only (no scratch, no walk)
// ---- four pairs per wavefront, one 16-lane DPP row = one vector of the crate = one pair (block4_kernel.hip); same launch record ----
constexpr int BLOCK4_MAX_SIZE = 128;     // (tuning aid: MMGPU_BLOCK4_ROWS=128) 1 KB of LDS per pair
constexpr int BLOCK4_LARGE_SIZE = 256;   // the default: 2 KB of LDS per pair
constexpr int BLOCK4_SKEW_SIZE = 1024;   // the skewed form (one pair per wavefront, rows pipelined over columns): first launch; the second holds BLOCK_REF_MAX_SIZE
hipError_t launch_sw_block4(const Block2Launch &L, bool trace, int form, uint32_t n_waves, hipStream_t stream);
hipError_t launch_sw_block4_walk(const Block2Launch &L, hipStream_t stream);
void warm_block4();

// ---- search semantics on the device (block_select.hip, mmgpu_sw_block_starts) ----
struct BlockSelectArgs {
    const mmgpu_sw_hit *res;      // the batch's result records
    uint32_t pairs;
    const uint32_t *qout_off;     // [n_queries + 1] first result slot of every query
    uint32_t n_queries;
    const int32_t *q_minstart;
    const uint32_t *slot_target;  // target id of every result slot
    BlockJob *jobs;               // out: the selected pairs, slot = position in this list
    uint32_t *pair_of_slot;       // out: their result slots
    mmgpu_sw_block *blk;          // out: their answers, initialised (MMGPU_BLOCK_TOO_LARGE until a kernel decides)
    uint32_t *count;              // out (zero before the launch)
    uint32_t cap;
};
struct BlockScatterArgs {
    const mmgpu_sw_block *blk;
    const uint32_t *pair_of_slot;
    uint32_t n;
    mmgpu_sw_hit *res;
    uint8_t *rev_force;           // [pairs], zero before the launch: set for the pairs the block aligner declined
    uint32_t *counts;             // [3] OK / DECLINED / TOO_LARGE (zero before the launch)
};
hipError_t launch_block_select(const BlockSelectArgs &A, hipStream_t stream);
hipError_t launch_block_scatter(const BlockScatterArgs &A, hipStream_t stream);

hipError_t launch_sw_traceback(const BtLaunch &L, hipStream_t stream);
hipError_t launch_sw_traceback_wave(const BtLaunch &L, hipStream_t stream);

// host-side loops over targets / index entries / queries: plain std::thread chunks (the library carries no OpenMP runtime)
inline unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    return n == 0 ? 1 : std::min(n, 64u);
}

template <typename F>
inline void parallel_for(size_t n, F f) {
    const unsigned nt = (unsigned)std::min<size_t>(host_threads(), std::max<size_t>(n, 1));
    if (nt <= 1 || n < 4096) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const size_t a = std::min(n, (size_t)t * chunk), b = std::min(n, a + chunk);
        if (a < b) th.emplace_back([=] { f(a, b); });
    }
    for (auto &x : th) x.join();
}

// ---------------------------------------------------------------------------------------------------------
// host-side plumbing shared by mmgpu_api.hip and pf_api.hip
extern thread_local std::string g_last_error;
inline int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            return mmgpu::fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));         \
    } while (0)

// every device buffer of the library is allocated and freed through these two
inline hipError_t dev_malloc(void **p, size_t n) { return hipMalloc(p, n); }
inline void dev_free(void *p) {
    if (p) (void)hipFree(p);
}

// Freed device blocks of one context, kept for the next batch: a streaming search prepares and frees one alignment
// batch per prefilter batch, and hipMalloc / hipFree (which also drains the device) per buffer would cost more than
// the kernels.  Reuse is safe because every use of a block is ordered on the context's stream.  Sizes are rounded to 3 significant bits so that batches of
// similar shape hit the same classes.
struct BlockCache {
    std::multimap<size_t, void *> blocks;
    size_t cached = 0;
    bool closed = false;   // the context is gone (batches may outlive it): blocks go straight back to the runtime
    std::mutex lock;       // a prefilter thread and an alignment thread may work on one context (the fused search of the drop-in)
    // a third of the device's memory (96 GB of 288), read once; an allocation that fails trims the cache and retries
    static size_t limit() {
        static const size_t v = []() -> size_t {
            size_t f = 0, t = 0;
            if (hipMemGetInfo(&f, &t) != hipSuccess || t == 0) return (size_t)16 << 30;
            return t / 3;
        }();
        return v;
    }
    static size_t round_up(size_t n) {
        if (n <= 512) return 512;
        int top = 63 - __builtin_clzll((unsigned long long)n);
        const size_t step = (size_t)1 << (top - 2);
        return (n + step - 1) & ~(step - 1);
    }
    // the smallest parked block of cap .. 2 cap bytes (cap becomes its size): the buffers of consecutive batches differ in size by
    // what their queries happen to need, and on some hosts a fresh hipMalloc costs 25 - 40 ms per GB
    void *take(size_t &cap) {
        std::lock_guard<std::mutex> guard(lock);
        auto it = blocks.lower_bound(cap);
        if (it == blocks.end() || it->first > 2 * cap) return nullptr;
        void *p = it->second;
        cap = it->first;
        blocks.erase(it);
        cached -= cap;
        return p;
    }
    void give(void *p, size_t cap) {
        std::lock_guard<std::mutex> guard(lock);
        if (closed || cached + cap > limit()) { dev_free(p); return; }
        blocks.emplace(cap, p);
        cached += cap;
    }
    void trim() {
        std::lock_guard<std::mutex> guard(lock);
        for (auto &kv : blocks) dev_free(kv.second);
        blocks.clear();
        cached = 0;
    }
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    size_t cap = 0;                // allocated size when the block came from / goes back to a cache
    std::shared_ptr<BlockCache> cache;   // optional (bind): where freed blocks go
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes), cap(o.cap), cache(std::move(o.cache)) { o.p = nullptr; o.bytes = 0; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p; bytes = o.bytes; cap = o.cap; cache = std::move(o.cache);
            o.p = nullptr; o.bytes = 0; o.cap = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void bind(const std::shared_ptr<BlockCache> &c) {
        cache = c;
    }
    void release() {
        if (!p) return;
        if (cache && cap) cache->give(p, cap);
        else dev_free(p);
        p = nullptr;
        cap = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        bytes = n;
        if (n == 0) return hipSuccess;
        if (cache) {
            cap = BlockCache::round_up(n);
            p = cache->take(cap);
            if (p) return hipSuccess;
            hipError_t e = dev_malloc(&p, cap);
            if (e != hipSuccess) {      // out of memory with blocks parked in the cache: give them back and retry
                (void)hipGetLastError();
                cache->trim();
                e = dev_malloc(&p, cap);
            }
            if (e != hipSuccess) { p = nullptr; cap = 0; bytes = 0; }   // (bytes = 0: a later, smaller reserve() must allocate)
            return e;
        }
        const hipError_t e = dev_malloc(&p, n);
        if (e != hipSuccess) { p = nullptr; bytes = 0; (void)hipGetLastError(); }
        return e;
    }
    // grow-only (contents are not preserved)
    hipError_t reserve(size_t n) { return n <= bytes ? hipSuccess : alloc(n + n / 8); }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

template <typename T>
inline hipError_t upload(DevBuf &b, const std::vector<T> &v, hipStream_t s) {
    hipError_t e = b.alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (v.empty()) return hipSuccess;
    return hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

// the same through a slice of pinned staging memory (base / cap / used: the caller's arena, large enough - upload_pinned_need):
// the copy is enqueued and the host goes on; the arena must stay untouched until the stream has passed the copy
inline size_t upload_pinned_need(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
template <typename T>
inline hipError_t upload_pinned(DevBuf &b, const std::vector<T> &v, hipStream_t s, void *base, size_t cap, size_t &used) {
    hipError_t e = b.alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (v.empty()) return hipSuccess;
    const size_t n = v.size() * sizeof(T);
    if (!base || used + upload_pinned_need(n) > cap) return hipMemcpyAsync(b.p, v.data(), n, hipMemcpyHostToDevice, s);
    void *st = static_cast<char *>(base) + used;
    used += upload_pinned_need(n);
    memcpy(st, v.data(), n);
    return hipMemcpyAsync(b.p, st, n, hipMemcpyHostToDevice, s);
}

// nucleotide alignment step (nucl_kernel.hip; NuclLaunch is declared in nucl_core.h)
hipError_t launch_nucl_align(const NuclLaunch &L, unsigned blocks, hipStream_t stream);     // 16 lanes per alignment
hipError_t launch_nucl_align_wave(const NuclLaunch &L, unsigned blocks, hipStream_t stream);   // one wavefront per alignment, state in registers (nucl_wave.h): the default

hipError_t launch_pf_order(const PfKmerArgs &A, uint32_t *order, const std::shared_ptr<BlockCache> &cache, hipStream_t s);   // pf_order.hip

struct PfIndex;   // pf_api.hip

}  // namespace mmgpu

namespace mmgpu {
// communicator over the contexts that hold the shards of one target database (comm.hip)
struct Comm {
    void *nccl = nullptr;          // ncclComm_t; null = the copy transport of a one-process run (multi_api.hip)
    int rank = 0, n_ranks = 1;
    std::string transport;         // "rccl" | "copy"
};
}  // namespace mmgpu

struct mmgpu_ctx {
    int device = 0;
    mmgpu::Comm *comm = nullptr;   // multi-GPU runs: mmgpu_comm_init_rank / mmgpu_init_multi
    hipStream_t stream = nullptr;
    bool owns_stream = false;      // mmgpu_init_multi gives every context a stream of its own
    mmgpu::DeviceDb db;
    std::vector<uint32_t> h_len;   // host copy of target lengths (scheduling)
    uint32_t mean_len = 0;
    int compute_units = 0;
    std::string name;
    mmgpu::PfIndex *pf = nullptr;  // prefilter index resident in HBM (pf_api.hip)
    // the prefilter's view of the targets when it differs from the alignment's: residues with tantan-masked letters replaced by
    // X (mmgpu_pf_mask_targets; same offsets / lengths as `db`).  null: the prefilter reads db.res (--mask 0, or a caller that
    // loaded an already masked SequenceLookup)
    uint8_t *pf_masked_res = nullptr;
    const uint8_t *pf_res() const { return pf_masked_res ? pf_masked_res : db.res; }
    // shard of a multi-GPU run (mmgpu_pf_set_shard); reset by mmgpu_load_targets
    struct Shard {
        bool on = false;
        uint32_t n_shards = 1, shard = 0, global_n = 0;
        mmgpu::DevBuf d_global_ids, d_shard_of, d_local_id;
    } shard;
    std::shared_ptr<mmgpu::BlockCache> cache = std::make_shared<mmgpu::BlockCache>();   // device blocks of freed alignment batches
    // the alignment kernel groups run concurrently on side streams forked from / joined to `stream` (mmgpu_sw_run)
    hipStream_t side[4] = {};      // (SW_GROUPS of them are used)
    hipEvent_t fork = nullptr, join[4] = {};
    // pinned staging for the uploads of mmgpu_sw_prepare_from_pf: a copy from pageable memory blocks the calling thread until the
    // stream has reached it - i.e. until the prefilter batch before it has finished - and everything the host still had to do for
    // the alignment batch (ordering ~1e5 jobs) then ran with the device idle
    void *pinned = nullptr;
    size_t pinned_cap = 0, pinned_used = 0;
};

// device memory for a context's long-lived buffers (targets, masked view): when the runtime is out of memory the blocks the
// context's own cache has parked go back first (DevBuf::alloc does the same for batch buffers)
inline hipError_t dev_malloc_ctx(mmgpu_ctx *c, void **p, size_t n) {
    hipError_t e = mmgpu::dev_malloc(p, n);
    if (e != hipSuccess && c && c->cache) {
        (void)hipGetLastError();
        c->cache->trim();
        e = mmgpu::dev_malloc(p, n);
    }
    return e;
}

struct mmgpu_pf_batch_t;
struct mmgpu_sw_batch_t;
namespace mmgpu {
// comm.hip
int comm_require_rccl();
int comm_group_start();
int comm_group_end();
int comm_init_all(mmgpu_ctx **ctxs, int n);
int comm_allgather(mmgpu_ctx *c, const void *send, void *recv, size_t bytes);
void comm_free(mmgpu_ctx *c);
// the phases of the two exchange steps (pf_api.hip, mmgpu_api.hip); mmgpu_pf_exchange_merge / mmgpu_sw_gather_owned run them on
// one context, multi_api.hip runs each phase over the contexts of a process (the collective inside an RCCL group)
struct XchgBlock { const void *send; void *recv; size_t bytes; };   // one all-gather: `bytes` per rank
int pf_xchg_begin(mmgpu_ctx *c, mmgpu_pf_batch_t *b, int n_ranks, XchgBlock blocks[2]);
int pf_xchg_merge(mmgpu_ctx *c, mmgpu_pf_batch_t *b, int n_ranks, const uint32_t *identity_global);
// sharded runs: the queries whose merged list is flagged inexact, run once more against a context that holds the whole database
// (mmgpu_pf_exchange_redo_unsplit; multi_api.hip applies one run's rows to every context of the process)
struct PfRedoRows {
    std::vector<uint32_t> q;            // the flagged queries
    std::vector<mmgpu_pf_hit> hits;     // [q.size()][stride]: the unsplit run's lists (ids of the whole database)
    std::vector<uint32_t> counts;
    std::vector<int32_t> status;        // the unsplit run's own status (MMGPU_PF_OK, or what the host has to do itself)
    uint32_t stride = 0;
};
int pf_redo_flagged(mmgpu_ctx *c, mmgpu_pf_batch_t *b, std::vector<uint32_t> &flagged);      // synchronises the context's stream
int pf_redo_run(mmgpu_ctx *full, const mmgpu_pf_params *par, const mmgpu_pf_query *qs, const std::vector<uint32_t> &flagged, PfRedoRows &rows);
int pf_redo_apply(mmgpu_ctx *c, mmgpu_pf_batch_t *b, const PfRedoRows &rows);               // synchronises the context's stream
const int32_t *pf_batch_redo_status(const mmgpu_pf_batch_t *b);      // [nq]: -1 = not re-run, else the unsplit run's status
int sw_gather_begin(mmgpu_ctx *c, mmgpu_sw_batch_t *b, int n_ranks, XchgBlock blocks[2]);
int sw_gather_finish(mmgpu_ctx *c, mmgpu_sw_batch_t *b, int n_ranks);
int sw_gather_overflowed(mmgpu_ctx *c, mmgpu_sw_batch_t *b, bool *overflowed);   // true once: the caller repeats the phases (dense buffers)
int pf_batch_merged_flags(mmgpu_pf_batch_t *b, const void **d_flags);
const int32_t *pf_batch_host_status(const mmgpu_pf_batch_t *b);
bool pf_batch_merged_lists(mmgpu_pf_batch_t *b, const mmgpu_pf_hit **hits, const uint32_t **counts, uint32_t *stride, uint32_t *nq);
void pf_index_free(mmgpu_ctx *c);
void db_release(mmgpu_ctx *c);      // mmgpu_api.hip: targets, masked view, index and shard description of a context
// device-resident results of a prefilter batch that has been run (false if it has not)
bool pf_batch_device_lists(mmgpu_pf_batch_t *b, const mmgpu_pf_hit **hits, const uint32_t **counts, uint32_t *stride, uint32_t *nq);
}
#endif
