// Host side of the prefilter half of libmmgpu: table builders, index residency, batch preparation and the
// launch sequence around pf_kernels.hip.  See include/mmgpu.h for the contract.
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "mmgpu_internal.h"
#include "sat_ties.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace mmgpu;

namespace mmgpu {

struct PfIndex {
    int k = 0, alphabet = 0, kalph = 0, spaced = 0;
    int kbase = 0;               // base of the k-mer index (kalph, or the full alphabet for a handed-over index of profile targets)
    uint8_t pat[16] = {0};
    int pattern_len = 0;
    bool has_tables = false;   // similar-k-mer score tables present (false: exact k-mer matching only)
    uint32_t n3 = 0;
    uint64_t table = 0, n_entries = 0;
    DevBuf d_s3, d_i3, d_cum3, d_s2, d_i2, d_cum2, d_offsets, d_entries, d_mat;
    DevBuf d_cofs;               // compact offset table (pf_cofs_kernel): what the similar-k-mer kernels look lists up in
    bool use_cofs = false;
    DevBuf d_nonempty;           // one bit per k-mer (pf_bitmap_kernel), used when the index is sparse
    bool use_bitmap = false;
    double nonempty_frac = 1.0;  // k-mers with a list / all k-mers
    uint32_t cum_w = 0, cum2_w = 0;
    int32_t score_min = 0, score2_min = 0;
    std::vector<int8_t> h_mat;   // ungapped matrix (host copy for the self score)
    // Large working buffers, shared by all batches of this context (grow-only; batches run one at a time on the
    // context's stream): index lists, split tiles, candidates, survivors.
    DevBuf w_lists, w_split, w_split_hi, w_bin_off, w_cand, w_surv, w_tile_q, w_tile_idx;
    const void *w_owner = nullptr;   // batch whose last run the working buffers hold (debug fetch)
};

// Sparse indexes (a shard of a multi-GPU run holds 1/N of the entries over the same k-mer space; small databases): most similar
// k-mers of a query have no list.  The bit table is consulted by pf_kmers_kernel when fewer than 60 % of the k-mers have one
// similar-k-mer searches with k = 6 only (the table of k = 7 has 1.3e9 k-mers).
static hipError_t pf_index_bitmap(mmgpu_ctx *c, PfIndex *P) {
    P->use_bitmap = false;
    if (!P->has_tables || P->k != 6 || P->table > (1ull << 28)) return hipSuccess;
    hipStream_t s = c->stream;
    DevBuf d_cnt;
    hipError_t rc = P->d_nonempty.alloc(((P->table + 31) / 32 + 1) * 4);
    if (rc == hipSuccess) rc = d_cnt.alloc(8);
    if (rc == hipSuccess) rc = hipMemsetAsync(d_cnt.p, 0, 8, s);
    if (rc == hipSuccess) rc = launch_pf_bitmap(P->d_offsets.as<uint32_t>(), P->table, P->d_nonempty.as<uint32_t>(), d_cnt.as<unsigned long long>(), s);
    unsigned long long n = 0;
    if (rc == hipSuccess) rc = hipMemcpyAsync(&n, d_cnt.p, 8, hipMemcpyDeviceToHost, s);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);
    if (rc != hipSuccess) return rc;
    P->nonempty_frac = P->table ? (double)n / (double)P->table : 1.0;
    P->use_bitmap = P->nonempty_frac < 0.6;
    if (!P->use_bitmap) P->d_nonempty.release();
    return hipSuccess;
}

// the compact offset table of the similar-k-mer kernels (k = 6 and k = 7 with score tables; not for exact matching, whose one
// look-up per window does not pay for it).  MMGPU_PF_COFS=0 keeps the look-ups on the full table (A/B runs).
static hipError_t pf_index_cofs(mmgpu_ctx *c, PfIndex *P) {
    P->use_cofs = false;
    const char *e = getenv("MMGPU_PF_COFS");
    if (!P->has_tables || P->n_entries >= (1ull << 31) || P->table > (1ull << 31) || (e && e[0] == '0')) return hipSuccess;
    hipError_t rc = P->d_cofs.alloc(pf_cofs_bytes(P->table));
    if (rc == hipSuccess) rc = launch_pf_cofs(P->d_offsets.as<uint32_t>(), P->table, P->d_cofs.p, c->stream);
    if (rc == hipSuccess) P->use_cofs = true;
    return rc;
}

void pf_index_free(mmgpu_ctx *c) {
    if (c && c->pf) {
        delete c->pf;
        c->pf = nullptr;
    }
}

}  // namespace mmgpu

// spaced_seed_<k> of Sequence.h:20-50 as bit masks (bit i = pattern position i) and their lengths
static const uint32_t SPACED_BITS[16] = {0, 0, 0, 0, 0x17u, 0xA13u, 0x32Bu, 0x66Bu, 0xCEBu, 0x366Bu, 0x6D6Bu, 0x1B66Bu, 0x6B66Bu, 0xD6CEBu,
                                         0x1B6CEBu, 0x6D1BD7u};
static const uint8_t SPACED_LEN[16] = {0, 0, 0, 0, 5, 12, 10, 11, 12, 14, 15, 17, 19, 20, 21, 23};

static int window_pattern(int k, int spaced, uint8_t *pat) {     // k in [4, 15]
    if (!spaced) {
        for (int i = 0; i < k; i++) pat[i] = (uint8_t)i;
        return k;
    }
    const int n = SPACED_LEN[k];
    int c = 0;
    for (int i = 0; i < n; i++)
        if ((SPACED_BITS[k] >> i) & 1u) pat[c++] = (uint8_t)i;
    return n;
}

// ---------------------------------------------------------------------------------------------------------
// ExtendedSubstitutionMatrix::calcScoreMatrix (src/prefiltering/ExtendedSubstitutionMatrix.cpp:20-71).  Row r lists
// every span-mer by descending score; the reference uses std::stable_sort over the cartesian-product enumeration
// (first residue slowest, :103-128).  Scores are small integers, so a stable counting sort per row does the same.
extern "C" int mmgpu_host_score_matrix(const int16_t *submat, int alphabet, int span, int16_t *score, uint32_t *index) {
    return mmgpu_host_score_matrix_rows(submat, alphabet, span, 0, score, index);
}

// the same into rows of row_stride elements (ScoreMatrix::rowSize: the reference pads its rows for SIMD loads, :24-25; the
// padding elements are the caller's); row_stride 0 = (alphabet - 1)^span, no padding
extern "C" int mmgpu_host_score_matrix_rows(const int16_t *submat, int alphabet, int span, size_t row_stride, int16_t *score, uint32_t *index) {
    if (!submat || !score || !index) return fail(MMGPU_ERR_ARG, "mmgpu_host_score_matrix: NULL argument");
    if (alphabet < 3 || alphabet > 32 || span < 1 || span > 3) return fail(MMGPU_ERR_ARG, "mmgpu_host_score_matrix: bad alphabet/span");
    const int ka = alphabet - 1;
    size_t n = 1;
    for (int i = 0; i < span; i++) n *= (size_t)ka;
    const size_t stride = row_stride ? row_stride : n;
    if (stride < n) return fail(MMGPU_ERR_ARG, "mmgpu_host_score_matrix_rows: row_stride smaller than the row");
    // enumeration e -> (index in int2index order, residues)
    std::vector<uint32_t> e2idx(n);
    std::vector<uint8_t> e2res(n * (size_t)span);
    for (size_t e = 0; e < n; e++) {
        size_t t = e;
        for (int p = span - 1; p >= 0; p--) {
            e2res[e * span + p] = (uint8_t)(t % (size_t)ka);
            t /= (size_t)ka;
        }
        size_t idx = 0, pw = 1;
        for (int p = 0; p < span; p++) {
            idx += e2res[e * span + p] * pw;
            pw *= (size_t)ka;
        }
        e2idx[e] = (uint32_t)idx;
    }
    int lo = 0, hi = 0;
    for (int a = 0; a < ka; a++)
        for (int b = 0; b < ka; b++) {
            lo = std::min<int>(lo, submat[a * alphabet + b]);
            hi = std::max<int>(hi, submat[a * alphabet + b]);
        }
    const int smin = lo * span, smax = hi * span, range = smax - smin + 1;
    parallel_for(n, [&](size_t r0, size_t r1) {
        std::vector<int16_t> sc(n);
        std::vector<uint32_t> cursor((size_t)range + 1);
        for (size_t e = r0; e < r1; e++) {
            const uint8_t *a = &e2res[e * span];
            std::fill(cursor.begin(), cursor.end(), 0u);
            for (size_t f = 0; f < n; f++) {
                const uint8_t *b = &e2res[f * span];
                int s = 0;
                for (int p = 0; p < span; p++) s += submat[a[p] * alphabet + b[p]];
                sc[f] = (int16_t)s;
                cursor[(size_t)(smax - s) + 1]++;   // bucket 0 = highest score
            }
            for (int z = 0; z < range; z++) cursor[(size_t)z + 1] += cursor[z];
            int16_t *srow = score + (size_t)e2idx[e] * stride;
            uint32_t *irow = index + (size_t)e2idx[e] * stride;
            for (size_t f = 0; f < n; f++) {
                const uint32_t o = cursor[(size_t)(smax - sc[f])]++;
                srow[o] = sc[f];
                irow[o] = e2idx[f];
            }
        }
    });
    return MMGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// IndexTable::addKmerCount / addSequence / sortDBSeqLists (src/prefiltering/IndexTable.h:135-191,350-403) as
// IndexBuilder::fillDatabase drives them with masking off (IndexBuilder.cpp:118-166,226-270).
namespace {
struct KmerWindows {
    int k, ka, alphabet, plen, thr;
    uint8_t pat[16];
    int8_t self[32];   // (char) subMatrix[a][a], IndexBuilder.cpp:11-22
    // unique k-mers of one target with their first position, sorted by k-mer; key = kmer << 16 | pos
    void extract(const uint8_t *s, uint64_t len, std::vector<uint64_t> &buf) const {
        buf.clear();
        for (uint64_t i = 0; i + (uint64_t)plen <= len; i++) {
            uint64_t idx = 0, pw = 1;
            int sc = 0;
            bool x = false;
            for (int p = 0; p < k; p++) {
                const uint8_t r = s[i + pat[p]];
                if (r >= ka) { x = true; break; }
                idx += r * pw;
                pw *= (uint64_t)ka;
                sc += self[r];
            }
            if (x || (thr > 0 && sc < thr)) continue;
            buf.push_back((idx << 16) | (i & 0xFFFFu));
        }
        std::sort(buf.begin(), buf.end());
        size_t w = 0;
        uint64_t prev = ~0ull;
        for (size_t z = 0; z < buf.size(); z++) {
            if ((buf[z] >> 16) != prev) buf[w++] = buf[z];
            prev = buf[z] >> 16;
        }
        buf.resize(w);
    }
};
}  // namespace

extern "C" int mmgpu_host_index_build(const uint8_t *residues, const uint64_t *seq_off, uint32_t n, const int16_t *kmer_submat,
                                      int alphabet, int k, int spaced, int kmer_thr, uint64_t *offsets, uint32_t *ids,
                                      uint16_t *pos, uint64_t *n_entries) {
    if (!residues || !seq_off || !kmer_submat || !offsets || !n_entries)
        return fail(MMGPU_ERR_ARG, "mmgpu_host_index_build: NULL argument");
    if ((k < 5 || k > 7) || alphabet < 3 || alphabet > 32) return fail(MMGPU_ERR_ARG, "mmgpu_host_index_build: bad k/alphabet");
    if ((ids == nullptr) != (pos == nullptr)) return fail(MMGPU_ERR_ARG, "mmgpu_host_index_build: ids and pos go together");
    KmerWindows W;
    W.k = k;
    W.ka = alphabet - 1;
    W.alphabet = alphabet;
    W.thr = kmer_thr;
    W.plen = window_pattern(k, spaced, W.pat);
    for (int a = 0; a < alphabet; a++) W.self[a] = (int8_t)(char)kmer_submat[a * alphabet + a];
    uint64_t table = 1;
    for (int i = 0; i < k; i++) table *= (uint64_t)W.ka;
    std::vector<std::atomic<uint32_t>> cnt(table);
    for (auto &c : cnt) c.store(0, std::memory_order_relaxed);
    parallel_for(n, [&](size_t a, size_t b) {
        std::vector<uint64_t> buf;
        for (size_t t = a; t < b; t++) {
            W.extract(residues + seq_off[t], seq_off[t + 1] - seq_off[t], buf);
            for (uint64_t v : buf) cnt[v >> 16].fetch_add(1, std::memory_order_relaxed);
        }
    });
    uint64_t run = 0;
    for (uint64_t z = 0; z < table; z++) {
        offsets[z] = run;
        run += cnt[z].load(std::memory_order_relaxed);
    }
    offsets[table] = run;
    *n_entries = run;
    if (!ids) return MMGPU_OK;
    for (auto &c : cnt) c.store(0, std::memory_order_relaxed);
    parallel_for(n, [&](size_t a, size_t b) {
        std::vector<uint64_t> buf;
        for (size_t t = a; t < b; t++) {
            W.extract(residues + seq_off[t], seq_off[t + 1] - seq_off[t], buf);
            for (uint64_t v : buf) {
                const uint64_t o = offsets[v >> 16] + cnt[v >> 16].fetch_add(1, std::memory_order_relaxed);
                ids[o] = (uint32_t)t;
                pos[o] = (uint16_t)(v & 0xFFFFu);
            }
        }
    });
    // sortDBSeqLists: every list by (seqId, position); one entry per (k-mer, target), so seqId alone decides
    parallel_for(table, [&](size_t a, size_t b) {
        std::vector<uint64_t> tmp;
        for (size_t z = a; z < b; z++) {
            const uint64_t o0 = offsets[z], o1 = offsets[z + 1];
            if (o1 - o0 < 2) continue;
            tmp.resize(o1 - o0);
            for (uint64_t e = o0; e < o1; e++) tmp[e - o0] = ((uint64_t)ids[e] << 16) | pos[e];
            std::sort(tmp.begin(), tmp.end());
            for (uint64_t e = o0; e < o1; e++) {
                ids[e] = (uint32_t)(tmp[e - o0] >> 16);
                pos[e] = (uint16_t)(tmp[e - o0] & 0xFFFFu);
            }
        }
    });
    return MMGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// tables every QueryMatcher gets (score matrices, cumulative counts, ungapped matrix): shared by the two ways of
// obtaining the index (host arrays / built in HBM).  `from_host` = validate and copy the host index as well.
static int pf_setup(mmgpu_ctx *c, const mmgpu_pf_index *ix, bool from_host, PfIndex **out) {
    if (!c || !ix) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: NULL argument");
    if (!c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_pf_load_index: load the targets (SequenceLookup) first");
    const bool tables = ix->score3 != nullptr && ix->index3 != nullptr;    // without them: exact k-mer matching only
    if (tables && (ix->kmer_size < 5 || ix->kmer_size > 7)) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_load_index: similar k-mers need k = 5, 6 or 7");
    if (ix->kmer_size < 4 || ix->kmer_size > 15) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_load_index: k must be in [4, 15]");
    if (tables && ix->kmer_size != 6 && (!ix->score2 || !ix->index2)) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: k = 5 and k = 7 need the 2-mer ScoreMatrix");
    if (ix->alphabet != c->db.alphabet) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: alphabet differs from the loaded targets");
    if (ix->alphabet < 2 || ix->alphabet > 32) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: alphabet must be in [2, 32]");
    if (!ix->ungapped_mat) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: NULL table");
    const int kbase = (from_host && ix->kmer_alphabet > 0) ? ix->kmer_alphabet : ix->alphabet - 1;
    if (kbase < ix->alphabet - 1 || kbase > ix->alphabet) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: kmer_alphabet must be alphabet - 1 or alphabet");
    {
        double tb = 1;
        for (int i = 0; i < ix->kmer_size; i++) tb *= (double)kbase;
        if (tb > 2147483648.0) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_load_index: more than 2^31 k-mers");
    }
    if (from_host) {
        if (!ix->offsets) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: NULL table");
        if (!ix->entries6 && !(ix->entry_ids && ix->entry_pos) && ix->n_entries) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: no index entries");
        if (ix->n_entries >= 0xFFFFFFFFull) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_load_index: >= 2^32 index entries per shard");
    }
    HIP_TRY(hipSetDevice(c->device));
    pf_index_free(c);
    PfIndex *P = new PfIndex();
    P->k = ix->kmer_size;
    P->alphabet = ix->alphabet;
    P->kalph = ix->alphabet - 1;
    P->kbase = kbase;
    P->spaced = ix->spaced;
    P->pattern_len = window_pattern(P->k, P->spaced, P->pat);
    P->n3 = (uint32_t)(P->kalph * P->kalph * P->kalph);
    P->table = 1;
    for (int i = 0; i < P->k; i++) P->table *= (uint64_t)P->kbase;
    P->n_entries = from_host ? ix->n_entries : 0;
    P->has_tables = tables;
    const size_t n3 = P->n3;
    if (tables && ix->row3 < n3) { delete P; return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: row3 smaller than kalph^3"); }
#define P_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { delete P; return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    if (tables) {
        P_TRY(P->d_s3.alloc(n3 * n3 * sizeof(int16_t)));
        P_TRY(P->d_i3.alloc(n3 * n3 * sizeof(uint32_t)));
        P_TRY(hipMemcpy2D(P->d_s3.p, n3 * sizeof(int16_t), ix->score3, ix->row3 * sizeof(int16_t), n3 * sizeof(int16_t), n3, hipMemcpyHostToDevice));
        P_TRY(hipMemcpy2D(P->d_i3.p, n3 * sizeof(uint32_t), ix->index3, ix->row3 * sizeof(uint32_t), n3 * sizeof(uint32_t), n3, hipMemcpyHostToDevice));
    }
    // cumulative counts per row: #entries with score >= c is one lookup instead of a binary search in the row
    auto build_cum = [&](const int16_t *score, size_t row_stride, size_t n, DevBuf &dst, uint32_t *w_out, int32_t *min_out) -> int {
        int lo = 32767, hi = -32768;
        for (size_t r = 0; r < n; r++) {
            lo = std::min<int>(lo, score[r * row_stride + n - 1]);
            hi = std::max<int>(hi, score[r * row_stride]);
        }
        const uint32_t w = (uint32_t)(hi - lo + 2);
        std::vector<uint16_t> cum(n * (size_t)w);
        for (size_t r = 0; r < n; r++) {
            const int16_t *row = score + r * row_stride;
            for (size_t z = 1; z < n; z++)
                if (row[z] > row[z - 1]) return 1;
            size_t j = 0;   // rows are sorted descending: walk thresholds from high to low
            for (int k = (int)w - 1; k >= 0; k--) {
                const int c = lo + k;
                while (j < n && row[j] >= c) j++;
                cum[r * w + (size_t)k] = (uint16_t)j;
            }
        }
        if (upload(dst, cum, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 2;
        *w_out = w;
        *min_out = lo;
        return 0;
    };
    if (tables) {
        const int rc = build_cum(ix->score3, ix->row3, n3, P->d_cum3, &P->cum_w, &P->score_min);
        if (rc) { delete P; return fail(rc == 1 ? MMGPU_ERR_ARG : MMGPU_ERR_HIP, rc == 1 ? "mmgpu_pf_load_index: score3 rows are not sorted by descending score" : "mmgpu_pf_load_index: upload failed"); }
    }
    if (tables && P->k != 6) {      // k = 5: (2, 3), k = 7: (2, 2, 3) - KmerGenerator::setDivideStrategy
        const size_t n2 = (size_t)P->kalph * P->kalph;
        if (ix->row2 < n2) { delete P; return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: row2 smaller than kalph^2"); }
        P_TRY(P->d_s2.alloc(n2 * n2 * sizeof(int16_t)));
        P_TRY(P->d_i2.alloc(n2 * n2 * sizeof(uint32_t)));
        P_TRY(hipMemcpy2D(P->d_s2.p, n2 * sizeof(int16_t), ix->score2, ix->row2 * sizeof(int16_t), n2 * sizeof(int16_t), n2, hipMemcpyHostToDevice));
        P_TRY(hipMemcpy2D(P->d_i2.p, n2 * sizeof(uint32_t), ix->index2, ix->row2 * sizeof(uint32_t), n2 * sizeof(uint32_t), n2, hipMemcpyHostToDevice));
        const int rc = build_cum(ix->score2, ix->row2, n2, P->d_cum2, &P->cum2_w, &P->score2_min);
        if (rc) { delete P; return fail(rc == 1 ? MMGPU_ERR_ARG : MMGPU_ERR_HIP, rc == 1 ? "mmgpu_pf_load_index: score2 rows are not sorted by descending score" : "mmgpu_pf_load_index: upload failed"); }
    }
    if (from_host) {
        {
            std::vector<uint32_t> off32(P->table + 1);
            std::atomic<bool> bad_off(false);
            parallel_for((size_t)P->table + 1, [&](size_t a, size_t b) {
                for (size_t z = a; z < b; z++) {
                    if (ix->offsets[z] > ix->n_entries || (z && ix->offsets[z] < ix->offsets[z - 1])) bad_off = true;
                    off32[z] = (uint32_t)ix->offsets[z];
                }
            });
            if (bad_off) { delete P; return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: offsets not monotone / out of range"); }
            P_TRY(upload(P->d_offsets, off32, nullptr));
            P_TRY(hipDeviceSynchronize());
        }
        {
            const size_t ne = (size_t)ix->n_entries;
            std::vector<uint64_t> ent(std::max<size_t>(ne, 1));
            const uint8_t *e6 = (const uint8_t *)ix->entries6;
            std::atomic<bool> bad_ent(false);
            const uint32_t ntargets = c->db.n;
            parallel_for(ne, [&](size_t a, size_t b) {
                for (size_t e = a; e < b; e++) {
                    uint32_t id;
                    uint16_t pj;
                    if (ix->entry_ids) {
                        id = ix->entry_ids[e];
                        pj = ix->entry_pos[e];
                    } else {
                        memcpy(&id, e6 + e * 6, 4);
                        memcpy(&pj, e6 + e * 6 + 4, 2);
                    }
                    if (id >= ntargets) bad_ent = true;
                    ent[e] = (uint64_t)id | ((uint64_t)pj << 32);
                }
            });
            if (bad_ent) { delete P; return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: index entry names a target that is not loaded"); }
            P_TRY(upload(P->d_entries, ent, nullptr));
            P_TRY(hipDeviceSynchronize());
        }
    }
    P->h_mat.assign(ix->ungapped_mat, ix->ungapped_mat + ix->alphabet * ix->alphabet);
    P_TRY(upload(P->d_mat, P->h_mat, nullptr));
    P_TRY(hipDeviceSynchronize());
#undef P_TRY
    *out = P;
    return MMGPU_OK;
}


extern "C" int mmgpu_pf_load_index(mmgpu_ctx *c, const mmgpu_pf_index *ix) {
    PfIndex *P = nullptr;
    const int rc = pf_setup(c, ix, true, &P);
    if (rc != MMGPU_OK) return rc;
    HIP_TRY(pf_index_bitmap(c, P));
    HIP_TRY(pf_index_cofs(c, P));
    c->pf = P;
    return MMGPU_OK;
}

// IndexBuilder::fillDatabase on the device (ix_kernels.hip): count, scan, scatter, sort every list by seqId.
extern "C" int mmgpu_pf_build_index(mmgpu_ctx *c, const mmgpu_pf_index *ix, const int16_t *kmer_submat, int kmer_thr) {
    if (!kmer_submat) return fail(MMGPU_ERR_ARG, "mmgpu_pf_build_index: NULL argument");
    PfIndex *P = nullptr;
    int rc = pf_setup(c, ix, false, &P);
    if (rc != MMGPU_OK) return rc;
    hipStream_t s = c->stream;
#define X_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { delete P; return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    const uint64_t table = P->table;
    if (table >= 0xFFFFFFF0ull) { delete P; return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_build_index: k-mer table too large"); }
    DevBuf d_counts, d_scratch, d_chunk_off, d_chunk_tot, d_chunk_base, d_tmp, d_long, d_nlong;
    X_TRY(d_counts.alloc(table * 4));
    X_TRY(P->d_offsets.alloc((table + 1) * 4));
    X_TRY(hipMemsetAsync(d_counts.p, 0, table * 4, s));
    IxArgs A;
    memset(&A, 0, sizeof(A));
    A.t_res = c->pf_res();
    A.t_off4 = c->db.off4;
    A.t_len = c->db.len;
    A.n_targets = c->db.n;
    A.k = P->k;
    A.pattern_len = P->pattern_len;
    A.kmer_thr = kmer_thr;
    A.kalph = (uint32_t)P->kalph;
    memcpy(A.pat, P->pat, sizeof(A.pat));
    for (int a = 0; a < P->alphabet && a < 32; a++) A.self_score[a] = (int8_t)(char)kmer_submat[a * P->alphabet + a];
    A.counts = d_counts.as<uint32_t>();
    if ((int)c->db.max_len - P->pattern_len + 1 > 4096) {
        // one uint32 per residue slot of the packed target array
        uint64_t slots = 0;
        for (uint32_t l : c->h_len) slots += ((uint64_t)l + 3) / 4 * 4;
        X_TRY(d_scratch.alloc((slots + 64) * 4));
    }
    A.scratch = d_scratch.as<uint32_t>();
    X_TRY(launch_ix_target(A, false, s));
    // offsets = exclusive scan of the counts: 64 K chunks, chunk totals summed on the host
    const uint32_t CH = 65536;
    const uint32_t nch = (uint32_t)((table + CH - 1) / CH);
    std::vector<uint32_t> choff(nch + 1);
    for (uint32_t z = 0; z <= nch; z++) choff[z] = (uint32_t)std::min<uint64_t>((uint64_t)z * CH, table);
    std::vector<uint64_t> chtot(nch), chbase(nch);
    X_TRY(upload(d_chunk_off, choff, s));
    X_TRY(d_chunk_tot.alloc((size_t)nch * 8));
    X_TRY(d_chunk_base.alloc((size_t)nch * 8));
    X_TRY(launch_pf_scan(d_counts.as<uint32_t>(), d_chunk_off.as<uint32_t>(), nch, nullptr, nullptr, d_chunk_tot.as<uint64_t>(), s));
    X_TRY(hipMemcpyAsync(chtot.data(), d_chunk_tot.p, (size_t)nch * 8, hipMemcpyDeviceToHost, s));
    X_TRY(hipStreamSynchronize(s));
    uint64_t run = 0;
    for (uint32_t z = 0; z < nch; z++) { chbase[z] = run; run += chtot[z]; }
    if (run >= 0xFFFFFFFFull) { delete P; return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_build_index: >= 2^32 index entries per shard"); }
    P->n_entries = run;
    X_TRY(hipMemcpyAsync(d_chunk_base.p, chbase.data(), (size_t)nch * 8, hipMemcpyHostToDevice, s));
    X_TRY(launch_pf_scan(d_counts.as<uint32_t>(), d_chunk_off.as<uint32_t>(), nch, d_chunk_base.as<uint64_t>(), P->d_offsets.as<uint32_t>(), nullptr, s));
    X_TRY(d_tmp.alloc(std::max<uint64_t>(run, 1) * 8));
    X_TRY(P->d_entries.alloc(std::max<uint64_t>(run, 1) * 8));
    X_TRY(hipMemsetAsync(d_counts.p, 0, table * 4, s));
    A.offsets = P->d_offsets.as<uint32_t>();
    A.entries = d_tmp.as<uint64_t>();
    X_TRY(launch_ix_target(A, true, s));
    IxSortArgs S;
    S.table = table;
    S.offsets = P->d_offsets.as<uint32_t>();
    S.src = d_tmp.as<uint64_t>();
    S.dst = P->d_entries.as<uint64_t>();
    S.long_cap = (uint32_t)(run / 17 + 1);
    X_TRY(d_long.alloc((size_t)S.long_cap * 4));
    X_TRY(d_nlong.alloc(4));
    X_TRY(hipMemsetAsync(d_nlong.p, 0, 4, s));
    S.long_lists = d_long.as<uint32_t>();
    S.n_long = d_nlong.as<uint32_t>();
    X_TRY(launch_ix_sort_short(S, s));
    uint32_t n_long = 0;
    X_TRY(hipMemcpyAsync(&n_long, d_nlong.p, 4, hipMemcpyDeviceToHost, s));
    X_TRY(hipStreamSynchronize(s));
    X_TRY(launch_ix_sort_long(S, std::min(n_long, S.long_cap), s));
    X_TRY(hipStreamSynchronize(s));
#undef X_TRY
    HIP_TRY(pf_index_bitmap(c, P));
    HIP_TRY(pf_index_cofs(c, P));
    c->pf = P;
    return MMGPU_OK;
}

// test hook: the resident index as the host builder would return it
extern "C" int mmgpu_pf_debug_index(mmgpu_ctx *c, uint64_t *offsets, uint32_t *ids, uint16_t *pos, uint64_t *n_entries) {
    if (!c || !c->pf || !n_entries) return fail(MMGPU_ERR_ARG, "mmgpu_pf_debug_index: no index");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const PfIndex &P = *c->pf;
    *n_entries = P.n_entries;
    if (offsets) {
        std::vector<uint32_t> o(P.table + 1);
        HIP_TRY(hipMemcpy(o.data(), P.d_offsets.p, (P.table + 1) * 4, hipMemcpyDeviceToHost));
        for (uint64_t z = 0; z <= P.table; z++) offsets[z] = o[z];
    }
    if (ids && pos && P.n_entries) {
        std::vector<uint64_t> e(P.n_entries);
        HIP_TRY(hipMemcpy(e.data(), P.d_entries.p, P.n_entries * 8, hipMemcpyDeviceToHost));
        for (uint64_t z = 0; z < P.n_entries; z++) { ids[z] = (uint32_t)e[z]; pos[z] = (uint16_t)(e[z] >> 32); }
    }
    return MMGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------
struct mmgpu_pf_batch_t {
    mmgpu_pf_params par;
    uint32_t nq = 0, n_pos = 0;
    uint32_t max_hits = 0;     // min(par.max_hits, dbSize)
    uint32_t bins = 0, ref_bins = 0;
    uint64_t max_db_matches = 0;
    std::vector<uint32_t> q_off;
    // device: inputs
    DevBuf d_qres, d_qthr, d_qcorr, d_qoff, d_qident, d_qself;
    DevBuf d_qkind, d_qisprof, d_pscore, d_pletter, d_qrows;   // profile queries only
    DevBuf d_qncand;                                           // nucleotide searches only
    DevBuf d_sat, d_qnsat;                                     // nucleotide searches: saturated elements per query (PF_SAT_CAP each) + their number
    DevBuf d_work_list, d_work_count;                          // buckets with more than 64 candidates, per stage chunk (PfDedupArgs::big_list)
    DevBuf d_big_keys, d_big_diags;                            // max_hits > PF_MAX_HITS only: the select kernel's sort scratch
    uint32_t big_stride = 0;
    bool any_profile = false;
    // device: working set (grow-only, reused across runs)
    DevBuf d_nsim, d_qtot, d_qbase, d_list_base, d_pos_entries, d_peb, d_qentries;
    DevBuf d_pos_order;                // [n_pos] work order of the similar-k-mer kernels (launch_pf_order), built by the first run
    bool order_ready = false, has_order = false;
    DevBuf d_qtile_base, d_qntiles, d_bucket_count, d_bucket_off;
    DevBuf d_ovf_queries, d_qnseg, d_seg_start, d_qfinal, d_ovf_base, d_ovf_a, d_ovf_b, d_ovf_ocount, d_ovf_totals;
    DevBuf d_cand_small, d_cand_base, d_cand_count, d_cells, d_surv_count, d_hits, d_hit_count, d_diag_thr, d_qflags;
    std::vector<uint8_t> long_query;   // queries of 32768 residues or more that are not processed on the device (MMGPU_PF_LONG_SEQ)
    bool long_queries_on_device = false;   // ... and whether the batch holds one that is (pf_longq_kernel)
    bool exchange = false;     // prepared while a shard was set (mmgpu_pf_set_shard): d_hits holds mmgpu_pf_xhit records
    // host mirrors of the last run
    std::vector<uint64_t> q_lists, q_entries;
    std::vector<int32_t> status;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // 5, 6: inside stage 2 (first chunk)
    std::vector<hipEvent_t> chunk_ev;  // later stage chunks: after replay, after ungapped, after stage 2, after stage 3
    uint32_t last_chunks = 0;
    uint64_t last_cells = 0;
    uint64_t last_lists = 0, last_entries = 0;
    uint32_t last_tiles = 0;
    bool ran = false;
    // exchange step of a sharded run (mmgpu_pf_exchange_merge): every rank's records, and the merged lists (global ids)
    DevBuf x_recv_hits, x_recv_counts, x_hits, x_counts, x_flags, x_ident;
    std::vector<uint32_t> x_ident_host;
    int x_ranks = 0;           // ranks of the last merge; 0 = no merged lists yet
    std::vector<int32_t> x_redo_status;      // [nq] after a merge: -1, or the status of the query's unsplit re-run (pf_redo_apply)
};

namespace mmgpu {
bool pf_batch_device_lists(mmgpu_pf_batch_t *b, const mmgpu_pf_hit **hits, const uint32_t **counts, uint32_t *stride, uint32_t *nq) {
    if (!b || !b->ran || b->exchange) return false;
    *hits = b->d_hits.as<mmgpu_pf_hit>();
    *counts = b->d_hit_count.as<uint32_t>();
    *stride = b->max_hits;
    *nq = b->nq;
    return true;
}
}  // namespace mmgpu

static uint32_t reference_bins(uint64_t dbsize) {
    // QueryMatcher::initDiagonalMatcher (QueryMatcher.cpp:460-488), Util::getL2CacheSize (Util.cpp:346-361)
    uint64_t l2 = 262144;
#if defined(_SC_LEVEL2_CACHE_SIZE)
    const long v = sysconf(_SC_LEVEL2_CACHE_SIZE);
    if (v > 0) l2 = (uint64_t)v;
#endif
    for (uint32_t b = 2; b <= 1024; b <<= 1)
        if (dbsize / b < l2) return b;
    return 2048;
}

extern "C" int mmgpu_pf_prepare(mmgpu_ctx *c, const mmgpu_pf_params *par, const mmgpu_pf_query *qs, uint32_t nq,
                                mmgpu_pf_batch_t **out) {
    if (!c || !par || !out || (!qs && nq)) return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: NULL argument");
    if (!c->pf || !c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_pf_prepare: no index loaded");
    // (with kmer_score every element has a count >= 1, so a cut at 0 is the cut at 1: `mmseqs cluster` runs its first
    // prefilter with --diag-score 0 --min-ungapped-score 0, Cluster.cpp:225-227)
    if (par->min_diag_score < 1 && !par->kmer_score) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: min_diag_score must be >= 1");
    if (par->max_hits < 1) return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: max_hits must be >= 1");
    const PfIndex &P = *c->pf;
    if (!par->exact_kmer && !P.has_tables) return fail(MMGPU_ERR_STATE, "mmgpu_pf_prepare: the index was loaded without similar-k-mer tables (exact k-mer matching only)");
    if (par->nucleotide && c->shard.on) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: nucleotide searches on a sharded database are not implemented");
    if (par->kmer_score) {     // --diag-score 0
        if (c->shard.on || par->nucleotide) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: kmer_score (--diag-score 0) with a sharded database or a nucleotide search is not implemented");
        for (uint32_t i = 0; i < nq; i++)
            if (qs[i].profile_score) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: kmer_score (--diag-score 0) with profile queries is not implemented");
    }
    // a shard of a multi-GPU run answers for the whole database: list length and cache bins as in the unsplit run
    const bool exchange = c->shard.on;
    const uint64_t db_size = exchange ? c->shard.global_n : c->db.n;
    const uint32_t max_hits = (uint32_t)std::min<uint64_t>(par->max_hits, db_size);
    if (max_hits > (uint32_t)PF_MAX_HITS_BIG) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: max_hits above 131072 is not implemented");
    if (max_hits > (uint32_t)PF_MAX_HITS && exchange) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: max_hits above 4096 on a sharded database is not implemented");
    if (par->ref_bins && (par->ref_bins < 2 || par->ref_bins > 2048 || (par->ref_bins & (par->ref_bins - 1))))
        return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: ref_bins must be a power of two in [2, 2048]");
    uint32_t bins = 1;
    while ((uint64_t)bins * PF_IDS_PER_BIN < c->db.n) bins <<= 1;
    if (bins > 2048) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: more than 8M targets per shard");
    HIP_TRY(hipSetDevice(c->device));

    if (c->pf && c->pf->kbase != c->pf->kalph && !par->exact_kmer)
        return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: an index over the full alphabet (profile targets) serves exact k-mer matching only");
    mmgpu_pf_batch_t *b = new mmgpu_pf_batch_t();
    // the batch's buffers come from / go back to the context's block cache: a process prepares batch after batch (and, like the
    // drop-in's prefilter hook, the next one while this one runs), a fresh hipMalloc costs 25 - 40 ms per GB on some hosts
    for (DevBuf *d : {&b->d_qres, &b->d_qthr, &b->d_qcorr, &b->d_qoff, &b->d_qident, &b->d_qself, &b->d_qkind, &b->d_qisprof, &b->d_pscore,
                      &b->d_pletter, &b->d_qrows, &b->d_qncand, &b->d_sat, &b->d_qnsat, &b->d_big_keys, &b->d_big_diags, &b->d_nsim, &b->d_qtot,
                      &b->d_qbase, &b->d_list_base, &b->d_pos_entries, &b->d_peb, &b->d_qentries, &b->d_qtile_base, &b->d_qntiles,
                      &b->d_bucket_count, &b->d_bucket_off, &b->d_ovf_queries, &b->d_qnseg, &b->d_seg_start, &b->d_qfinal, &b->d_ovf_base,
                      &b->d_ovf_a, &b->d_ovf_b, &b->d_ovf_ocount, &b->d_ovf_totals, &b->d_cand_small, &b->d_cand_base, &b->d_cand_count,
                      &b->d_cells, &b->d_surv_count, &b->d_hits, &b->d_hit_count, &b->d_diag_thr, &b->d_qflags, &b->x_recv_hits,
                      &b->x_recv_counts, &b->x_hits, &b->x_counts, &b->x_flags, &b->x_ident, &b->d_pos_order})
        d->bind(c->cache);
    b->par = *par;
    if (b->par.min_diag_score < 1) b->par.min_diag_score = 1;
    b->nq = nq;
    b->max_hits = max_hits;
    b->bins = bins;
    b->ref_bins = par->ref_bins ? par->ref_bins : reference_bins(db_size);
    b->exchange = exchange;
    b->max_db_matches = std::max<uint64_t>(1000000, db_size) * 2;   // QueryMatcher.cpp:44-45 (dbSize of the WHOLE database)
    if (const char *e = getenv("MMGPU_PF_MAX_DB_MATCHES")) b->max_db_matches = std::max<uint64_t>(64, strtoull(e, nullptr, 10));      // tests: a small database reaches the overflow path / a shard its share
    // a shard gathers its share of a query's index entries: the unsplit run's overflow path (QueryMatcher.cpp:310-346) is taken
    // when the shares add up to max_db_matches.  If NO shard reaches max_db_matches / n_shards the sum stays below the limit:
    // a shard that reaches its share flags the query (bit 31 of its exchanged count -> MMGPU_PF_X_INEXACT_ORDER in the merge,
    // whichever elements survive) and the caller re-runs it unsplit
    if (exchange) b->max_db_matches = std::max<uint64_t>(1, b->max_db_matches / std::max<uint32_t>(1, c->shard.n_shards));
    b->q_off.assign(nq + 1, 0);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < nq; i++) {
        if (!qs[i].q || qs[i].qlen == 0) { delete b; return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: bad query"); }
        if (qs[i].qlen > 65535) { delete b; return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: query longer than 65535 (Parameters.h:271)"); }
        tot += qs[i].qlen;
        if (tot > 0x7FFFFFFFull) { delete b; return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: more than 2^31 query residues per batch"); }
        b->q_off[i + 1] = (uint32_t)tot;
    }
    b->n_pos = (uint32_t)tot;
    b->long_query.assign(nq, 0);
    // queries of 32768 residues or more (UngappedAlignment::computeLongScore for every element): on the device for amino-acid and
    // profile searches with diagonal scoring on an unsplit database (pf_longq_kernel); a shard, a nucleotide search and --diag-score 0
    // decline them here, on the host (MMGPU_PF_LONG_SEQ)
    bool any_long_on_device = false;
    for (uint32_t i = 0; i < nq; i++) {
        const bool is_long = qs[i].qlen >= 32768;
        const bool declined = exchange || par->nucleotide || par->kmer_score;
        b->long_query[i] = (is_long && declined) ? 1 : 0;
        any_long_on_device = any_long_on_device || (is_long && !declined);
    }
    b->long_queries_on_device = any_long_on_device;
    std::vector<uint8_t> qres(tot + 64, 0);    // + slack: the ungapped kernel reads whole dwords
    std::vector<int16_t> qthr(tot, -1);
    std::vector<int8_t> qcorr(tot + 64, 0);
    std::vector<uint32_t> qident(std::max<uint32_t>(nq, 1), 0xFFFFFFFFu);
    std::vector<int32_t> qself(std::max<uint32_t>(nq, 1), 0);
    // profile queries: per-position kind flag, the 20 sorted scores / letters of every position, the ungapped score rows
    bool any_prof = false;
    for (uint32_t i = 0; i < nq; i++) {
        if (!qs[i].profile_score && !qs[i].profile_index && !qs[i].profile) continue;
        if (!qs[i].profile_score || !qs[i].profile_index || !qs[i].profile || qs[i].profile_row < (uint32_t)PF_PROF_LETTERS) {
            delete b;
            return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: a profile query needs profile_score, profile_index (row >= 20) and profile");
        }
        if (P.kalph != PF_PROF_LETTERS) { delete b; return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_prepare: profile queries need the 20-letter k-mer alphabet"); }
        any_prof = true;
    }
    std::vector<uint8_t> qkind, qisprof, pletter;
    std::vector<int16_t> pscore;
    std::vector<int8_t> qrows;
    if (any_prof) {
        qkind.assign(tot + 64, 0);
        qisprof.assign(std::max<uint32_t>(nq, 1), 0);
        pscore.assign((tot + 64) * PF_PROF_LETTERS, 0);
        pletter.assign((tot + 64) * PF_PROF_LETTERS, 0);
        qrows.assign((tot + 64) * PF_PROW, 0);
    }
    b->any_profile = any_prof;
    std::atomic<bool> bad(false);
    parallel_for(nq, [&](size_t a, size_t e) {
        for (size_t i = a; i < e; i++) {
            const mmgpu_pf_query &Q = qs[i];
            const uint32_t o = b->q_off[i];
            const int L = (int)Q.qlen;
            for (int p = 0; p < L; p++) {
                if (Q.q[p] >= P.alphabet) bad = true;
                qres[o + p] = Q.q[p];
            }
            qident[i] = Q.identity_id;
            const bool prof = Q.profile != nullptr;
            const float *cbias = prof ? nullptr : Q.comp_bias;     // no composition bias for profile queries (QueryMatcher.cpp:110-114)
            if (prof) {
                qisprof[i] = 1;
                for (int p = 0; p < L; p++) {
                    qkind[o + p] = 1;
                    for (int z = 0; z < PF_PROF_LETTERS; z++) {
                        pscore[(size_t)(o + p) * PF_PROF_LETTERS + z] = Q.profile_score[(size_t)p * Q.profile_row + z];
                        const uint32_t le = Q.profile_index[(size_t)p * Q.profile_row + z];
                        if (le >= (uint32_t)P.kalph) bad = true;
                        pletter[(size_t)(o + p) * PF_PROF_LETTERS + z] = (uint8_t)le;
                        // UngappedAlignment::createProfile, profile branch: queryProfile[pos][aa] = alignment profile, X = 0
                        qrows[(size_t)(o + p) * PF_PROW + z] = Q.profile[(size_t)z * L + p];
                    }
                }
            }
            // QueryMatcher::match, QueryMatcher.cpp:255-274: per-window threshold (a query handed back to the host gets none:
            // no window, no work)
            for (int p = 0; p + P.pattern_len <= L && !b->long_query[i]; p++) {
                float bc = 0;
                bool x = false;
                for (int z = 0; z < P.k; z++) {
                    bc += cbias ? cbias[p + (short)P.pat[z]] : 0.0f;
                    if (Q.q[p + P.pat[z]] >= P.kalph) x = true;
                }
                if (x) continue;
                const short bias = (short)((bc < 0.0) ? bc - 0.5 : bc + 0.5);
                const int t0 = par->kmer_thr - bias;
                qthr[o + p] = (int16_t)(short)(t0 > 0 ? t0 : 0);
            }
            // UngappedAlignment::createProfile, UngappedAlignment.cpp:396-400
            for (int p = 0; p < L; p++) {
                float v = cbias ? cbias[p] : 0.0f;
                v = (v < 0.0) ? v / 4 - 0.5 : v / 4 + 0.5;
                qcorr[o + p] = (int8_t)(char)v;
            }
            // rescoreHits' self score (QueryMatcher.cpp:566): best ungapped segment of the query against itself
            int sc = 0, mx = 0;
            for (int p = 0; p < L && !bad; p++) {
                const int cur = prof ? (int)qrows[(size_t)(o + p) * PF_PROW + (Q.q[p] & (PF_PROW - 1))]
                                     : (int)(int8_t)(P.h_mat[(size_t)Q.q[p] * P.alphabet + Q.q[p]] + qcorr[o + p]);
                sc += cur;
                sc = sc < 0 ? 0 : sc;
                mx = sc > mx ? sc : mx;
            }
            qself[i] = mx;
        }
    });
    if (bad) { delete b; return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: query residue code >= alphabet"); }
    hipStream_t s = c->stream;
#define B_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { delete b; return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    B_TRY(upload(b->d_qres, qres, s));
    B_TRY(upload(b->d_qthr, qthr, s));
    B_TRY(upload(b->d_qcorr, qcorr, s));
    B_TRY(upload(b->d_qoff, b->q_off, s));
    B_TRY(upload(b->d_qident, qident, s));
    B_TRY(upload(b->d_qself, qself, s));
    if (any_prof) {
        B_TRY(upload(b->d_qkind, qkind, s));
        B_TRY(upload(b->d_qisprof, qisprof, s));
        B_TRY(upload(b->d_pscore, pscore, s));
        B_TRY(upload(b->d_pletter, pletter, s));
        B_TRY(upload(b->d_qrows, qrows, s));
    }
    const size_t np = std::max<size_t>(tot, 1), nqq = std::max<uint32_t>(nq, 1);
    B_TRY(b->d_nsim.alloc(np * 4));
    B_TRY(b->d_list_base.alloc((np + 1) * 4));
    B_TRY(b->d_pos_entries.alloc(np * 4));
    B_TRY(b->d_peb.alloc((np + 1) * 4));
    B_TRY(b->d_qtot.alloc(nqq * 8));
    B_TRY(b->d_qbase.alloc(nqq * 8));
    B_TRY(b->d_qentries.alloc(nqq * 4));
    B_TRY(b->d_qtile_base.alloc(nqq * 4));
    B_TRY(b->d_qntiles.alloc(nqq * 4));
    B_TRY(b->d_bucket_count.alloc((size_t)nqq * bins * 4));
    B_TRY(b->d_bucket_off.alloc(((size_t)nqq + 1) * 4));
    B_TRY(b->d_cand_base.alloc(((size_t)nqq * bins + 1) * 4));
    B_TRY(b->d_cand_count.alloc((size_t)nqq * bins * 4));
    B_TRY(b->d_cand_small.alloc((size_t)nqq * bins * PF_CAND0 * sizeof(PfCand)));
    B_TRY(b->d_cells.alloc((size_t)nqq * 8));
    B_TRY(b->d_surv_count.alloc(nqq * 4));
    B_TRY(b->d_hits.alloc((size_t)nqq * max_hits * (exchange ? sizeof(mmgpu_pf_xhit) : sizeof(mmgpu_pf_hit))));
    B_TRY(b->d_hit_count.alloc(nqq * 4));
    B_TRY(b->d_diag_thr.alloc(nqq * 4));
    B_TRY(b->d_qflags.alloc(nqq * 4));
    if (par->nucleotide || par->kmer_score) B_TRY(b->d_qncand.alloc(nqq * 4));
    if (par->nucleotide) {
        B_TRY(b->d_sat.alloc((size_t)nqq * PF_SAT_CAP * sizeof(PfCand)));
        B_TRY(b->d_qnsat.alloc(nqq * 4));
    }
    if (max_hits > (uint32_t)PF_MAX_HITS) {
        b->big_stride = 1;
        while (b->big_stride < max_hits) b->big_stride <<= 1;
        B_TRY(b->d_big_keys.alloc((size_t)nqq * b->big_stride * 8));
        B_TRY(b->d_big_diags.alloc((size_t)nqq * b->big_stride * 2));
    }
    for (auto &e : b->ev) B_TRY(hipEventCreate(&e));
    B_TRY(hipStreamSynchronize(s));
#undef B_TRY
    *out = b;
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_run(mmgpu_ctx *c, mmgpu_pf_batch_t *b) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_pf_run: NULL argument");
    if (!c->pf) return fail(MMGPU_ERR_STATE, "mmgpu_pf_run: no index loaded");
    HIP_TRY(hipSetDevice(c->device));
    PfIndex &P = *c->pf;
    P.w_owner = b;
    hipStream_t s = c->stream;
    const uint32_t nq = b->nq;
    b->status.assign(nq, MMGPU_PF_OK);
    for (uint32_t i = 0; i < nq; i++)
        if (b->long_query[i]) b->status[i] = MMGPU_PF_LONG_SEQ;
    b->q_lists.assign(nq, 0);
    b->q_entries.assign(nq, 0);
    if (nq == 0) { b->ran = true; return MMGPU_OK; }
    HIP_TRY(hipEventRecord(b->ev[0], s));

    // ---- stage 0: similar k-mers and their index lists, in the reference's order ----
    PfKmerArgs K;
    memset(&K, 0, sizeof(K));
    K.q_res = b->d_qres.as<uint8_t>();
    K.q_thr = b->d_qthr.as<int16_t>();
    K.n_pos = b->n_pos;
    K.exact = b->par.exact_kmer ? 1 : 0;
    if (b->any_profile) {
        K.q_kind = b->d_qkind.as<uint8_t>();
        K.prof_score = b->d_pscore.as<int16_t>();
        K.prof_letter = b->d_pletter.as<uint8_t>();
    }
    memcpy(K.pat, P.pat, sizeof(K.pat));
    K.kalph = (uint32_t)P.kalph;
    K.kbase = (uint32_t)P.kbase;
    K.n3 = P.n3;
    K.s3 = P.d_s3.as<int16_t>();
    K.i3 = P.d_i3.as<uint32_t>();
    K.offsets = P.d_offsets.as<uint32_t>();
    K.nonempty = P.use_bitmap ? P.d_nonempty.as<uint32_t>() : nullptr;
    K.cofs = P.use_cofs ? P.d_cofs.as<uint4>() : nullptr;
    K.cum3 = P.d_cum3.as<uint16_t>();
    K.cum_w = P.cum_w;
    K.score_min = P.score_min;
    K.k = P.k;
    K.s2 = P.d_s2.as<int16_t>();
    K.i2 = P.d_i2.as<uint32_t>();
    K.cum2 = P.d_cum2.as<uint16_t>();
    K.cum2_w = P.cum2_w;
    K.score2_min = P.score2_min;
    K.nsim = b->d_nsim.as<uint32_t>();
    // work order of the positions (pf_order.hip): built with the batch's first run
    if (!b->order_ready && !K.exact) {
        HIP_TRY(b->d_pos_order.alloc((size_t)b->n_pos * 4));
        HIP_TRY(launch_pf_order(K, b->d_pos_order.as<uint32_t>(), c->cache, s));
        b->has_order = true;
        b->order_ready = true;
    }
    K.order = b->has_order ? b->d_pos_order.as<uint32_t>() : nullptr;
    HIP_TRY(launch_pf_kmers(K, false, s));
    HIP_TRY(launch_pf_scan(b->d_nsim.as<uint32_t>(), b->d_qoff.as<uint32_t>(), nq, nullptr, nullptr, b->d_qtot.as<uint64_t>(), s));
    HIP_TRY(hipMemcpyAsync(b->q_lists.data(), b->d_qtot.p, nq * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<uint64_t> qbase(nq);
    uint64_t total_lists = 0;
    for (uint32_t i = 0; i < nq; i++) {
        qbase[i] = total_lists;
        total_lists += b->q_lists[i];
    }
    if (total_lists >= 0xFFFFFFFFull) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_run: >= 2^32 similar k-mers in one batch; use smaller batches");
    HIP_TRY(hipMemcpyAsync(b->d_qbase.p, qbase.data(), nq * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_pf_scan(b->d_nsim.as<uint32_t>(), b->d_qoff.as<uint32_t>(), nq, b->d_qbase.as<uint64_t>(), b->d_list_base.as<uint32_t>(), nullptr, s));
    b->last_lists = total_lists;
    HIP_TRY(P.w_lists.reserve(std::max<uint64_t>(total_lists, 1) * sizeof(PfList)));
    K.list_base = b->d_list_base.as<uint32_t>();
    K.lists = P.w_lists.as<PfList>();
    K.pos_entries = b->d_pos_entries.as<uint32_t>();
    HIP_TRY(launch_pf_kmers(K, true, s));
    HIP_TRY(launch_pf_scan(b->d_pos_entries.as<uint32_t>(), b->d_qoff.as<uint32_t>(), nq, nullptr, b->d_peb.as<uint32_t>(), b->d_qtot.as<uint64_t>(), s));
    HIP_TRY(hipMemcpyAsync(b->q_entries.data(), b->d_qtot.p, nq * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));   // qbase (pageable) has been consumed, q_entries is valid
    HIP_TRY(hipEventRecord(b->ev[1], s));

    // ---- queries on the reference's overflow path (QueryMatcher.cpp:310-346): segment boundaries ----
    std::vector<uint32_t> ovf_q;
    for (uint32_t i = 0; i < nq; i++)
        if (b->q_entries[i] >= b->max_db_matches) ovf_q.push_back(i);
    std::vector<uint32_t> h_nseg(nq, 0);
    uint32_t max_seg = 0;
    if (!ovf_q.empty()) {
        HIP_TRY(b->d_ovf_queries.reserve(ovf_q.size() * 4));
        HIP_TRY(b->d_qnseg.reserve((size_t)nq * 4));
        HIP_TRY(b->d_qfinal.reserve((size_t)nq * 4));
        HIP_TRY(b->d_seg_start.reserve((size_t)nq * (PF_MAX_SEG + 2) * 4));
        HIP_TRY(hipMemcpyAsync(b->d_ovf_queries.p, ovf_q.data(), ovf_q.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(b->d_qnseg.p, 0, (size_t)nq * 4, s));
        HIP_TRY(hipMemsetAsync(b->d_qfinal.p, 0, (size_t)nq * 4, s));
        std::vector<uint32_t> e32(nq);
        for (uint32_t i = 0; i < nq; i++) e32[i] = (uint32_t)std::min<uint64_t>(b->q_entries[i], 0xFFFFFFFFull);
        HIP_TRY(hipMemcpyAsync(b->d_qentries.p, e32.data(), (size_t)nq * 4, hipMemcpyHostToDevice, s));
        PfSegArgs G;
        G.ovf_queries = b->d_ovf_queries.as<uint32_t>();
        G.n_ovf = (uint32_t)ovf_q.size();
        G.q_off = b->d_qoff.as<uint32_t>();
        G.list_base = b->d_list_base.as<uint32_t>();
        G.pos_entry_base = b->d_peb.as<uint32_t>();
        G.lists = P.w_lists.as<PfList>();
        G.cap = b->max_db_matches;
        G.seg_start = b->d_seg_start.as<uint32_t>();
        G.q_nseg = b->d_qnseg.as<uint32_t>();
        G.q_final = b->d_qfinal.as<uint32_t>();
        G.q_entries = b->d_qentries.as<uint32_t>();
        HIP_TRY(launch_pf_segments(G, s));
        HIP_TRY(hipMemcpyAsync(h_nseg.data(), b->d_qnseg.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        bool rewrite = false;
        std::vector<uint32_t> keep;
        for (uint32_t i : ovf_q) {
            // (--diag-score 0 merges the segments by score, QueryMatcher.cpp:514-533: not on the device either)
            if (h_nseg[i] == 0 || h_nseg[i] > (uint32_t)PF_MAX_SEG || b->q_entries[i] >= 0xF0000000ull || b->par.kmer_score) {
                h_nseg[i] = 0;      // more flushes than the device emulates: the host runs the reference for this query
                b->status[i] = MMGPU_PF_OVERFLOW;
                rewrite = true;
            } else {
                keep.push_back(i);
                max_seg = std::max(max_seg, h_nseg[i]);
            }
        }
        ovf_q.swap(keep);
        if (rewrite) {
            HIP_TRY(hipMemcpyAsync(b->d_qnseg.p, h_nseg.data(), (size_t)nq * 4, hipMemcpyHostToDevice, s));
            if (!ovf_q.empty()) HIP_TRY(hipMemcpyAsync(b->d_ovf_queries.p, ovf_q.data(), ovf_q.size() * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    }

    // ---- tiles ----
    std::vector<uint32_t> qent(nq), qtb(nq), qnt(nq);      // (the per-tile lists are written on the device: pf_tiles_kernel)
    std::vector<uint64_t> qebase(nq);
    uint64_t total_entries = 0, tiles_so_far = 0;
    for (uint32_t i = 0; i < nq; i++) {
        uint64_t e = b->q_entries[i];
        if (b->status[i] != MMGPU_PF_OK) e = 0;   // not processed on the device
        qent[i] = (uint32_t)e;
        qebase[i] = total_entries;
        total_entries += e;
        qtb[i] = (uint32_t)tiles_so_far;
        qnt[i] = (uint32_t)((e + PF_T - 1) / PF_T);
        tiles_so_far += qnt[i];
    }
    if (tiles_so_far >= 0xFFFFFFFFull) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_run: >= 2^32 tiles in one batch; use smaller batches");
    // (total_entries may pass 2^32: the 32-bit bases derived from it - cand_base, cand_origin - only ever meet as differences
    // inside one stage chunk, whose entries are bounded below)
    const uint32_t n_tiles = (uint32_t)tiles_so_far;
    b->last_tiles = n_tiles;
    b->last_entries = total_entries;
    const uint32_t B = b->bins;
    HIP_TRY(hipMemcpyAsync(b->d_qentries.p, qent.data(), nq * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->d_qtile_base.p, qtb.data(), nq * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->d_qntiles.p, qnt.data(), nq * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->d_qbase.p, qebase.data(), nq * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(P.w_tile_q.reserve(std::max<size_t>(n_tiles, 1) * 4));
    HIP_TRY(P.w_tile_idx.reserve(std::max<size_t>(n_tiles, 1) * 4));
    if (n_tiles) HIP_TRY(launch_pf_tiles(b->d_qtile_base.as<uint32_t>(), b->d_qntiles.as<uint32_t>(), nq, P.w_tile_q.as<uint32_t>(), P.w_tile_idx.as<uint32_t>(), s));
    HIP_TRY(P.w_split.reserve(std::max<size_t>(n_tiles, 1) * PF_T * sizeof(uint32_t)));
    HIP_TRY(P.w_split_hi.reserve(std::max<size_t>(n_tiles, 1) * PF_T));
    HIP_TRY(P.w_bin_off.reserve(std::max<size_t>(n_tiles, 1) * (B + 1) * sizeof(uint16_t)));
    // Stages 2 and 3 run over CHUNKS of consecutive queries: the candidate / survivor arrays are entry-sized (every entry
    // of a bin can be a candidate), 32 B per entry, of which ~1 % is touched - 115 GB for a 10 000-query batch against
    // 1 M targets.  A chunk is the longest run of queries whose entries fit MMGPU_PF_STAGE_GB (default 16 GB for both
    // arrays); the arrays are re-used from chunk to chunk (same stream, so the order is the program order).
    std::vector<uint32_t> chunk_first;
    uint64_t chunk_max_entries = 0;
    {
        // (one chunk per ~40 GB on a 288 GB device: every chunk ends with the tail of its replay grid and four small launches -
        // 7 chunks of 16 GB cost the 10 000-query batch 1 ms more than 3 of 40, profiles/r05_exp_pf_stage_chunks.txt)
        double gb = 16.0;
        {
            size_t mem_free = 0, mem_total = 0;
            if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_total >= (192ull << 30) && mem_free >= (120ull << 30)) gb = 40.0;
        }
        if (const char *e = getenv("MMGPU_PF_STAGE_GB")) gb = atof(e);
        // (a chunk's entries stay below 2^32 whatever the budget: the candidate bases are 32-bit differences inside a chunk)
        const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>((uint64_t)(gb * 1073741824.0 / (2.0 * sizeof(PfCand))), 1), 0xF0000000ull);
        uint64_t run = 0;
        for (uint32_t i = 0; i < nq; i++) {
            if (i == 0 || run + qent[i] > cap) {
                chunk_first.push_back(i);
                run = 0;
            }
            run += qent[i];
            chunk_max_entries = std::max(chunk_max_entries, run);
        }
        chunk_first.push_back(nq);
    }
    const uint32_t n_chunks = (uint32_t)chunk_first.size() - 1;
    b->last_chunks = n_chunks;
    while (b->chunk_ev.size() < (size_t)4 * n_chunks) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        b->chunk_ev.push_back(e);
    }
    HIP_TRY(P.w_cand.reserve(std::max<uint64_t>(chunk_max_entries, 1) * sizeof(PfCand)));
    HIP_TRY(P.w_surv.reserve(std::max<uint64_t>(chunk_max_entries, 1) * sizeof(PfCand)));
    HIP_TRY(hipMemsetAsync(b->d_bucket_count.p, 0, (size_t)nq * B * 4, s));
    HIP_TRY(hipMemsetAsync(b->d_surv_count.p, 0, (size_t)nq * 4, s));
    HIP_TRY(hipMemsetAsync(b->d_cells.p, 0, (size_t)nq * 8, s));
    HIP_TRY(hipMemsetAsync(b->d_qflags.p, 0, (size_t)nq * 4, s));
    if (b->par.nucleotide || b->par.kmer_score) HIP_TRY(hipMemsetAsync(b->d_qncand.p, 0, (size_t)nq * 4, s));
    if (b->par.nucleotide) HIP_TRY(hipMemsetAsync(b->d_qnsat.p, 0, (size_t)nq * 4, s));

    // ---- stage 1: gather + stable split ----
    PfSplitArgs SA;
    SA.tile_q = P.w_tile_q.as<uint32_t>();
    SA.tile_idx = P.w_tile_idx.as<uint32_t>();
    SA.q_off = b->d_qoff.as<uint32_t>();
    SA.q_entries = b->d_qentries.as<uint32_t>();
    SA.pos_entry_base = b->d_peb.as<uint32_t>();
    SA.list_base = b->d_list_base.as<uint32_t>();
    SA.lists = P.w_lists.as<PfList>();
    SA.idx_entries = P.d_entries.as<uint64_t>();
    SA.bins = B;
    SA.split = P.w_split.as<uint32_t>();
    SA.split_hi = P.w_split_hi.as<uint8_t>();
    SA.bin_off = P.w_bin_off.as<uint16_t>();
    SA.bucket_count = b->d_bucket_count.as<uint32_t>();
    HIP_TRY(launch_pf_split(SA, n_tiles, s));
    // cand_base[q * B + bin] = entries of earlier queries + entries of earlier bins of this query
    {
        std::vector<uint32_t> boff(nq + 1);
        for (uint32_t i = 0; i <= nq; i++) boff[i] = i * B;
        HIP_TRY(hipMemcpyAsync(b->d_bucket_off.p, boff.data(), (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(launch_pf_scan(b->d_bucket_count.as<uint32_t>(), b->d_bucket_off.as<uint32_t>(), nq, b->d_qbase.as<uint64_t>(),
                               b->d_cand_base.as<uint32_t>(), nullptr, s));
        HIP_TRY(hipStreamSynchronize(s));   // host vectors above are pageable
    }
    HIP_TRY(hipEventRecord(b->ev[2], s));

    // ---- stage 2: replay, ungapped score, best element per target ----
    PfDedupArgs D;
    D.n_queries = nq;
    D.bins = B;
    D.q_tile_base = b->d_qtile_base.as<uint32_t>();
    D.q_ntiles = b->d_qntiles.as<uint32_t>();
    D.split = P.w_split.as<uint32_t>();
    D.split_hi = P.w_split_hi.as<uint8_t>();
    D.bin_off = P.w_bin_off.as<uint16_t>();
    D.cand_base = b->d_cand_base.as<uint32_t>();
    D.cand = P.w_cand.as<PfCand>();
    D.surv = P.w_surv.as<PfCand>();
    D.surv_count = b->d_surv_count.as<uint32_t>();
    D.q_off = b->d_qoff.as<uint32_t>();
    D.q_res = b->d_qres.as<uint8_t>();
    D.q_corr = b->d_qcorr.as<int8_t>();
    D.nucl = b->par.nucleotide ? 1 : 0;
    D.sort_cap = (uint32_t)(std::max<uint64_t>(1000000, b->exchange ? c->shard.global_n : c->db.n) / 2);     // foundDiagonalsSize / 2 (QueryMatcher.cpp:44,146)
    D.q_ncand = (b->par.nucleotide || b->par.kmer_score) ? b->d_qncand.as<uint32_t>() : nullptr;
    D.sat = b->par.nucleotide ? b->d_sat.as<PfCand>() : nullptr;
    D.q_nsat = b->par.nucleotide ? b->d_qnsat.as<uint32_t>() : nullptr;
    D.sat_cap = (uint32_t)PF_SAT_CAP;
    D.q_rows = b->any_profile ? b->d_qrows.as<int8_t>() : nullptr;
    D.q_isprof = b->any_profile ? b->d_qisprof.as<uint8_t>() : nullptr;
    D.mat = P.d_mat.as<int8_t>();
    D.alphabet = P.alphabet;
    D.t_res = c->pf_res();
    D.t_off4 = c->db.off4;
    D.t_len = c->db.len;
    D.min_diag_score = b->par.min_diag_score;
    D.cand_count = b->d_cand_count.as<uint32_t>();
    D.cand_small = b->d_cand_small.as<PfCand>();
    D.q_nseg = ovf_q.empty() ? nullptr : b->d_qnseg.as<uint32_t>();
    D.seg_start = ovf_q.empty() ? nullptr : b->d_seg_start.as<uint32_t>();
    D.cell_counter = b->d_cells.as<uint64_t>();
    D.q_flags = b->d_qflags.as<uint32_t>();
    D.ref_bins = b->ref_bins;
    D.big_list = D.big_count = nullptr;
    {   // the work list of the larger buckets: one list (re-used from chunk to chunk, same stream) and one counter per chunk
        uint64_t chunk_buckets = 0;
        for (uint32_t ch = 0; ch < n_chunks; ch++) chunk_buckets = std::max<uint64_t>(chunk_buckets, (uint64_t)(chunk_first[ch + 1] - chunk_first[ch]) * B);
        if (!b->par.kmer_score && chunk_buckets > 0 && chunk_buckets < 0xFFFFFFFFull) {
            HIP_TRY(b->d_work_list.reserve(chunk_buckets * 4));
            HIP_TRY(b->d_work_count.reserve((size_t)n_chunks * 4));
            HIP_TRY(hipMemsetAsync(b->d_work_count.p, 0, (size_t)n_chunks * 4, s));
            D.big_list = b->d_work_list.as<uint32_t>();
        }
    }
    // the flushes of the overflow path: per-query bases relative to the first overflow query of the same chunk
    std::vector<uint32_t> ovf_chunk_lo(n_chunks + 1, 0);
    PfOvfArgs O;
    if (!ovf_q.empty()) {
        std::vector<uint64_t> obase(ovf_q.size());
        uint64_t oe_max = 0;
        size_t z = 0;
        for (uint32_t ch = 0; ch < n_chunks; ch++) {
            ovf_chunk_lo[ch] = (uint32_t)z;
            uint64_t oe = 0;
            for (; z < ovf_q.size() && ovf_q[z] < chunk_first[ch + 1]; z++) { obase[z] = oe; oe += qent[ovf_q[z]]; }
            oe_max = std::max(oe_max, oe);
        }
        ovf_chunk_lo[n_chunks] = (uint32_t)ovf_q.size();
        HIP_TRY(b->d_ovf_base.reserve(obase.size() * 8));
        HIP_TRY(b->d_ovf_a.reserve(std::max<uint64_t>(oe_max, 1) * sizeof(PfOvfElem)));
        HIP_TRY(b->d_ovf_b.reserve(std::max<uint64_t>(oe_max, 1) * sizeof(PfOvfElem)));
        HIP_TRY(b->d_ovf_ocount.reserve(ovf_q.size() * (size_t)B * 4));
        HIP_TRY(b->d_ovf_totals.reserve(ovf_q.size() * (size_t)(PF_MAX_SEG + 2) * 4));
        HIP_TRY(hipMemcpyAsync(b->d_ovf_base.p, obase.data(), obase.size() * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(b->d_ovf_ocount.p, 0, ovf_q.size() * (size_t)B * 4, s));
        HIP_TRY(hipMemsetAsync(b->d_ovf_totals.p, 0, ovf_q.size() * (size_t)(PF_MAX_SEG + 2) * 4, s));
        HIP_TRY(hipStreamSynchronize(s));   // obase is pageable
        O.q_final = b->d_qfinal.as<uint32_t>();
    }

    // ---- stage 3: top max_hits per query ----
    PfSelectArgs S;
    S.surv = P.w_surv.as<PfCand>();
    S.surv_count = b->d_surv_count.as<uint32_t>();
    S.q_identity = b->d_qident.as<uint32_t>();
    S.q_self_score = b->d_qself.as<int32_t>();
    S.max_hits = b->max_hits;
    S.min_diag_score = b->par.min_diag_score;
    S.ref_bins = b->ref_bins;
    S.hits = b->d_hits.as<mmgpu_pf_hit>();
    S.hit_stride = b->max_hits;
    S.hit_count = b->d_hit_count.as<uint32_t>();
    S.q_diag_thr = b->d_diag_thr.as<uint32_t>();
    S.cand_base = b->d_cand_base.as<uint32_t>();
    S.bins = B;
    S.xhits = nullptr;
    S.global_ids = nullptr;
    S.q_nseg = nullptr;
    S.q_flags = b->d_qflags.as<uint32_t>();
    S.nucl = b->par.nucleotide ? 1 : 0;
    S.kmer_score = b->par.kmer_score ? 1 : 0;
    {   // foundDiagonalsSize / 2 over the WHOLE database (QueryMatcher.cpp:44,188); a shard sees its share of the candidates:
        // if no shard reaches cap / n_shards the total stays below cap, so a shard flags at its share (the query is then re-run unsplit)
        const uint64_t db_all = b->exchange ? c->shard.global_n : c->db.n;
        uint64_t cap = std::max<uint64_t>(1000000, db_all) / 2;
        if (b->exchange) cap = std::max<uint64_t>(1, cap / std::max<uint32_t>(1, c->shard.n_shards));
        if (const char *e = getenv("MMGPU_PF_SORT_CAP")) cap = strtoull(e, nullptr, 10);     // tests: a small database reaches the branch
        S.cand_cap = (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFFFull);
        S.cand_count = (b->par.nucleotide || b->par.kmer_score) ? nullptr : b->d_cand_count.as<uint32_t>();
    }
    S.big_keys = b->big_stride ? b->d_big_keys.as<uint64_t>() : nullptr;
    S.big_diags = b->big_stride ? b->d_big_diags.as<uint16_t>() : nullptr;
    S.big_stride = b->big_stride;
    S.q_off = S.peb = S.list_base = nullptr;
    S.lists = nullptr;
    if (b->exchange) {
        if (!c->shard.on) return fail(MMGPU_ERR_STATE, "mmgpu_pf_run: the batch was prepared for a sharded run, but no shard is set");
        S.xhits = b->d_hits.as<mmgpu_pf_xhit>();
        S.hits = nullptr;
        S.global_ids = c->shard.d_global_ids.as<uint32_t>();
        S.q_nseg = ovf_q.empty() ? nullptr : b->d_qnseg.as<uint32_t>();
        S.q_off = b->d_qoff.as<uint32_t>();
        S.peb = b->d_peb.as<uint32_t>();
        S.list_base = b->d_list_base.as<uint32_t>();
        S.lists = P.w_lists.as<PfList>();
    }
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        const uint32_t q0 = chunk_first[ch], cn = chunk_first[ch + 1] - q0;
        hipEvent_t *cev = &b->chunk_ev[(size_t)4 * ch];
        // ---- stage 2: replay, ungapped score, best element per target ----
        D.q_first = q0;
        D.n_queries = cn;
        D.cand_origin = (uint32_t)qebase[q0];
        if (D.big_list) D.big_count = b->d_work_count.as<uint32_t>() + ch;
        if (b->par.kmer_score) HIP_TRY(launch_pf_count(D, cev[0], cev[1], s));
        else {
            HIP_TRY(launch_pf_dedup(D, cev[0], cev[1], s));
            // candidates on targets of 32768 residues or more: computeLongScore and the batches of scoreDiagonalAndUpdateHits (a shard
            // leaves such queries flagged: the unsplit re-run scores them)
            if ((c->db.max_len >= 32768u || b->long_queries_on_device) && !b->exchange && !b->par.nucleotide) HIP_TRY(launch_pf_long(D, b->long_queries_on_device, s));
        }
        const uint32_t z0 = ovf_chunk_lo[ch], z1 = ovf_chunk_lo[ch + 1];
        if (z1 > z0) {
            // one launch per flush (the total kept after flush k decides what flush k+1 does)
            O.D = D;
            O.ovf_queries = b->d_ovf_queries.as<uint32_t>() + z0;
            O.n_ovf = z1 - z0;
            O.ovf_base = b->d_ovf_base.as<uint64_t>() + z0;
            O.buf_a = b->d_ovf_a.as<PfOvfElem>();
            O.buf_b = b->d_ovf_b.as<PfOvfElem>();
            O.o_count = b->d_ovf_ocount.as<uint32_t>() + (size_t)z0 * B;
            O.totals = b->d_ovf_totals.as<uint32_t>() + (size_t)z0 * (PF_MAX_SEG + 2);
            uint32_t chunk_seg = 0;
            for (uint32_t zz = z0; zz < z1; zz++) chunk_seg = std::max(chunk_seg, h_nseg[ovf_q[zz]]);
            for (uint32_t step = 1; step <= chunk_seg + 1; step++) {
                O.step = step;
                HIP_TRY(launch_pf_overflow(O, s));
            }
        }
        HIP_TRY(hipEventRecord(cev[2], s));
        // ---- stage 3: top max_hits per query ----
        S.q_first = q0;
        S.cand_origin = D.cand_origin;
        HIP_TRY(launch_pf_select(S, cn, s));
        HIP_TRY(hipEventRecord(cev[3], s));
    }
    HIP_TRY(hipEventRecord(b->ev[4], s));
    b->ran = true;
    return MMGPU_OK;
}

// Ties between saturated diagonals of one target in a nucleotide query with more than 16 saturated elements: the reference's
// std::sort replayed over the exported elements (sat_ties.h); only the diagonal of the target's hit can differ from what the device
// chose (score and count are equal by the definition of the tie).
static int pf_resolve_saturated_ties(mmgpu_ctx *c, mmgpu_pf_batch_t *b, uint32_t q, uint32_t n, mmgpu_pf_hit *hits, uint32_t n_hits) {
    std::vector<PfCand> el(n);
    hipError_t e = hipMemcpy(el.data(), b->d_sat.as<PfCand>() + (size_t)q * PF_SAT_CAP, (size_t)n * sizeof(PfCand), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(MMGPU_ERR_HIP, std::string("mmgpu_pf_fetch: ") + hipGetErrorString(e));
    mmgpu::resolve_saturated_ties(el, b->ref_bins - 1);
    uint32_t prev = 0xFFFFFFFFu;
    for (size_t i = 0; i < el.size(); i++) {      // the first element of every target now carries the diagonal the reference keeps
        if (el[i].id == prev) continue;
        prev = el[i].id;
        for (uint32_t k = 0; k < n_hits; k++)
            if (hits[k].id == el[i].id) {
                hits[k].diagonal = el[i].diag;
                break;
            }
    }
    (void)c;
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_fetch(mmgpu_ctx *c, mmgpu_pf_batch_t *b, mmgpu_pf_hit *hits, uint32_t hit_stride, uint32_t *counts,
                              int32_t *status, mmgpu_pf_qstat *stats) {
    if (!c || !b || ((!hits || !counts) && b->nq)) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_pf_fetch: batch was never run");
    if (b->exchange) return fail(MMGPU_ERR_STATE, "mmgpu_pf_fetch: exchange batch of a sharded run (use mmgpu_pf_fetch_exchange + mmgpu_pf_merge_exchange)");
    if (hit_stride < b->max_hits) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch: hit_stride smaller than min(max_hits, dbSize)");
    const uint32_t nq = b->nq;
    if (nq == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<uint32_t> thr(nq), surv(nq), flags(nq);
    HIP_TRY(hipMemcpyAsync(flags.data(), b->d_qflags.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpy2DAsync(hits, (size_t)hit_stride * sizeof(mmgpu_pf_hit), b->d_hits.p, (size_t)b->max_hits * sizeof(mmgpu_pf_hit),
                             (size_t)b->max_hits * sizeof(mmgpu_pf_hit), nq, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(counts, b->d_hit_count.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(thr.data(), b->d_diag_thr.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(surv.data(), b->d_surv_count.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    std::vector<uint64_t> match_sum;      // --diag-score 0: statistics_t::doubleMatches (sum of the match counts)
    if (b->par.kmer_score && stats) {
        match_sum.resize(nq);
        HIP_TRY(hipMemcpyAsync(match_sum.data(), b->d_cells.p, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
    }
    std::vector<uint32_t> nsat;
    if (b->par.nucleotide && b->d_qnsat.p) {
        nsat.resize(nq);
        HIP_TRY(hipMemcpyAsync(nsat.data(), b->d_qnsat.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < nq; i++) {
        if ((flags[i] & 1u) && b->status[i] == MMGPU_PF_OK) b->status[i] = MMGPU_PF_LONG_SEQ;
        if ((flags[i] & 2u) && b->status[i] == MMGPU_PF_OK) b->status[i] = MMGPU_PF_SAT_TIE;
        // nucleotide searches: a tie between saturated diagonals of one target (pf_keepmax_nucl_kernel).  Up to 16 saturated elements
        // in the query the device's choice is the reference's (insertion sort = stable); beyond, the reference's std::sort is
        // replayed over the exported elements and the hit's diagonal corrected
        if ((flags[i] & 4u) && b->status[i] == MMGPU_PF_OK && i < nsat.size() && nsat[i] > 16u) {
            if (nsat[i] > (uint32_t)PF_SAT_CAP) b->status[i] = MMGPU_PF_SAT_TIE;
            else {
                const int rc = pf_resolve_saturated_ties(c, b, i, nsat[i], hits + (size_t)i * hit_stride, counts[i]);
                if (rc != MMGPU_OK) return rc;
            }
        }
        if (b->status[i] != MMGPU_PF_OK) counts[i] = 0;
        if (status) status[i] = b->status[i];
        if (stats) {
            stats[i].db_matches = b->q_entries[i];
            stats[i].kmer_list_len = b->q_lists[i];
            stats[i].double_hits = match_sum.empty() ? surv[i] : match_sum[i];
            stats[i].diag_thr = thr[i];
        }
    }
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_fetch_device(mmgpu_ctx *c, mmgpu_pf_batch_t *b, void *d_hits, uint32_t hit_stride, void *d_counts) {
    if (!c || !b || ((!d_hits || !d_counts) && b->nq)) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch_device: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_pf_fetch_device: batch was never run");
    if (b->exchange) return fail(MMGPU_ERR_STATE, "mmgpu_pf_fetch_device: exchange batch of a sharded run (use mmgpu_pf_fetch_exchange)");
    if (hit_stride < b->max_hits) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch_device: hit_stride smaller than min(max_hits, dbSize)");
    if (b->nq == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy2DAsync(d_hits, (size_t)hit_stride * sizeof(mmgpu_pf_hit), b->d_hits.p, (size_t)b->max_hits * sizeof(mmgpu_pf_hit),
                             (size_t)b->max_hits * sizeof(mmgpu_pf_hit), b->nq, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_counts, b->d_hit_count.p, (size_t)b->nq * 4, hipMemcpyDeviceToDevice, c->stream));
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_merge_splits(mmgpu_ctx *c, const void *d_hits, const void *d_counts, uint32_t n_splits, uint32_t nq,
                                     uint32_t stride, const uint32_t *id_offsets, void *d_out_hits, void *d_out_counts) {
    if (!c || ((!d_hits || !d_counts || !d_out_hits || !d_out_counts) && nq) || !id_offsets)
        return fail(MMGPU_ERR_ARG, "mmgpu_pf_merge_splits: NULL argument");
    if (n_splits < 1 || n_splits > 64) return fail(MMGPU_ERR_ARG, "mmgpu_pf_merge_splits: n_splits must be in [1, 64]");
    if ((uint64_t)n_splits * stride > PF_MERGE_CAP) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_merge_splits: more than 8192 hits per query");
    HIP_TRY(hipSetDevice(c->device));
    PfMergeArgs A;
    A.hits = (const mmgpu_pf_hit *)d_hits;
    A.counts = (const uint32_t *)d_counts;
    A.n_splits = n_splits;
    A.nq = nq;
    A.stride = stride;
    memset(A.id_offset, 0, sizeof(A.id_offset));
    for (uint32_t i = 0; i < n_splits; i++) A.id_offset[i] = id_offsets[i];
    A.out_hits = (mmgpu_pf_hit *)d_out_hits;
    A.out_counts = (uint32_t *)d_out_counts;
    HIP_TRY(launch_pf_merge(A, c->stream));
    return MMGPU_OK;
}

// ---- multi-GPU runs equal to the unsplit run (see include/mmgpu.h) ----
extern "C" int mmgpu_pf_set_shard(mmgpu_ctx *c, const mmgpu_pf_shard *sh) {
    if (!c) return fail(MMGPU_ERR_ARG, "mmgpu_pf_set_shard: NULL context");
    if (!sh) { c->shard.on = false; return MMGPU_OK; }
    if (!c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_pf_set_shard: no targets loaded");
    if (!sh->global_ids || !sh->shard_of || !sh->local_id || sh->n_shards < 1 || sh->n_shards > 64 || sh->shard >= sh->n_shards ||
        sh->global_db_size < c->db.n)
        return fail(MMGPU_ERR_ARG, "mmgpu_pf_set_shard: bad shard description");
    for (uint32_t i = 0; i < c->db.n; i++) {
        const uint32_t g = sh->global_ids[i];
        if (g >= sh->global_db_size || (i && g <= sh->global_ids[i - 1]) || sh->shard_of[g] != sh->shard || sh->local_id[g] != i)
            return fail(MMGPU_ERR_ARG, "mmgpu_pf_set_shard: global_ids must ascend and agree with shard_of / local_id");
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->shard.d_global_ids.alloc(std::max<size_t>(c->db.n, 1) * 4));
    HIP_TRY(c->shard.d_shard_of.alloc((size_t)sh->global_db_size * 4));
    HIP_TRY(c->shard.d_local_id.alloc((size_t)sh->global_db_size * 4));
    HIP_TRY(hipMemcpy(c->shard.d_global_ids.p, sh->global_ids, (size_t)c->db.n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->shard.d_shard_of.p, sh->shard_of, (size_t)sh->global_db_size * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->shard.d_local_id.p, sh->local_id, (size_t)sh->global_db_size * 4, hipMemcpyHostToDevice));
    c->shard.n_shards = sh->n_shards;
    c->shard.shard = sh->shard;
    c->shard.global_n = sh->global_db_size;
    c->shard.on = true;
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_fetch_exchange(mmgpu_ctx *c, mmgpu_pf_batch_t *b, void *d_xhits, uint32_t stride, void *d_counts) {
    if (!c || !b || ((!d_xhits || !d_counts) && b->nq)) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch_exchange: NULL argument");
    if (!b->ran || !b->exchange) return fail(MMGPU_ERR_STATE, "mmgpu_pf_fetch_exchange: not an exchange batch that has been run");
    if (stride < b->max_hits) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch_exchange: stride smaller than min(max_hits, dbSize)");
    if (b->nq == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy2DAsync(d_xhits, (size_t)stride * sizeof(mmgpu_pf_xhit), b->d_hits.p, (size_t)b->max_hits * sizeof(mmgpu_pf_xhit),
                             (size_t)b->max_hits * sizeof(mmgpu_pf_xhit), b->nq, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_counts, b->d_hit_count.p, (size_t)b->nq * 4, hipMemcpyDeviceToDevice, c->stream));
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_merge_exchange(mmgpu_ctx *c, mmgpu_pf_batch_t *b, const void *d_xhits, const void *d_counts, uint32_t n_shards,
                                       uint32_t stride, const uint32_t *identity_global, void *d_out_hits, uint32_t out_stride,
                                       void *d_out_counts, void *d_out_flags) {
    if (!c || !b || ((!d_xhits || !d_counts || !d_out_hits || !d_out_counts) && b->nq))
        return fail(MMGPU_ERR_ARG, "mmgpu_pf_merge_exchange: NULL argument");
    if (!b->exchange) return fail(MMGPU_ERR_STATE, "mmgpu_pf_merge_exchange: `batch` must be this device's exchange batch of the same queries");
    if (n_shards < 1 || n_shards > 64) return fail(MMGPU_ERR_ARG, "mmgpu_pf_merge_exchange: n_shards must be in [1, 64]");
    if ((uint64_t)n_shards * stride > PF_XMERGE_CAP) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_merge_exchange: more than 4096 records per query");
    if (out_stride < b->max_hits) return fail(MMGPU_ERR_ARG, "mmgpu_pf_merge_exchange: out_stride smaller than min(max_hits, dbSize)");
    if (b->nq == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint32_t> ident(b->nq, 0xFFFFFFFFu);
    if (identity_global) ident.assign(identity_global, identity_global + b->nq);
    // the batch's identity buffer holds the LOCAL id for the select kernel; the merge wants the global one
    DevBuf d_ident;
    HIP_TRY(d_ident.alloc((size_t)b->nq * 4));
    HIP_TRY(hipMemcpyAsync(d_ident.p, ident.data(), (size_t)b->nq * 4, hipMemcpyHostToDevice, c->stream));
    PfXMergeArgs A;
    A.xhits = (const mmgpu_pf_xhit *)d_xhits;
    A.counts = (const uint32_t *)d_counts;
    A.n_shards = n_shards;
    A.nq = b->nq;
    A.stride = stride;
    A.max_hits = b->max_hits;
    A.min_diag_score = b->par.min_diag_score;
    A.ref_bins = b->ref_bins;
    A.q_self_score = b->d_qself.as<int32_t>();
    A.q_identity = d_ident.as<uint32_t>();
    A.out_hits = (mmgpu_pf_hit *)d_out_hits;
    A.out_stride = out_stride;
    A.out_counts = (uint32_t *)d_out_counts;
    A.out_flags = (uint32_t *)d_out_flags;
    HIP_TRY(launch_pf_xmerge(A, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // `ident` (pageable) and d_ident die with this scope
    return MMGPU_OK;
}

// ---- the exchange step with the collectives inside the library (Prefiltering::mergeTargetSplits'
// role, Prefiltering.cpp:412-526, but with the unsplit run's result) ----
namespace mmgpu {

// phase 1: receive buffers; the two all-gathers of the step read the batch's own result buffers (no staging copy)
int pf_xchg_begin(mmgpu_ctx *c, mmgpu_pf_batch_t *b, int n_ranks, XchgBlock blocks[2]) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_pf_exchange_merge: NULL argument");
    if (!b->ran || !b->exchange) return fail(MMGPU_ERR_STATE, "mmgpu_pf_exchange_merge: not an exchange batch that has been run (mmgpu_pf_set_shard before mmgpu_pf_prepare)");
    if (!c->shard.on || (int)c->shard.n_shards != n_ranks)
        return fail(MMGPU_ERR_STATE, "mmgpu_pf_exchange_merge: the communicator's ranks and the shard description's n_shards differ");
    if ((uint64_t)n_ranks * b->max_hits > PF_XMERGE_CAP) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_pf_exchange_merge: more than 4096 records per query");
    HIP_TRY(hipSetDevice(c->device));
    const size_t hb = (size_t)b->nq * b->max_hits * sizeof(mmgpu_pf_xhit), cb = (size_t)b->nq * 4;
    if (b->x_recv_hits.bytes != hb * n_ranks) HIP_TRY(b->x_recv_hits.alloc(hb * n_ranks));
    if (b->x_recv_counts.bytes != cb * n_ranks) HIP_TRY(b->x_recv_counts.alloc(cb * n_ranks));
    if (b->x_hits.bytes != (size_t)b->nq * b->max_hits * sizeof(mmgpu_pf_hit)) {
        HIP_TRY(b->x_hits.alloc((size_t)b->nq * b->max_hits * sizeof(mmgpu_pf_hit)));
        HIP_TRY(b->x_counts.alloc(cb));
        HIP_TRY(b->x_flags.alloc(cb));
        HIP_TRY(b->x_ident.alloc(cb));
    }
    // A query mmgpu_pf_run declined on the HOST (MMGPU_PF_LONG_SEQ, MMGPU_PF_OVERFLOW beyond PF_MAX_SEG) contributed no records
    // on this shard: its exchanged count says so in bit 31 (the "depends on the whole database" flag the merge kernel ORs over
    // the shards), so every rank reports the merged list as not exact instead of one that silently lacks this shard's hits.
    {
        static const uint32_t declined = 0x80000000u;      // (outlives the asynchronous copies)
        for (uint32_t q = 0; q < b->nq && q < b->status.size(); q++)
            if (b->status[q] != MMGPU_PF_OK)
                HIP_TRY(hipMemcpyAsync(b->d_hit_count.as<uint32_t>() + q, &declined, 4, hipMemcpyHostToDevice, c->stream));
    }
    blocks[0] = XchgBlock{b->d_hits.p, b->x_recv_hits.p, hb};
    blocks[1] = XchgBlock{b->d_hit_count.p, b->x_recv_counts.p, cb};
    return MMGPU_OK;
}

// phase 3: threshold, truncation and final order of the unsplit run over the union (pf_xmerge_kernel)
int pf_xchg_merge(mmgpu_ctx *c, mmgpu_pf_batch_t *b, int n_ranks, const uint32_t *identity_global) {
    if (b->nq == 0) { b->x_ranks = n_ranks; return MMGPU_OK; }
    HIP_TRY(hipSetDevice(c->device));
    // the batch's identity buffer holds the LOCAL id for the select kernel; the merge wants the global one.  The host copy
    // lives in the batch: the asynchronous upload may read it after this call returns
    b->x_ident_host.assign(b->nq, 0xFFFFFFFFu);
    if (identity_global) b->x_ident_host.assign(identity_global, identity_global + b->nq);
    HIP_TRY(hipMemcpyAsync(b->x_ident.p, b->x_ident_host.data(), (size_t)b->nq * 4, hipMemcpyHostToDevice, c->stream));
    PfXMergeArgs A;
    A.xhits = b->x_recv_hits.as<mmgpu_pf_xhit>();
    A.counts = b->x_recv_counts.as<uint32_t>();
    A.n_shards = (uint32_t)n_ranks;
    A.nq = b->nq;
    A.stride = b->max_hits;
    A.max_hits = b->max_hits;
    A.min_diag_score = b->par.min_diag_score;
    A.ref_bins = b->ref_bins;
    A.q_self_score = b->d_qself.as<int32_t>();
    A.q_identity = b->x_ident.as<uint32_t>();
    A.out_hits = b->x_hits.as<mmgpu_pf_hit>();
    A.out_stride = b->max_hits;
    A.out_counts = b->x_counts.as<uint32_t>();
    A.out_flags = b->x_flags.as<uint32_t>();
    HIP_TRY(launch_pf_xmerge(A, c->stream));
    b->x_ranks = n_ranks;
    b->x_redo_status.assign(b->nq, -1);
    return MMGPU_OK;
}

// ---- queries whose merged list is flagged inexact (a shard reached its share of the reference's databaseHits buffer, or declined
// the query): Prefiltering::mergeTargetSplits (Prefiltering.cpp:412-526) never hands a query back, so neither does a sharded run
// here - the query runs once more against a context that holds the WHOLE database, and that list replaces the merged one ----
int pf_redo_flagged(mmgpu_ctx *c, mmgpu_pf_batch_t *b, std::vector<uint32_t> &flagged) {
    flagged.clear();
    if (!c || !b || !b->x_ranks) return fail(MMGPU_ERR_STATE, "mmgpu_pf_exchange_redo_unsplit: the batch holds no merged lists (mmgpu_pf_exchange_merge first)");
    if (b->nq == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint32_t> flags(b->nq);
    HIP_TRY(hipMemcpyAsync(flags.data(), b->x_flags.p, (size_t)b->nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // (a query of 32768 residues or more is declined by a shard on the host and never reaches the merged flags: the unsplit
    // context scores it, pf_longq_kernel)
    for (uint32_t q = 0; q < b->nq; q++)
        if ((flags[q] & 1u) || (q < b->long_query.size() && b->long_query[q])) flagged.push_back(q);
    return MMGPU_OK;
}

int pf_redo_run(mmgpu_ctx *full, const mmgpu_pf_params *par, const mmgpu_pf_query *qs, const std::vector<uint32_t> &flagged, PfRedoRows &rows) {
    rows.q = flagged;
    rows.hits.clear(); rows.counts.clear(); rows.status.clear();
    if (flagged.empty()) return MMGPU_OK;
    if (!full || !par || !qs) return fail(MMGPU_ERR_ARG, "mmgpu_pf_exchange_redo_unsplit: NULL argument");
    if (full->shard.on) return fail(MMGPU_ERR_STATE, "mmgpu_pf_exchange_redo_unsplit: the second context holds a shard, not the whole database");
    std::vector<mmgpu_pf_query> sub(flagged.size());
    for (size_t k = 0; k < flagged.size(); k++) sub[k] = qs[flagged[k]];
    mmgpu_pf_batch_t *fb = nullptr;
    if (int e = mmgpu_pf_prepare(full, par, sub.data(), (uint32_t)sub.size(), &fb)) return e;
    int rc = mmgpu_pf_run(full, fb);
    rows.stride = std::max<uint32_t>(fb->max_hits, 1);
    rows.hits.assign(flagged.size() * (size_t)rows.stride, mmgpu_pf_hit());
    rows.counts.assign(flagged.size(), 0);
    rows.status.assign(flagged.size(), 0);
    if (rc == MMGPU_OK) rc = mmgpu_pf_fetch(full, fb, rows.hits.data(), rows.stride, rows.counts.data(), rows.status.data(), nullptr);
    mmgpu_pf_free(full, fb);
    return rc;
}

int pf_redo_apply(mmgpu_ctx *c, mmgpu_pf_batch_t *b, const PfRedoRows &rows) {
    if (rows.q.empty()) return MMGPU_OK;
    if (!c || !b || !b->x_ranks) return fail(MMGPU_ERR_STATE, "mmgpu_pf_exchange_redo_unsplit: the batch holds no merged lists");
    if (rows.stride != b->max_hits) return fail(MMGPU_ERR_STATE, "mmgpu_pf_exchange_redo_unsplit: the unsplit run's lists have another stride (max_hits / database size differ)");
    HIP_TRY(hipSetDevice(c->device));
    static const uint32_t zero = 0;
    for (size_t k = 0; k < rows.q.size(); k++) {
        const uint32_t q = rows.q[k];
        if (q >= b->nq) return fail(MMGPU_ERR_ARG, "mmgpu_pf_exchange_redo_unsplit: query index out of range");
        b->x_redo_status[q] = rows.status[k];
        if (rows.status[k] != MMGPU_PF_OK) continue;      // (the unsplit run hands it to the host as well: the flag stays)
        HIP_TRY(hipMemcpyAsync(b->x_hits.as<mmgpu_pf_hit>() + (size_t)q * b->max_hits, rows.hits.data() + k * (size_t)rows.stride,
                               (size_t)rows.stride * sizeof(mmgpu_pf_hit), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(b->x_counts.as<uint32_t>() + q, &rows.counts[k], 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(b->x_flags.as<uint32_t>() + q, &zero, 4, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));      // (the rows are the caller's)
    return MMGPU_OK;
}

const int32_t *pf_batch_redo_status(const mmgpu_pf_batch_t *b) { return b && b->x_redo_status.size() == b->nq ? b->x_redo_status.data() : nullptr; }

int pf_batch_merged_flags(mmgpu_pf_batch_t *b, const void **d_flags) {
    if (!b || !b->x_ranks) return fail(MMGPU_ERR_STATE, "no merged lists in this batch");
    *d_flags = b->x_flags.p;
    return MMGPU_OK;
}

// the status mmgpu_pf_run decided on the HOST for each query of a batch (MMGPU_PF_LONG_SEQ, MMGPU_PF_OVERFLOW beyond PF_MAX_SEG):
// such a query contributes no records to the exchange and its device-side flag stays clear
const int32_t *pf_batch_host_status(const mmgpu_pf_batch_t *b) { return b && b->status.size() == b->nq ? b->status.data() : nullptr; }

bool pf_batch_merged_lists(mmgpu_pf_batch_t *b, const mmgpu_pf_hit **hits, const uint32_t **counts, uint32_t *stride, uint32_t *nq) {
    if (!b || !b->x_ranks) return false;
    *hits = b->x_hits.as<mmgpu_pf_hit>();
    *counts = b->x_counts.as<uint32_t>();
    *stride = b->max_hits;
    *nq = b->nq;
    return true;
}

}  // namespace mmgpu

extern "C" int mmgpu_pf_exchange_merge(mmgpu_ctx *c, mmgpu_pf_batch_t *b, const uint32_t *identity_global, const void **d_hits,
                                       const void **d_counts, const void **d_flags, uint32_t *stride) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_pf_exchange_merge: NULL argument");
    const int n = c->comm ? c->comm->n_ranks : 1;
    mmgpu::XchgBlock blk[2];
    if (int e = mmgpu::pf_xchg_begin(c, b, n, blk)) return e;
    for (int k = 0; k < 2; k++)
        if (int e = mmgpu::comm_allgather(c, blk[k].send, blk[k].recv, blk[k].bytes)) return e;
    if (int e = mmgpu::pf_xchg_merge(c, b, n, identity_global)) return e;
    if (d_hits) *d_hits = b->x_hits.p;
    if (d_counts) *d_counts = b->x_counts.p;
    if (d_flags) *d_flags = b->x_flags.p;
    if (stride) *stride = b->max_hits;
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_exchange_redo_unsplit(mmgpu_ctx *c, mmgpu_pf_batch_t *b, mmgpu_ctx *full, const mmgpu_pf_params *par,
                                              const mmgpu_pf_query *qs, uint32_t nq, uint32_t *n_redone, uint32_t *n_left) {
    if (n_redone) *n_redone = 0;
    if (n_left) *n_left = 0;
    if (!c || !b || !full) return fail(MMGPU_ERR_ARG, "mmgpu_pf_exchange_redo_unsplit: NULL argument");
    if (nq != b->nq) return fail(MMGPU_ERR_ARG, "mmgpu_pf_exchange_redo_unsplit: query count differs from the batch");
    std::vector<uint32_t> flagged;
    if (int e = mmgpu::pf_redo_flagged(c, b, flagged)) return e;
    if (flagged.empty()) return MMGPU_OK;
    mmgpu::PfRedoRows rows;
    if (int e = mmgpu::pf_redo_run(full, par, qs, flagged, rows)) return e;
    if (int e = mmgpu::pf_redo_apply(c, b, rows)) return e;
    uint32_t left = 0;
    for (int32_t st : rows.status) left += st != MMGPU_PF_OK;
    if (n_redone) *n_redone = (uint32_t)flagged.size();
    if (n_left) *n_left = left;
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_localize_lists(mmgpu_ctx *c, const void *d_hits, const void *d_counts, uint32_t nq, uint32_t stride,
                                       void *d_local_hits, void *d_local_counts, void *d_local_slot) {
    if (!c || ((!d_hits || !d_counts || !d_local_hits || !d_local_counts || !d_local_slot) && nq))
        return fail(MMGPU_ERR_ARG, "mmgpu_pf_localize_lists: NULL argument");
    if (!c->shard.on) return fail(MMGPU_ERR_STATE, "mmgpu_pf_localize_lists: no shard set (mmgpu_pf_set_shard)");
    HIP_TRY(hipSetDevice(c->device));
    PfLocalizeArgs A;
    A.hits = (const mmgpu_pf_hit *)d_hits;
    A.counts = (const uint32_t *)d_counts;
    A.nq = nq;
    A.stride = stride;
    A.shard = c->shard.shard;
    A.shard_of = c->shard.d_shard_of.as<uint32_t>();
    A.local_id = c->shard.d_local_id.as<uint32_t>();
    A.local_hits = (mmgpu_pf_hit *)d_local_hits;
    A.local_counts = (uint32_t *)d_local_counts;
    A.local_slot = (uint32_t *)d_local_slot;
    HIP_TRY(launch_pf_localize(A, c->stream));
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_stage_ms(mmgpu_ctx *c, mmgpu_pf_batch_t *b, float ms[7]) {
    if (!c || !b || !ms) return fail(MMGPU_ERR_ARG, "mmgpu_pf_stage_ms: NULL argument");
    if (!b->ran || b->nq == 0) return fail(MMGPU_ERR_STATE, "mmgpu_pf_stage_ms: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(b->ev[4]));
    HIP_TRY(hipEventElapsedTime(&ms[0], b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms[1], b->ev[1], b->ev[2]));
    ms[2] = ms[3] = ms[4] = ms[5] = 0.f;     // summed over the stage chunks
    for (uint32_t ch = 0; ch < b->last_chunks; ch++) {
        hipEvent_t *cev = &b->chunk_ev[(size_t)4 * ch];
        hipEvent_t from = ch == 0 ? b->ev[2] : b->chunk_ev[(size_t)4 * ch - 1];
        for (int k = 0; k < 4; k++) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, k == 0 ? from : cev[k - 1], cev[k]));
            ms[2 + k] += t;
        }
    }
    HIP_TRY(hipEventElapsedTime(&ms[6], b->ev[0], b->ev[4]));
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_last_cells(mmgpu_ctx *c, mmgpu_pf_batch_t *b, uint64_t *cells, uint64_t *candidates) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_pf_last_cells: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_pf_last_cells: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (cells) {
        std::vector<uint64_t> qc(b->nq);
        if (b->nq) HIP_TRY(hipMemcpy(qc.data(), b->d_cells.p, (size_t)b->nq * 8, hipMemcpyDeviceToHost));
        uint64_t t = 0;
        for (uint64_t v : qc) t += v;
        *cells = t;
    }
    if (candidates) {
        std::vector<uint32_t> cc((size_t)b->nq * b->bins);
        if (!cc.empty()) HIP_TRY(hipMemcpy(cc.data(), b->d_cand_count.p, cc.size() * 4, hipMemcpyDeviceToHost));
        uint64_t t = 0;
        for (uint32_t v : cc) t += v;
        *candidates = t;
    }
    return MMGPU_OK;
}

extern "C" int mmgpu_pf_debug_fetch(mmgpu_ctx *c, mmgpu_pf_batch_t *b, int what, void *dst, size_t cap, size_t *bytes) {
    if (!c || !b || !bytes) return fail(MMGPU_ERR_ARG, "mmgpu_pf_debug_fetch: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_pf_debug_fetch: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->pf || c->pf->w_owner != b) return fail(MMGPU_ERR_STATE, "mmgpu_pf_debug_fetch: another batch ran after this one");
    PfIndex &P = *c->pf;
    const void *src = nullptr;
    size_t n = 0;
    uint32_t binsv[3] = {b->bins, b->ref_bins, (uint32_t)PF_T};
    switch (what) {
        case MMGPU_PF_DBG_NSIM: src = b->d_nsim.p; n = (size_t)b->n_pos * 4; break;
        case MMGPU_PF_DBG_LIST_BASE: src = b->d_list_base.p; n = ((size_t)b->n_pos + 1) * 4; break;
        case MMGPU_PF_DBG_LISTS: src = P.w_lists.p; n = (size_t)b->last_lists * sizeof(PfList); break;
        case MMGPU_PF_DBG_PEB: src = b->d_peb.p; n = ((size_t)b->n_pos + 1) * 4; break;
        case MMGPU_PF_DBG_SPLIT: {
            // the tiles in the form the tests read (one 8-byte word per entry: id | diagonal << 32 | slot << 48), put together from
            // the device's 4-byte words, the high diagonal bytes and the bin offsets (the bin is where an entry stands)
            n = (size_t)b->last_tiles * PF_T * 8;
            *bytes = n;
            if (!dst || cap < n) return MMGPU_OK;
            const size_t nt = b->last_tiles, B = b->bins;
            std::vector<uint32_t> w(nt * PF_T);
            std::vector<uint8_t> hi(nt * PF_T);
            std::vector<uint16_t> bo(nt * (B + 1));
            if (nt) {
                HIP_TRY(hipMemcpy(w.data(), P.w_split.p, w.size() * 4, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(hi.data(), P.w_split_hi.p, hi.size(), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(bo.data(), P.w_bin_off.p, bo.size() * 2, hipMemcpyDeviceToHost));
            }
            int bshift = 0;
            while ((1u << bshift) < B) bshift++;
            uint64_t *o = static_cast<uint64_t *>(dst);
            for (size_t t = 0; t < nt; t++) {
                const uint16_t *tb = bo.data() + t * (B + 1);
                size_t bin = 0;
                const size_t tn = tb[B];
                for (size_t k = 0; k < (size_t)PF_T; k++) {
                    uint64_t v = 0;
                    if (k < tn) {
                        while (bin + 1 < B && k >= tb[bin + 1]) bin++;
                        const uint32_t e = w[t * PF_T + k];
                        const uint64_t id = ((uint64_t)(e & 0xFFFu) << bshift) | bin;
                        const uint64_t diag = ((e >> 12) & 0xFFu) | ((uint64_t)hi[t * PF_T + k] << 8);
                        v = id | (diag << 32) | ((uint64_t)(e >> 20) << 48);
                    }
                    o[t * PF_T + k] = v;
                }
            }
            return MMGPU_OK;
        }
        case MMGPU_PF_DBG_BIN_OFF: src = P.w_bin_off.p; n = (size_t)b->last_tiles * (b->bins + 1) * 2; break;
        case MMGPU_PF_DBG_CAND_BASE: src = b->d_cand_base.p; n = ((size_t)b->nq * b->bins + 1) * 4; break;
        case MMGPU_PF_DBG_SURV:
            if (b->last_chunks > 1) return fail(MMGPU_ERR_STATE, "mmgpu_pf_debug_fetch: the survivors of a batch run in several stage chunks are gone (raise MMGPU_PF_STAGE_GB)");
            src = P.w_surv.p;
            n = (size_t)b->last_entries * sizeof(PfCand);
            break;
        case MMGPU_PF_DBG_SURV_COUNT: src = b->d_surv_count.p; n = (size_t)b->nq * 4; break;
        case MMGPU_PF_DBG_BINS:
            *bytes = sizeof(binsv);
            if (dst && cap >= sizeof(binsv)) memcpy(dst, binsv, sizeof(binsv));
            return MMGPU_OK;
        default: return fail(MMGPU_ERR_ARG, "mmgpu_pf_debug_fetch: unknown buffer");
    }
    *bytes = n;
    const size_t m = std::min(n, cap);
    if (dst && m) HIP_TRY(hipMemcpy(dst, src, m, hipMemcpyDeviceToHost));
    return MMGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Persisted device layout.  The reference prepares a database for its GPU path ahead of time
// (`makepaddedseqdb`, src/util/makepaddedseqdb.cpp: sequences padded and ordered for the device) and keeps precomputed prefilter
// indexes on disk (`createindex`; PrefilteringIndexReader.cpp reads them back).  Here it is ONE file with what a context has
// resident, in the layout it has on the device: targets (4-byte aligned residues, offsets, lengths), the prefilter's masked view
// when there is one, and the k-mer index (uint32 offsets, 8-byte entries).  mmgpu_db_load brings it back without the host
// touching a sequence: no SequenceLookup fill, no masking, no index build.  What is NOT in the file: the similar-k-mer score
// tables (they belong to the matrix, the caller hands them over as for mmgpu_pf_build_index) and anything derived on the device in
// milliseconds (compact offset table, non-empty bit table).
namespace {

struct DbFileHeader {
    char magic[8];                 // "MMGPUDB1"
    uint32_t version, header_bytes;
    uint64_t source_fp, index_fp;  // the caller's fingerprints: of the source database; of what the index depends on beyond it
    uint32_t n, alphabet, max_len, mean_len;
    uint64_t total_residues, res_bytes;
    uint32_t has_masked, has_index;
    int32_t k, spaced, kbase, pad0;
    uint64_t table, n_entries;
    uint64_t at_off4, at_len, at_res, at_masked, at_offsets, at_entries, file_bytes;
    uint64_t sum[6];               // version 2: checksum of every section (db_kernels.hip), in the order of the at_ fields
};
constexpr uint32_t DB_VERSION = 2;
static const char DB_MAGIC[8] = {'M', 'M', 'G', 'P', 'U', 'D', 'B', '1'};

static uint64_t align4k(uint64_t x) { return (x + 4095ull) & ~4095ull; }

// device -> file, through a pinned staging buffer
static int write_section(FILE *f, uint64_t at, const void *dev, size_t bytes) {
    if (fseeko(f, (off_t)at, SEEK_SET) != 0) return fail(MMGPU_ERR_ARG, "mmgpu_db_save: seek failed");
    const size_t chunk = 64ull << 20;
    uint8_t *stage = nullptr;
    HIP_TRY(hipHostMalloc((void **)&stage, std::min(chunk, std::max<size_t>(bytes, 1)), hipHostMallocDefault));
    int rc = MMGPU_OK;
    for (size_t o = 0; o < bytes && rc == MMGPU_OK; o += chunk) {
        const size_t m = std::min(chunk, bytes - o);
        if (hipMemcpy(stage, static_cast<const uint8_t *>(dev) + o, m, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MMGPU_ERR_HIP, "mmgpu_db_save: device read failed");
        else if (fwrite(stage, 1, m, f) != m) rc = fail(MMGPU_ERR_ARG, "mmgpu_db_save: write failed (disk full?)");
    }
    (void)hipHostFree(stage);
    return rc;
}

// file -> device: DB_READERS host threads, each with a pinned buffer of its own, take the chunks of a section in turn - pread (the
// kernel copies out of the page cache, no fault per page as through a mapping), then the copy engine moves the chunk while the
// thread reads its next one.  (One thread per chunk, not all threads on every chunk: starting 64 threads for each 32 MB chunk was
// most of the 0.24 s the 2.97 GB of a 1 M-target database took.)
constexpr int DB_READERS = 8;
constexpr size_t DB_CHUNK = 16ull << 20;
static int upload_section(int device, void *dev, int fd, uint64_t at, size_t bytes, hipStream_t up, uint8_t *const stage[DB_READERS], hipEvent_t const moved[DB_READERS]) {
    const size_t n_chunks = (bytes + DB_CHUNK - 1) / DB_CHUNK;
    std::atomic<int> bad{0};
    auto reader = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { bad = 2; return; }
        for (size_t j = (size_t)t; j < n_chunks && !bad; j += DB_READERS) {
            const size_t o = j * DB_CHUNK, m = std::min(DB_CHUNK, bytes - o);
            if (hipEventSynchronize(moved[t]) != hipSuccess) { bad = 2; return; }
            for (size_t a = 0; a < m;) {
                const ssize_t got = pread(fd, stage[t] + a, m - a, (off_t)(at + o + a));
                if (got <= 0) { bad = 1; return; }
                a += (size_t)got;
            }
            if (hipMemcpyAsync(static_cast<uint8_t *>(dev) + o, stage[t], m, hipMemcpyHostToDevice, up) != hipSuccess ||
                hipEventRecord(moved[t], up) != hipSuccess) { bad = 2; return; }
        }
    };
    const int nt = (int)std::min<size_t>(DB_READERS, n_chunks);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(reader, t);
    if (nt > 0) reader(0);
    for (auto &x : th) x.join();
    if (bad == 1) return fail(MMGPU_ERR_STATE, "mmgpu_db_load: short read (file truncated?)");
    if (bad) return fail(MMGPU_ERR_HIP, "mmgpu_db_load: upload failed");
    return MMGPU_OK;
}

}  // namespace

extern "C" int mmgpu_db_save(mmgpu_ctx *c, const char *path, uint64_t source_fingerprint, uint64_t index_fingerprint) {
    if (!c || !path) return fail(MMGPU_ERR_ARG, "mmgpu_db_save: NULL argument");
    if (!c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_db_save: no targets loaded");
    if (c->shard.on) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_db_save: the context holds a shard of a multi-GPU run (save the unsplit database)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const PfIndex *P = c->pf;
    const bool with_index = P != nullptr && index_fingerprint != 0 && P->d_offsets.p && P->kbase == P->kalph;
    DbFileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, DB_MAGIC, 8);
    h.version = DB_VERSION;
    h.header_bytes = (uint32_t)sizeof(h);
    h.source_fp = source_fingerprint;
    h.index_fp = with_index ? index_fingerprint : 0;
    h.n = c->db.n;
    h.alphabet = (uint32_t)c->db.alphabet;
    h.max_len = c->db.max_len;
    h.mean_len = c->mean_len;
    h.total_residues = c->db.total_residues;
    h.res_bytes = c->db.res_bytes;
    h.has_masked = c->pf_masked_res ? 1u : 0u;
    h.has_index = with_index ? 1u : 0u;
    const size_t nn = std::max<uint32_t>(c->db.n, 1);
    uint64_t at = align4k(sizeof(h));
    h.at_off4 = at; at = align4k(at + nn * 4);
    h.at_len = at; at = align4k(at + nn * 4);
    h.at_res = at; at = align4k(at + h.res_bytes);
    if (h.has_masked) { h.at_masked = at; at = align4k(at + h.res_bytes); }
    if (with_index) {
        h.k = P->k; h.spaced = P->spaced; h.kbase = P->kbase;
        h.table = P->table; h.n_entries = P->n_entries;
        h.at_offsets = at; at = align4k(at + (P->table + 1) * 4);
        h.at_entries = at; at = align4k(at + std::max<uint64_t>(P->n_entries, 1) * 8);
    }
    h.file_bytes = at;
    {   // the sections' checksums, taken where they lie
        DevBuf d_sum;
        HIP_TRY(d_sum.alloc(8));
        unsigned long long *sc = d_sum.as<unsigned long long>();
        HIP_TRY(db_section_checksum(c->db.off4, nn * 4, sc, &h.sum[0], c->stream));
        HIP_TRY(db_section_checksum(c->db.len, nn * 4, sc, &h.sum[1], c->stream));
        HIP_TRY(db_section_checksum(c->db.res, h.res_bytes, sc, &h.sum[2], c->stream));
        if (h.has_masked) HIP_TRY(db_section_checksum(c->pf_masked_res, h.res_bytes, sc, &h.sum[3], c->stream));
        if (with_index) {
            HIP_TRY(db_section_checksum(P->d_offsets.p, (P->table + 1) * 4, sc, &h.sum[4], c->stream));
            HIP_TRY(db_section_checksum(P->d_entries.p, P->n_entries * 8, sc, &h.sum[5], c->stream));
        }
    }
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(MMGPU_ERR_ARG, std::string("mmgpu_db_save: cannot create ") + tmp);
    int rc = MMGPU_OK;
    if (fwrite(&h, sizeof(h), 1, f) != 1) rc = fail(MMGPU_ERR_ARG, "mmgpu_db_save: write failed");
    if (rc == MMGPU_OK) rc = write_section(f, h.at_off4, c->db.off4, nn * 4);
    if (rc == MMGPU_OK) rc = write_section(f, h.at_len, c->db.len, nn * 4);
    if (rc == MMGPU_OK) rc = write_section(f, h.at_res, c->db.res, h.res_bytes);
    if (rc == MMGPU_OK && h.has_masked) rc = write_section(f, h.at_masked, c->pf_masked_res, h.res_bytes);
    if (rc == MMGPU_OK && with_index) rc = write_section(f, h.at_offsets, P->d_offsets.p, (P->table + 1) * 4);
    if (rc == MMGPU_OK && with_index && P->n_entries) rc = write_section(f, h.at_entries, P->d_entries.p, P->n_entries * 8);
    if (rc == MMGPU_OK && (ftruncate(fileno(f), (off_t)h.file_bytes) != 0)) rc = fail(MMGPU_ERR_ARG, "mmgpu_db_save: cannot size the file");
    if (fclose(f) != 0 && rc == MMGPU_OK) rc = fail(MMGPU_ERR_ARG, "mmgpu_db_save: close failed");
    if (rc == MMGPU_OK && rename(tmp.c_str(), path) != 0) rc = fail(MMGPU_ERR_ARG, std::string("mmgpu_db_save: cannot rename to ") + path);
    if (rc != MMGPU_OK) (void)remove(tmp.c_str());
    return rc;
}

static int db_read_header(const char *path, DbFileHeader *h, int *fd_out) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(MMGPU_ERR_STATE, std::string("mmgpu_db: cannot open ") + path);
    struct stat st;
    bool ok = read(fd, h, sizeof(*h)) == (ssize_t)sizeof(*h) && memcmp(h->magic, DB_MAGIC, 8) == 0 && h->version == DB_VERSION &&
              h->header_bytes == sizeof(*h) && fstat(fd, &st) == 0 && (uint64_t)st.st_size >= h->file_bytes;
    // every section inside the file (sizes that cannot overflow the sums first)
    auto inside = [&](uint64_t at, uint64_t bytes) { return at >= sizeof(*h) && bytes <= h->file_bytes && at <= h->file_bytes - bytes; };
    if (ok) {
        const uint64_t nn = std::max<uint32_t>(h->n, 1);
        ok = h->table < (1ull << 40) && h->n_entries < (1ull << 40) && h->alphabet >= 1 && h->alphabet <= 255 &&
             inside(h->at_off4, nn * 4) && inside(h->at_len, nn * 4) && inside(h->at_res, h->res_bytes) &&
             (!h->has_masked || inside(h->at_masked, h->res_bytes)) &&
             (!h->has_index || (inside(h->at_offsets, (h->table + 1) * 4) && inside(h->at_entries, h->n_entries * 8)));
    }
    if (!ok) {
        close(fd);
        return fail(MMGPU_ERR_STATE, std::string("mmgpu_db: not a database file of this library version: ") + path);
    }
    if (fd_out) *fd_out = fd; else close(fd);
    return MMGPU_OK;
}

extern "C" int mmgpu_db_probe(const char *path, mmgpu_db_info *info) {
    if (!path || !info) return fail(MMGPU_ERR_ARG, "mmgpu_db_probe: NULL argument");
    DbFileHeader h;
    const int rc = db_read_header(path, &h, nullptr);
    if (rc != MMGPU_OK) return rc;
    memset(info, 0, sizeof(*info));
    info->source_fingerprint = h.source_fp;
    info->index_fingerprint = h.index_fp;
    info->n_targets = h.n;
    info->alphabet = h.alphabet;
    info->total_residues = h.total_residues;
    info->has_masked_view = (int32_t)h.has_masked;
    info->has_index = (int32_t)h.has_index;
    info->kmer_size = h.k;
    info->spaced = h.spaced;
    info->n_entries = h.n_entries;
    info->file_bytes = h.file_bytes;
    return MMGPU_OK;
}

extern "C" int mmgpu_db_load(mmgpu_ctx *c, const char *path, uint64_t source_fingerprint, uint64_t index_fingerprint, const mmgpu_pf_index *tables) {
    if (!c || !path) return fail(MMGPU_ERR_ARG, "mmgpu_db_load: NULL argument");
    DbFileHeader h;
    int fd = -1;
    int rc = db_read_header(path, &h, &fd);
    if (rc != MMGPU_OK) return rc;
    if (h.source_fp != source_fingerprint) { close(fd); return fail(MMGPU_ERR_STATE, "mmgpu_db_load: the file was made from another database (source fingerprint differs)"); }
    const bool want_index = index_fingerprint != 0;
    if (want_index && (!h.has_index || h.index_fp != index_fingerprint)) { close(fd); return fail(MMGPU_ERR_STATE, "mmgpu_db_load: the file holds no index built with these parameters (index fingerprint differs)"); }
    if (want_index && !tables) { close(fd); return fail(MMGPU_ERR_ARG, "mmgpu_db_load: the index needs the caller's score tables"); }
    if (want_index && (tables->kmer_size != h.k || tables->spaced != h.spaced || tables->alphabet != (int)h.alphabet)) { close(fd); return fail(MMGPU_ERR_STATE, "mmgpu_db_load: k-mer size / pattern / alphabet differ from the file's index"); }
    struct CloseFd { int fd; ~CloseFd() { close(fd); } } closer{fd};
    HIP_TRY(hipSetDevice(c->device));
    // (what the context holds stays until the file's content has been uploaded and checked: a file that turns out damaged leaves
    // the context as it was)
    const size_t nn = std::max<uint32_t>(h.n, 1);
    DeviceDb db;
    uint8_t *masked = nullptr;
    uint8_t *stage[DB_READERS] = {};
    hipEvent_t moved[DB_READERS] = {};
    hipStream_t up = nullptr;
    PfIndex *P = nullptr;
    auto drop = [&]() {
        for (int k = 0; k < DB_READERS; k++) {
            if (stage[k]) (void)hipHostFree(stage[k]);
            if (moved[k]) (void)hipEventDestroy(moved[k]);
        }
        if (up) (void)hipStreamDestroy(up);
    };
    auto undo = [&]() {
        drop();
        dev_free(db.res); dev_free(db.off4); dev_free(db.len);
        if (masked) dev_free(masked);
        delete P;
    };
#define L_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { undo(); return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
#define L_RC(expr) do { const int r__ = (expr); if (r__ != MMGPU_OK) { undo(); return r__; } } while (0)
    L_TRY(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    for (int k = 0; k < DB_READERS; k++) {
        L_TRY(hipHostMalloc((void **)&stage[k], DB_CHUNK, hipHostMallocDefault));
        L_TRY(hipEventCreateWithFlags(&moved[k], hipEventDisableTiming));
    }
    L_TRY(dev_malloc_ctx(c, (void **)&db.res, (size_t)h.res_bytes));
    L_TRY(dev_malloc_ctx(c, (void **)&db.off4, nn * 4));
    L_TRY(dev_malloc_ctx(c, (void **)&db.len, nn * 4));
    L_RC(upload_section(c->device, db.off4, fd, h.at_off4, nn * 4, up, stage, moved));
    L_RC(upload_section(c->device, db.len, fd, h.at_len, nn * 4, up, stage, moved));
    L_RC(upload_section(c->device, db.res, fd, h.at_res, (size_t)h.res_bytes, up, stage, moved));
    if (h.has_masked && want_index) {      // (the masked view serves the prefilter only: a caller that asks for the targets alone gets them alone)
        L_TRY(dev_malloc_ctx(c, (void **)&masked, (size_t)h.res_bytes));
        L_RC(upload_section(c->device, masked, fd, h.at_masked, (size_t)h.res_bytes, up, stage, moved));
    }
    db.n = h.n;
    db.res_bytes = (size_t)h.res_bytes;
    db.max_len = h.max_len;
    db.total_residues = h.total_residues;
    db.alphabet = (int)h.alphabet;
    std::vector<uint32_t> hlen(h.n);
    if (h.n && pread(fd, hlen.data(), (size_t)h.n * 4, (off_t)h.at_len) != (ssize_t)((size_t)h.n * 4)) { undo(); return fail(MMGPU_ERR_STATE, "mmgpu_db_load: short read (file truncated?)"); }
    L_TRY(hipStreamSynchronize(up));
    // what arrived is what was saved, and it has the properties the kernels rely on
    DevBuf d_chk;
    L_TRY(d_chk.alloc(8));
    auto section_ok = [&](const void *dev, size_t bytes, int which) {
        uint64_t sum = 0;
        return db_section_checksum(dev, bytes, d_chk.as<unsigned long long>(), &sum, up) == hipSuccess && sum == h.sum[which];
    };
    if (!section_ok(db.off4, nn * 4, 0) || !section_ok(db.len, nn * 4, 1) || !section_ok(db.res, (size_t)h.res_bytes, 2) ||
        (masked && !section_ok(masked, (size_t)h.res_bytes, 3))) {
        undo();
        return fail(MMGPU_ERR_STATE, "mmgpu_db_load: a section's checksum differs from the one in the header (damaged file)");
    }
    {
        uint32_t bad = 0;
        L_TRY(db_validate_layout(db.off4, db.len, h.n, h.res_bytes, h.max_len, nullptr, 0, nullptr, 0, d_chk.as<uint32_t>(), &bad, up));
        if (bad) { undo(); return fail(MMGPU_ERR_STATE, "mmgpu_db_load: a target of the file lies outside its residue block"); }
    }
    // the context takes the new database (pf_setup checks the alphabet against it); the old one is kept aside until the index is in
    DeviceDb old_db = c->db;
    uint8_t *old_masked = c->pf_masked_res;
    PfIndex *old_pf = c->pf;
    std::vector<uint32_t> old_hlen;
    old_hlen.swap(c->h_len);
    const uint32_t old_mean = c->mean_len;
    const bool old_shard = c->shard.on;
    c->pf = nullptr;
    c->shard.on = false;
    c->db = db;
    db = DeviceDb();
    c->pf_masked_res = masked;
    masked = nullptr;
    c->h_len.swap(hlen);
    c->mean_len = h.mean_len;
    auto back_to_old = [&]() {      // (the index failed: the new database goes, the old one is the context's again)
        db_release(c);
        c->db = old_db;
        c->pf_masked_res = old_masked;
        c->pf = old_pf;
        c->h_len.swap(old_hlen);
        c->mean_len = old_mean;
        c->shard.on = old_shard;
    };
    if (want_index) {
        int rc2 = pf_setup(c, tables, false, &P);
        if (rc2 == MMGPU_OK && (P->table != h.table || P->kbase != h.kbase)) rc2 = fail(MMGPU_ERR_STATE, "mmgpu_db_load: k-mer table size differs from the file's index");
        if (rc2 != MMGPU_OK) { drop(); delete P; back_to_old(); return rc2; }
        P->n_entries = h.n_entries;
        hipError_t e = P->d_offsets.alloc((P->table + 1) * 4);
        if (e == hipSuccess) e = P->d_entries.alloc(std::max<uint64_t>(P->n_entries, 1) * 8);
        int rc3 = e == hipSuccess ? MMGPU_OK : fail(MMGPU_ERR_HIP, "mmgpu_db_load: out of device memory for the index");
        if (rc3 == MMGPU_OK) rc3 = upload_section(c->device, P->d_offsets.p, fd, h.at_offsets, (P->table + 1) * 4, up, stage, moved);
        if (rc3 == MMGPU_OK && P->n_entries) rc3 = upload_section(c->device, P->d_entries.p, fd, h.at_entries, P->n_entries * 8, up, stage, moved);
        if (rc3 == MMGPU_OK && hipStreamSynchronize(up) != hipSuccess) rc3 = fail(MMGPU_ERR_HIP, "mmgpu_db_load: upload failed");
        if (rc3 == MMGPU_OK && (!section_ok(P->d_offsets.p, (P->table + 1) * 4, 4) || !section_ok(P->d_entries.p, P->n_entries * 8, 5)))
            rc3 = fail(MMGPU_ERR_STATE, "mmgpu_db_load: the index's checksum differs from the one in the header (damaged file)");
        if (rc3 == MMGPU_OK) {
            uint32_t bad = 0;
            if (db_validate_layout(c->db.off4, c->db.len, h.n, h.res_bytes, h.max_len, P->d_offsets.as<uint32_t>(), P->table, P->d_entries.as<uint64_t>(),
                                   P->n_entries, d_chk.as<uint32_t>(), &bad, up) != hipSuccess) rc3 = fail(MMGPU_ERR_HIP, "mmgpu_db_load: layout check failed");
            else if (bad) rc3 = fail(MMGPU_ERR_STATE, "mmgpu_db_load: the file's index is not a k-mer index over these targets (offsets not monotone / entry out of range)");
        }
        if (rc3 == MMGPU_OK && pf_index_bitmap(c, P) != hipSuccess) rc3 = fail(MMGPU_ERR_HIP, "mmgpu_db_load: bit table failed");
        if (rc3 == MMGPU_OK && pf_index_cofs(c, P) != hipSuccess) rc3 = fail(MMGPU_ERR_HIP, "mmgpu_db_load: compact offset table failed");
        if (rc3 != MMGPU_OK) { drop(); delete P; back_to_old(); return rc3; }
        c->pf = P;
        P = nullptr;
    }
    {   // the old database goes
        (void)hipStreamSynchronize(c->stream);
        dev_free(old_db.res); dev_free(old_db.off4); dev_free(old_db.len);
        if (old_masked) dev_free(old_masked);
        delete old_pf;
    }
    drop();
#undef L_TRY
#undef L_RC
    return MMGPU_OK;
}

extern "C" void mmgpu_pf_free(mmgpu_ctx *c, mmgpu_pf_batch_t *b) {
    if (!b) return;
    if (c) (void)hipSetDevice(c->device);
    for (auto &e : b->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto e : b->chunk_ev)
        if (e) (void)hipEventDestroy(e);
    delete b;
}

extern "C" int mmgpu_pf_batch(mmgpu_ctx *c, const mmgpu_pf_params *par, const mmgpu_pf_query *qs, uint32_t nq,
                              mmgpu_pf_hit *hits, uint32_t hit_stride, uint32_t *counts, int32_t *status) {
    mmgpu_pf_batch_t *b = nullptr;
    int rc = mmgpu_pf_prepare(c, par, qs, nq, &b);
    if (rc != MMGPU_OK) return rc;
    rc = mmgpu_pf_run(c, b);
    if (rc == MMGPU_OK) rc = mmgpu_pf_fetch(c, b, hits, hit_stride, counts, status, nullptr);
    mmgpu_pf_free(c, b);
    return rc;
}
