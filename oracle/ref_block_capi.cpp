// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
//
// The C API of the reference's Rust block-aligner crate (lib/block-aligner/c/block_aligner.h; crate 0.4.0, which this image
// cannot build: no rustc) for the calls the sequence-sequence path of the reference makes
// (SmithWaterman::alignStartPosBacktraceBlock<SEQ_SEQ>, src/alignment/StripedSmithWaterman.cpp:943-1127, and the set-up in
// the constructor / ssw_init, :706-710, :1464-1474), on top of the plain-C restatement oracle/block_oracle.c.  Linked into
// oracle/_ref/libmmref_block.so and into the two `mmseqs` binaries of integration/build_mmseqs.sh INSTEAD of the do-nothing
// stubs (oracle/gen_block_stub.py keeps generating stubs for everything not defined here).  Round 5: the AAProfile object and
// block_align_profile_aa_trace_xdrop are served too, so profile queries run the reference's PROFILE_SEQ branch (:963-990).  With it the reference's
// own code runs the block-aligner branch for int16-range hits, and the drop-in tests compare against a block-aligning
// reference.  Parity of the restatement itself against the Rust crate: see the header of block_oracle.c.
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "block_aligner.h"
extern "C" {
#include "mm_oracle.h"
}

struct AAMatrix { int8_t scores[27 * 32]; };                       // scores.rs:47-49
struct PaddedBytes { std::vector<uint8_t> b; };                    // the bytes between the paddings (scan_block.rs:2149-2152)
struct PosBias { std::vector<int16_t> b; };                        // scores.rs:703-706
struct Cigar { std::vector<OpLen> runs; };                         // runs in block_get_cigar order (origin -> end), cigar.rs:88-90
struct AAProfile {                                                 // scores.rs:475-489
    std::vector<int8_t> pos_aa;                                    // [max_len][32]
    std::vector<int16_t> aa_pos;                                   // [32][max_len]
    std::vector<int16_t> goc, gcc, gor;                            // pos_gap_open_C / pos_gap_close_C / pos_gap_open_R
    int8_t gap_extend;
    size_t max_len, curr_len, str_len;
};
struct MMBlock {                                                   // Block<true, true>: result + traceback of the last alignment
    AlignResult res;
    std::vector<uint8_t> ops;                                      // walk order (end -> origin) for the end position in `res`
    uint32_t n_ops;
};

extern "C" {

struct AAMatrix *block_new_simple_aamatrix(int8_t match_score, int8_t mismatch_score) {      // ffi.rs:32-35, scores.rs:53-66
    AAMatrix *m = new AAMatrix();
    memset(m->scores, -128, sizeof(m->scores));
    for (int a = 0; a < 26; a++)
        for (int b = 0; b < 26; b++) m->scores[a * 32 + b] = a == b ? match_score : mismatch_score;
    return m;
}
void block_set_aamatrix_num(struct AAMatrix *m, int8_t a, int8_t b, int8_t score) {           // scores.rs:105-110
    m->scores[(size_t)(uint8_t)a * 32 + (uint8_t)b] = score;
    m->scores[(size_t)(uint8_t)b * 32 + (uint8_t)a] = score;
}
void block_free_aamatrix(struct AAMatrix *m) { delete m; }

struct PaddedBytes *block_new_padded_aa(uintptr_t, uintptr_t) { return new PaddedBytes(); }
void block_set_bytes_padded_aa_numsequence(struct PaddedBytes *p, const uint8_t *s, uintptr_t len, uintptr_t) { p->b.assign(s, s + len); }
void block_free_padded_aa(struct PaddedBytes *p) { delete p; }

struct PosBias *block_new_pos_bias(uintptr_t len, uintptr_t) { PosBias *p = new PosBias(); p->b.assign(len, 0); return p; }   // PosBias::new: zeros
void block_set_pos_bias(struct PosBias *p, const int16_t *b, uintptr_t len) { p->b.assign(b, b + len); }
void block_free_pos_bias(struct PosBias *p) { delete p; }

BlockHandle block_new_aa_trace_xdrop(uintptr_t, uintptr_t, uintptr_t) {
    MMBlock *b = new MMBlock();
    b->res.score = 0;
    b->res.query_idx = b->res.reference_idx = 0;
    b->n_ops = 0;
    return b;
}
void block_free_aa_trace_xdrop(BlockHandle h) { delete static_cast<MMBlock *>(h); }

void block_align_aa_trace_xdrop_posbias(BlockHandle h, const struct PaddedBytes *q, const struct PosBias *qb, const struct PaddedBytes *r,
                                        const struct PosBias *rb, const struct AAMatrix *m, struct Gaps g, struct SizeRange s, int32_t x) {
    MMBlock *b = static_cast<MMBlock *>(h);
    const int ql = (int)q->b.size(), rl = (int)r->b.size();
    // PosBias::len must equal the sequence length (align_aa's asserts, scan_block.rs:1018-1019); positions beyond what was set are 0
    std::vector<int16_t> qbias(ql, 0), rbias(rl, 0);
    for (int k = 0; k < ql && k < (int)qb->b.size(); k++) qbias[k] = qb->b[k];
    for (int k = 0; k < rl && k < (int)rb->b.size(); k++) rbias[k] = rb->b[k];
    b->ops.resize((size_t)ql + rl + 8);
    mmo_block_res res;
    res.score = -1000000000;
    res.query_idx = res.reference_idx = 0;
    b->n_ops = 0;
    if (mmo_block_align_table(q->b.data(), qbias.data(), ql, r->b.data(), rbias.data(), rl, m->scores, g.open, g.extend, (int)s.min, (int)s.max, x,
                              &res, b->ops.data(), (uint32_t)b->ops.size(), &b->n_ops) != 0)
        abort();        // the crate panics on bad arguments
    b->res.score = res.score;
    b->res.query_idx = res.query_idx;
    b->res.reference_idx = res.reference_idx;
}
// profile queries (PROFILE_SEQ, :963-990, :1039-1051): the AAProfile object as the reference fills it - through the raw pointers to
// its two score arrays and the set_all gap setters - and Block<true, true>::align_profile on the restatement (round 5)
struct AAProfile *block_new_aaprofile(uintptr_t str_len, uintptr_t block_size, int8_t gap_extend) {       // AAProfile::new, scores.rs:494-507
    AAProfile *p = new AAProfile();
    p->max_len = p->curr_len = str_len + block_size + 1;
    p->str_len = str_len;
    p->gap_extend = gap_extend;
    p->pos_aa.assign(p->max_len * 32, (int8_t)-128);
    p->aa_pos.assign(32 * p->max_len, (int16_t)-128);
    p->goc.assign(p->max_len, (int16_t)-128);
    p->gcc.assign(p->max_len, (int16_t)-128);
    p->gor.assign(p->max_len, (int16_t)-128);
    return p;
}
int8_t *aaprofile_pos_aa(struct AAProfile *p) { return p->pos_aa.data(); }
int16_t *aaprofile_aa_pos(struct AAProfile *p) { return p->aa_pos.data(); }
size_t block_get_curr_len_aaprofile(const struct AAProfile *p) { return p->curr_len; }
void block_set_all_gap_open_C_aaprofile(struct AAProfile *p, int8_t gap) { std::fill(p->goc.begin(), p->goc.begin() + p->curr_len, (int16_t)gap); }     // :575-578
void block_set_all_gap_close_C_aaprofile(struct AAProfile *p, int8_t gap) { std::fill(p->gcc.begin(), p->gcc.begin() + p->curr_len, (int16_t)gap); }    // :580-582
void block_set_all_gap_open_R_aaprofile(struct AAProfile *p, int8_t gap) { std::fill(p->gor.begin(), p->gor.begin() + p->curr_len, (int16_t)gap); }     // :584-587
void block_free_aaprofile(struct AAProfile *p) { delete p; }
void block_align_profile_aa_trace_xdrop(BlockHandle h, const struct PaddedBytes *q, const struct AAProfile *r, struct SizeRange s, int32_t x) {
    MMBlock *b = static_cast<MMBlock *>(h);
    const int ql = (int)q->b.size(), pl = (int)r->str_len;
    // the reference writes aa_pos as the transpose of the rows it copied into pos_aa (:982-987): the restatement derives it the same way;
    // a caller that wrote the two arrays inconsistently would be a different program
    for (size_t i = 0; i <= r->str_len; i++)
        for (int a = 0; a < 32; a++)
            if (r->aa_pos[(size_t)a * r->curr_len + i] != (int16_t)r->pos_aa[i * 32 + a]) abort();
    b->ops.resize((size_t)ql + pl + 8);
    mmo_block_res res;
    res.score = -1000000000;
    res.query_idx = res.reference_idx = 0;
    b->n_ops = 0;
    if (mmo_block_align_profile(q->b.data(), ql, pl, (int)(r->max_len - r->str_len - 1), r->pos_aa.data() + 32, r->goc.data(), r->gcc.data(), r->gor.data(),
                                (int)r->curr_len, r->gap_extend, (int)s.min, (int)s.max, x, 1, 1, &res, b->ops.data(), (uint32_t)b->ops.size(), &b->n_ops) != 0)
        abort();
    b->res.score = res.score;
    b->res.query_idx = res.query_idx;
    b->res.reference_idx = res.reference_idx;
}
struct AlignResult block_res_aa_trace_xdrop(BlockHandle h) { return static_cast<MMBlock *>(h)->res; }

struct Cigar *block_new_cigar(uintptr_t, uintptr_t) { return new Cigar(); }
void block_cigar_aa_trace_xdrop(BlockHandle h, uintptr_t query_idx, uintptr_t reference_idx, struct Cigar *c) {
    MMBlock *b = static_cast<MMBlock *>(h);
    if (query_idx != b->res.query_idx || reference_idx != b->res.reference_idx) abort();   // the reference only asks for the reported end (:1063)
    c->runs.clear();
    for (uint32_t k = b->n_ops; k-- > 0;) {          // origin -> end, run-length encoded like Cigar::add (cigar.rs:72-81)
        const Operation op = (Operation)b->ops[k];
        if (!c->runs.empty() && c->runs.back().op == op) c->runs.back().len++;
        else { OpLen o; o.op = op; o.len = 1; c->runs.push_back(o); }
    }
}
uintptr_t block_len_cigar(const struct Cigar *c) { return c->runs.size(); }
struct OpLen block_get_cigar(const struct Cigar *c, uintptr_t i) { return c->runs[i]; }
void block_free_cigar(struct Cigar *c) { delete c; }

}  // extern "C"
