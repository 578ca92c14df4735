// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
//
// A CPU stand-in for the subset of the libmmgpu C-ABI (include/mmgpu.h) that the patched mmseqs binary calls
// (integration/MMGpu*.cpp), built on the plain-C oracle (oracle/*.c).  It exists for ONE purpose: to run the HOST side of
// the drop-in - the patched Alignment::run / Prefiltering::runSplit, their block logic, accept / reject replay, id <-> key
// mapping, serialisation - in a container without a GPU and diff the result DBs against the stock CPU binary
// (tests/test_mmseqs_dropin.py, `-m "not gpu"`).  The library is built as oracle/_build/emu/libmmgpu.so and is only ever
// put in front of the real one by that test through LD_LIBRARY_PATH; nothing in mmseqs2_amd/, include/ or integration/
// refers to it, and the GPU tests of the same script run against the real mmseqs2_amd/lib/libmmgpu.so.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/mmgpu.h"
#include "mm_oracle.h"

struct mmgpu_ctx {
    std::vector<uint8_t> pf_tres;      // the prefilter's (tantan-masked) view, empty = tres
    const uint8_t *pf_res() const { return pf_tres.empty() ? tres.data() : pf_tres.data(); }
    std::vector<uint8_t> tres;
    std::vector<uint64_t> toff;
    uint32_t n;
    int alphabet;
    // prefilter index
    bool have_index;
    int k, spaced, kalph, kbase;
    std::vector<int16_t> s3, s2;
    std::vector<uint32_t> i3, i2;
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> ids;
    std::vector<uint16_t> pos;
    std::vector<int8_t> ungapped;
    mmo_pf_gen gen;
};

struct mmgpu_sw_batch_t {
    std::vector<int8_t> mat;
    int alphabet, go, ge, mode;
    std::vector<std::vector<uint8_t> > q;
    std::vector<std::vector<int8_t> > cb;
    std::vector<std::vector<int8_t> > prof;      // profile queries: [letters][qlen]
    std::vector<int> prof_letters;
    std::vector<std::vector<uint32_t> > ids;
    std::vector<int32_t> min_start;
    std::vector<mmgpu_sw_hit> res;
    std::vector<std::pair<uint32_t, uint32_t> > pair;   // (query, target id) of every result slot
};

struct mmgpu_pf_batch_t {
    mmgpu_pf_params par;
    std::vector<std::vector<uint8_t> > q;
    std::vector<std::vector<float> > bias;
    std::vector<uint32_t> identity;
    std::vector<std::vector<int16_t> > pscore;     // profile queries (empty rows otherwise)
    std::vector<std::vector<uint32_t> > pindex;
    std::vector<std::vector<int8_t> > paln;
    std::vector<int> prow;
    std::vector<std::vector<mmo_pf_hit> > hits;
    std::vector<mmo_pf_stats> stats;
};

static thread_local std::string g_err;      // (the hooks drive several contexts from several threads)
static int fail(int code, const char *msg) {
    g_err = msg;
    return code;
}

extern "C" {

int mmgpu_init(mmgpu_ctx **ctx, int) {
    *ctx = new mmgpu_ctx();
    (*ctx)->n = 0;
    (*ctx)->have_index = false;
    return 0;
}
void mmgpu_destroy(mmgpu_ctx *ctx) { delete ctx; }
const char *mmgpu_last_error(void) { return g_err.c_str(); }
int mmgpu_device_info(mmgpu_ctx *, int *cus, char *name, int cap) {
    if (cus) *cus = 0;
    if (name && cap > 0) snprintf(name, cap, "CPU emulation of the C-ABI (test only)");
    return 0;
}
int mmgpu_synchronize(mmgpu_ctx *) { return 0; }

int mmgpu_load_targets(mmgpu_ctx *c, const uint8_t *res, const uint64_t *off, uint32_t n, int alphabet) {
    c->tres.assign(res, res + off[n]);
    c->pf_tres.clear();
    c->toff.assign(off, off + n + 1);
    c->n = n;
    c->alphabet = alphabet;
    c->have_index = false;
    return 0;
}

int mmgpu_sw_prepare(mmgpu_ctx *c, const mmgpu_sw_params *p, const mmgpu_sw_query *qs, uint32_t nq, int mode, mmgpu_sw_batch_t **out) {
    if (c->n == 0) return fail(MMGPU_ERR_STATE, "no targets");
    mmgpu_sw_batch_t *b = new mmgpu_sw_batch_t();
    b->mat.assign(p->mat, p->mat + p->alphabet * p->alphabet);
    b->alphabet = p->alphabet;
    b->go = p->gap_open;
    b->ge = p->gap_extend;
    b->mode = mode;
    // the device library's precondition (mmgpu_api.hip, sw_prepare_impl): outside it the textbook recurrence the kernels
    // compute is not the reference's striped loop; the stand-in refuses the same batches so that the host side is tested for it
    int minp = 0;
    for (int i = 0; i < p->alphabet * p->alphabet; i++) minp = std::min<int>(minp, p->mat[i]);
    for (uint32_t i = 0; i < nq; i++) {
        int qminp = minp, mincb = 0;
        if (qs[i].profile) {
            qminp = 0;
            for (size_t k = 0; k < (size_t)qs[i].profile_letters * qs[i].qlen; k++) qminp = std::min<int>(qminp, qs[i].profile[k]);
        } else if (qs[i].comp_bias) {
            for (uint32_t k = 0; k < qs[i].qlen; k++) mincb = std::min<int>(mincb, qs[i].comp_bias[k]);
        }
        if (qs[i].qlen && !(qminp + mincb + p->gap_extend > -p->gap_open)) {
            delete b;
            return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_prepare: gap penalties too small for this matrix (adjacent insertion+deletion could win)");
        }
    }
    for (uint32_t i = 0; i < nq; i++) {
        b->q.push_back(std::vector<uint8_t>(qs[i].q, qs[i].q + qs[i].qlen));
        if (qs[i].comp_bias) b->cb.push_back(std::vector<int8_t>(qs[i].comp_bias, qs[i].comp_bias + qs[i].qlen));
        else b->cb.push_back(std::vector<int8_t>(qs[i].qlen, 0));
        if (qs[i].profile) b->prof.push_back(std::vector<int8_t>(qs[i].profile, qs[i].profile + (size_t)qs[i].profile_letters * qs[i].qlen));
        else b->prof.push_back(std::vector<int8_t>());
        b->prof_letters.push_back(qs[i].profile ? (int)qs[i].profile_letters : 0);
        b->ids.push_back(std::vector<uint32_t>(qs[i].target_ids, qs[i].target_ids + qs[i].n_targets));
        b->min_start.push_back(qs[i].min_start_score);
        for (uint32_t k = 0; k < qs[i].n_targets; k++) {
            if (qs[i].target_ids[k] >= c->n) {
                delete b;
                return fail(MMGPU_ERR_ARG, "target id out of range");
            }
            b->pair.push_back(std::make_pair(i, qs[i].target_ids[k]));
        }
    }
    b->res.resize(b->pair.size());
    *out = b;
    return 0;
}

int mmgpu_sw_run(mmgpu_ctx *c, mmgpu_sw_batch_t *b) {
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t p = 0; p < b->pair.size(); p++) {
        const uint32_t qi = b->pair[p].first, id = b->pair[p].second;
        const std::vector<uint8_t> &q = b->q[qi];
        const uint8_t *t = c->tres.data() + c->toff[id];
        const int tlen = (int)(c->toff[id + 1] - c->toff[id]);
        mmo_sw_res r;
        const bool isProf = b->prof_letters[qi] > 0;
        char dummy[8];
        if (isProf) mmo_sw_align_profile(b->prof[qi].data(), b->prof_letters[qi], q.data(), (int)q.size(), t, tlen, b->alphabet, b->go, b->ge, 0, 0, &r, dummy, 0);
        else mmo_sw_score_end(q.data(), (int)q.size(), b->cb[qi].data(), t, tlen, b->mat.data(), b->alphabet, b->go, b->ge, &r);
        mmgpu_sw_hit h;
        h.score = r.score;
        h.q_end = r.q_end;
        h.t_end = r.t_end;
        h.q_start = -1;
        h.t_start = -1;
        h.word = r.word;
        if (r.t_end == -1) {
            h.score = 0;
            h.q_end = 0;
        } else if (b->mode >= MMGPU_SW_START && r.score >= b->min_start[qi] && !(b->mode == MMGPU_SW_START_NOT_WORD && r.word != 0)) {
            if (isProf) mmo_sw_align_profile(b->prof[qi].data(), b->prof_letters[qi], q.data(), (int)q.size(), t, tlen, b->alphabet, b->go, b->ge, 1, 0, &r, dummy, 0);
            else mmo_sw_start(q.data(), (int)q.size(), b->cb[qi].data(), t, b->mat.data(), b->alphabet, b->go, b->ge, &r);
            h.q_start = r.q_start;
            h.t_start = r.t_start;
        }
        b->res[p] = h;
    }
    return 0;
}

int mmgpu_sw_fetch(mmgpu_ctx *, mmgpu_sw_batch_t *b, mmgpu_sw_hit *out) {
    if (!b->res.empty()) memcpy(out, b->res.data(), b->res.size() * sizeof(mmgpu_sw_hit));
    return 0;
}

// the reverse scan of the named pairs after the fact (mode MMGPU_SW_START_NOT_WORD: what the block aligner declined)
int mmgpu_sw_reverse_pairs(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *idx, uint32_t n, mmgpu_sw_hit *out) {
    if (b->mode < MMGPU_SW_START) return fail(MMGPU_ERR_STATE, "mmgpu_sw_reverse_pairs: the batch was prepared without a start-position mode");
    for (uint32_t i = 0; i < n; i++) {
        if (idx[i] >= b->pair.size()) return fail(MMGPU_ERR_ARG, "mmgpu_sw_reverse_pairs: pair index out of range");
        const uint32_t qi = b->pair[idx[i]].first, id = b->pair[idx[i]].second;
        const std::vector<uint8_t> &q = b->q[qi];
        const uint8_t *t = c->tres.data() + c->toff[id];
        const int tlen = (int)(c->toff[id + 1] - c->toff[id]);
        mmgpu_sw_hit &h = b->res[idx[i]];
        if (h.score > 0 && h.t_end >= 0) {
            mmo_sw_res r;
            char dummy[8];
            r.score = h.score; r.q_end = h.q_end; r.t_end = h.t_end; r.word = h.word; r.q_start = -1; r.t_start = -1;
            if (b->prof_letters[qi] > 0) mmo_sw_align_profile(b->prof[qi].data(), b->prof_letters[qi], q.data(), (int)q.size(), t, tlen, b->alphabet, b->go, b->ge, 1, 0, &r, dummy, 0);
            else mmo_sw_start(q.data(), (int)q.size(), b->cb[qi].data(), t, b->mat.data(), b->alphabet, b->go, b->ge, &r);
            h.q_start = r.q_start;
            h.t_start = r.t_start;
        }
        if (out) out[i] = h;
    }
    return 0;
}
void mmgpu_sw_free(mmgpu_ctx *, mmgpu_sw_batch_t *b) { delete b; }

// the device's block aligner (block_kernel.hip) stand-in: the plain-C restatement (oracle/block_oracle.c).
// MMGPU_EMU_REFUSE_BLOCK=<len>: decline pairs whose reversed prefixes are longer, like the device does beyond its scratch slot
int mmgpu_sw_block_backtrace(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *idx, uint32_t n, mmgpu_sw_block *out, char *bt, size_t cap,
                             size_t *used) {
    size_t need = 0;
    std::vector<size_t> off(n);
    for (uint32_t i = 0; i < n; i++) {
        const mmgpu_sw_hit &h = b->res[idx[i]];
        const bool word = h.score > 0 && h.word == 1 && h.t_end >= 0;
        off[i] = need;
        if (word) need += (size_t)h.q_end + 1 + (size_t)h.t_end + 1 + 1;
    }
    if (used) *used = need;
    const bool startsOnly = bt == NULL && cap == MMGPU_BLOCK_STARTS_ONLY;
    const bool noStrings = startsOnly || (bt == NULL && cap == MMGPU_BLOCK_NO_STRINGS);
    std::vector<char> scratch;
    if (noStrings) {
        scratch.resize(need + 1);
        bt = scratch.data();
    } else if (need && (bt == NULL || cap < need)) return fail(MMGPU_ERR_ARG, "bt buffer too small");
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t i = 0; i < n; i++) {
        const mmgpu_sw_hit &h = b->res[idx[i]];
        const uint32_t qi = b->pair[idx[i]].first, id = b->pair[idx[i]].second;
        memset(&out[i], 0, sizeof(out[i]));
        out[i].q_start = out[i].t_start = -1;
        out[i].bt_off = off[i];
        if (!(h.score > 0 && h.word == 1 && h.t_end >= 0)) {
            out[i].status = MMGPU_BLOCK_NOT_WORD;
            continue;
        }
        if (getenv("MMGPU_EMU_REFUSE_BLOCK") && h.q_end + h.t_end + 2 > atoi(getenv("MMGPU_EMU_REFUSE_BLOCK"))) {
            out[i].status = MMGPU_BLOCK_TOO_LARGE;
            continue;
        }
        const std::vector<uint8_t> &q = b->q[qi];
        const uint8_t *t = c->tres.data() + c->toff[id];
        const int tlen = (int)(c->toff[id + 1] - c->toff[id]);
        int qs = -1, ts = -1, len = 0, bs = 0;
        uint32_t ident = 0;
        int ok;
        if (b->prof_letters[qi] != 0) {      // profile query: its score rows with the X row neutral (ssw_init :1389-1391), the query on the crate's reference side
            std::vector<int8_t> rows((size_t)b->alphabet * q.size(), 0);
            const int letters = std::min(b->prof_letters[qi], b->alphabet - 1);
            memcpy(rows.data(), b->prof[qi].data(), (size_t)letters * q.size());
            ok = mmo_sw_block_backtrace_profile(rows.data(), q.data(), (int)q.size(), t, tlen, b->alphabet, b->go, b->ge, h.score, h.q_end, h.t_end, &qs, &ts,
                                                &ident, bt + off[i], h.q_end + h.t_end + 3, &len, &bs);
        } else {
            ok = mmo_sw_block_backtrace(q.data(), b->cb[qi].data(), (int)q.size(), t, tlen, b->mat.data(), b->alphabet, b->go, b->ge, h.score,
                                        h.q_end, h.t_end, &qs, &ts, &ident, bt + off[i], h.q_end + h.t_end + 3, &len, &bs);
        }
        if (!ok) {
            out[i].status = MMGPU_BLOCK_DECLINED;
            continue;
        }
        out[i].status = MMGPU_BLOCK_OK;
        out[i].q_start = qs;
        out[i].t_start = ts;
        out[i].ident = startsOnly ? 0 : ident;      // (the device keeps no trace in that mode: both come back 0)
        out[i].bt_len = startsOnly ? 0 : (uint32_t)len;
    }
    return 0;
}

// search semantics in one call: select, block aligner for start positions, reverse scan of what it declined
int mmgpu_sw_block_starts(mmgpu_ctx *c, mmgpu_sw_batch_t *b, uint32_t *n_selected, uint32_t *n_declined, uint32_t *n_too_large) {
    if (b->mode != MMGPU_SW_START_NOT_WORD) return fail(MMGPU_ERR_STATE, "mmgpu_sw_block_starts: the batch was not prepared with MMGPU_SW_START_NOT_WORD");
    std::vector<uint32_t> sel;
    for (size_t p = 0; p < b->res.size(); p++) {
        const mmgpu_sw_hit &h = b->res[p];
        if (h.score > 0 && h.word == 1 && h.t_end >= 0 && h.score >= b->min_start[b->pair[p].first]) sel.push_back((uint32_t)p);
    }
    std::vector<mmgpu_sw_block> blk(sel.size());
    size_t used = 0;
    int rc = sel.empty() ? 0 : mmgpu_sw_block_backtrace(c, b, sel.data(), (uint32_t)sel.size(), blk.data(), NULL, MMGPU_BLOCK_STARTS_ONLY, &used);
    if (rc != 0) return rc;
    std::vector<uint32_t> declined;
    uint32_t tooLarge = 0;
    for (size_t k = 0; k < sel.size(); k++) {
        if (blk[k].status == MMGPU_BLOCK_OK) { b->res[sel[k]].q_start = blk[k].q_start; b->res[sel[k]].t_start = blk[k].t_start; }
        else if (blk[k].status == MMGPU_BLOCK_DECLINED) declined.push_back(sel[k]);
        else tooLarge++;
    }
    if (!declined.empty()) rc = mmgpu_sw_reverse_pairs(c, b, declined.data(), (uint32_t)declined.size(), NULL);
    if (n_selected) *n_selected = (uint32_t)sel.size();
    if (n_declined) *n_declined = (uint32_t)declined.size();
    if (n_too_large) *n_too_large = tooLarge;
    return rc;
}

int mmgpu_sw_block_growth(mmgpu_ctx *, mmgpu_sw_batch_t *, const uint32_t *, uint32_t, mmgpu_sw_block *, uint32_t *, uint32_t) {
    return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_block_growth: a test aid of the device kernel");
}

int mmgpu_sw_block_tiers(const mmgpu_sw_batch_t *, uint32_t *first_tier, uint32_t *second_tier) {
    if (first_tier) *first_tier = 0;
    if (second_tier) *second_tier = 0;
    return 0;
}

int mmgpu_sw_traceback(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *idx, uint32_t n, mmgpu_sw_bt *info, char *bt, size_t cap,
                       size_t *used) {
    size_t need = 0;
    std::vector<size_t> off(n);
    for (uint32_t i = 0; i < n; i++) {
        const mmgpu_sw_hit &h = b->res[idx[i]];
        off[i] = need;
        if (h.q_start >= 0) need += (size_t)(h.q_end - h.q_start + 1) + (size_t)(h.t_end - h.t_start + 1) + 1;
        else need += 1;
    }
    *used = need;
    if (bt == NULL || cap < need) return fail(MMGPU_ERR_ARG, "bt buffer too small");
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t i = 0; i < n; i++) {
        const mmgpu_sw_hit &h = b->res[idx[i]];
        info[i].bt_off = off[i];
        info[i].bt_len = 0;
        info[i].ident = 0;
        info[i].reserved = 0;
        if (h.q_start < 0) {
            info[i].status = MMGPU_BT_NO_START;
            continue;
        }
        // test switch: decline long backtraces like the device does when the band storage exceeds its scratch budget,
        // so that the host's fallback (Matcher::getSWResult for the pair) is exercised without a GPU
        if (getenv("MMGPU_EMU_REFUSE_BT") && (h.q_end - h.q_start + 1) + (h.t_end - h.t_start + 1) > atoi(getenv("MMGPU_EMU_REFUSE_BT"))) {
            info[i].status = MMGPU_BT_TOO_LARGE;
            continue;
        }
        const uint32_t qi = b->pair[idx[i]].first, id = b->pair[idx[i]].second;
        const std::vector<uint8_t> &q = b->q[qi];
        const uint8_t *t = c->tres.data() + c->toff[id];
        const int ql = h.q_end - h.q_start + 1, tl = h.t_end - h.t_start + 1;
        int len;
        if (b->prof_letters[qi] > 0) {
            mmo_sw_res r;
            const int tlen = (int)(c->toff[id + 1] - c->toff[id]);
            const int rc = mmo_sw_align_profile(b->prof[qi].data(), b->prof_letters[qi], q.data(), (int)q.size(), t, tlen, b->alphabet, b->go, b->ge,
                                                1, 1, &r, bt + off[i], ql + tl + 1);
            len = rc == 0 ? r.bt_len : -1;
        } else {
            len = mmo_sw_banded_backtrace(t + h.t_start, q.data() + h.q_start, b->cb[qi].data() + h.q_start, tl, ql, h.score, b->go,
                                          b->ge, b->mat.data(), b->alphabet, bt + off[i], ql + tl + 1);
        }
        if (len < 0) {
            info[i].status = MMGPU_BT_FAILED;
            continue;
        }
        int qp = h.q_start, tp = h.t_start;
        uint32_t ident = 0;
        for (int k = 0; k < len; k++) {
            const char ch = bt[off[i] + k];
            if (ch == 'M') {
                ident += q[qp] == t[tp];
                qp++;
                tp++;
            } else if (ch == 'I') qp++;
            else tp++;
        }
        info[i].bt_len = (uint32_t)len;
        info[i].ident = ident;
        info[i].status = MMGPU_BT_OK;
    }
    return 0;
}

int mmgpu_pf_load_index(mmgpu_ctx *c, const mmgpu_pf_index *ix) {
    if (c->n == 0) return fail(MMGPU_ERR_STATE, "no targets");
    c->k = ix->kmer_size;
    c->spaced = ix->spaced;
    c->kalph = ix->alphabet - 1;
    c->kbase = ix->kmer_alphabet > 0 ? ix->kmer_alphabet : c->kalph;
    const size_t n3 = (size_t)c->kalph * c->kalph * c->kalph, n2 = (size_t)c->kalph * c->kalph;
    c->s3.clear();
    c->i3.clear();
    if (ix->score3 && ix->index3) {       // absent: exact k-mer matching only (nucleotide databases)
        c->s3.resize(n3 * n3);
        c->i3.resize(n3 * n3);
        for (size_t r = 0; r < n3; r++) {
            memcpy(&c->s3[r * n3], ix->score3 + r * ix->row3, n3 * sizeof(int16_t));
            memcpy(&c->i3[r * n3], ix->index3 + r * ix->row3, n3 * sizeof(uint32_t));
        }
    }
    c->s2.clear();
    c->i2.clear();
    if (ix->score2) {
        c->s2.resize(n2 * n2);
        c->i2.resize(n2 * n2);
        for (size_t r = 0; r < n2; r++) {
            memcpy(&c->s2[r * n2], ix->score2 + r * ix->row2, n2 * sizeof(int16_t));
            memcpy(&c->i2[r * n2], ix->index2 + r * ix->row2, n2 * sizeof(uint32_t));
        }
    }
    size_t nk = 1;
    for (int i = 0; i < c->k; i++) nk *= c->kbase;
    c->offsets.assign(ix->offsets, ix->offsets + nk + 1);
    c->ids.resize(ix->n_entries);
    c->pos.resize(ix->n_entries);
    if (ix->entries6) {
        const uint8_t *e = (const uint8_t *)ix->entries6;
        for (uint64_t i = 0; i < ix->n_entries; i++) {
            memcpy(&c->ids[i], e + i * 6, 4);
            memcpy(&c->pos[i], e + i * 6 + 4, 2);
        }
    } else {
        memcpy(c->ids.data(), ix->entry_ids, ix->n_entries * 4);
        memcpy(c->pos.data(), ix->entry_pos, ix->n_entries * 2);
    }
    c->ungapped.assign(ix->ungapped_mat, ix->ungapped_mat + ix->alphabet * ix->alphabet);
    c->gen.k = c->k;
    c->gen.kalph = c->kalph;
    c->gen.s3 = c->s3.empty() ? NULL : c->s3.data();
    c->gen.i3 = c->i3.empty() ? NULL : c->i3.data();
    c->gen.s2 = c->s2.empty() ? NULL : c->s2.data();
    c->gen.i2 = c->i2.empty() ? NULL : c->i2.data();
    c->have_index = true;
    return 0;
}

int mmgpu_warmup(mmgpu_ctx *) { return 0; }

// the device's tantan masking stand-in: the plain-C restatement (oracle/tantan_oracle.c)
int mmgpu_pf_mask_targets(mmgpu_ctx *c, const double *lr, int alphabet, double min_mask_prob, int mask_letter, uint64_t *n_masked) {
    if (c->n == 0 && c->tres.empty()) return fail(MMGPU_ERR_STATE, "no targets");
    if (!lr) {      // back to the unmasked view
        c->pf_tres.clear();
        c->have_index = false;
        if (n_masked) *n_masked = 0;
        return 0;
    }
    c->pf_tres = c->tres;
    uint64_t masked = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : masked)
    for (uint32_t i = 0; i < c->n; i++)
        masked += (uint64_t)mmo_tantan_mask(c->pf_tres.data() + c->toff[i], (int)(c->toff[i + 1] - c->toff[i]), lr, alphabet, min_mask_prob,
                                            (uint8_t)mask_letter, NULL);
    if (n_masked) *n_masked = masked;
    c->have_index = false;
    return 0;
}
int mmgpu_pf_debug_masked_targets(mmgpu_ctx *c, const uint64_t *offsets, uint32_t n, uint8_t *residues) {
    if (n != c->n) return fail(MMGPU_ERR_STATE, "not the resident target set");
    memcpy(residues, c->pf_res(), (size_t)offsets[n]);
    return 0;
}

// IndexBuilder::fillDatabase's index from the resident (already masked) targets: the oracle's builder
int mmgpu_pf_build_index(mmgpu_ctx *c, const mmgpu_pf_index *ix, const int16_t *kmer_submat, int kmer_thr) {
    if (c->n == 0) return fail(MMGPU_ERR_STATE, "no targets");
    mmgpu_pf_index full = *ix;
    size_t nk = 1;
    for (int i = 0; i < ix->kmer_size; i++) nk *= (size_t)(ix->alphabet - 1);
    std::vector<uint64_t> off(nk + 1);
    const uint64_t total = mmo_pf_index_build(c->pf_res(), c->toff.data(), c->n, kmer_submat, ix->alphabet, ix->kmer_size, ix->spaced,
                                              kmer_thr, off.data(), NULL, NULL);
    std::vector<uint32_t> ids(total + 1);
    std::vector<uint16_t> pos(total + 1);
    mmo_pf_index_build(c->pf_res(), c->toff.data(), c->n, kmer_submat, ix->alphabet, ix->kmer_size, ix->spaced, kmer_thr, off.data(),
                       ids.data(), pos.data());
    full.offsets = off.data();
    full.entry_ids = ids.data();
    full.entry_pos = pos.data();
    full.entries6 = NULL;
    full.n_entries = total;
    full.kmer_alphabet = 0;      // (an index built here is over alphabet - 1 letters)
    return mmgpu_pf_load_index(c, &full);
}

// persisted layout (include/mmgpu.h "persisted device layout"): the emulation writes what it holds in a format of its own - the
// hooks' logic around the calls (fingerprints, fall-back to building, save after a build) is what the CPU tests exercise
struct EmuDbHeader {
    char magic[8];
    uint64_t source_fp, index_fp, n, alphabet, res_bytes, masked_bytes, table, n_entries;
    int32_t k, spaced, kalph, kbase;
};
int mmgpu_db_save(mmgpu_ctx *c, const char *path, uint64_t sfp, uint64_t ifp) {
    if (c->n == 0) return fail(MMGPU_ERR_STATE, "no targets");
    EmuDbHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "MMGPUEMU", 8);
    const bool idx = ifp != 0 && c->have_index && c->kbase == c->kalph;
    h.source_fp = sfp; h.index_fp = idx ? ifp : 0; h.n = c->n; h.alphabet = (uint64_t)c->alphabet;
    h.res_bytes = c->tres.size(); h.masked_bytes = c->pf_tres.size();
    if (idx) { h.table = c->offsets.size(); h.n_entries = c->ids.size(); h.k = c->k; h.spaced = c->spaced; h.kalph = c->kalph; h.kbase = c->kbase; }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(MMGPU_ERR_ARG, "cannot create the file");
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(c->toff.data(), 8, c->n + 1, f) == c->n + 1 &&
              (c->tres.empty() || fwrite(c->tres.data(), 1, c->tres.size(), f) == c->tres.size()) &&
              (c->pf_tres.empty() || fwrite(c->pf_tres.data(), 1, c->pf_tres.size(), f) == c->pf_tres.size());
    if (ok && idx)
        ok = fwrite(c->offsets.data(), 8, c->offsets.size(), f) == c->offsets.size() &&
             (c->ids.empty() || (fwrite(c->ids.data(), 4, c->ids.size(), f) == c->ids.size() && fwrite(c->pos.data(), 2, c->pos.size(), f) == c->pos.size()));
    ok = fclose(f) == 0 && ok;
    return ok ? 0 : fail(MMGPU_ERR_ARG, "write failed");
}
static int emu_read_header(const char *path, EmuDbHeader *h, FILE **out) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MMGPU_ERR_STATE, "no such file");
    if (fread(h, sizeof(*h), 1, f) != 1 || memcmp(h->magic, "MMGPUEMU", 8) != 0) {
        fclose(f);
        return fail(MMGPU_ERR_STATE, "not a database file of the emulation");
    }
    if (out) *out = f; else fclose(f);
    return 0;
}
int mmgpu_db_probe(const char *path, mmgpu_db_info *info) {
    EmuDbHeader h;
    const int rc = emu_read_header(path, &h, NULL);
    if (rc != 0) return rc;
    memset(info, 0, sizeof(*info));
    info->source_fingerprint = h.source_fp; info->index_fingerprint = h.index_fp; info->n_targets = (uint32_t)h.n;
    info->alphabet = (uint32_t)h.alphabet; info->total_residues = h.res_bytes; info->has_masked_view = h.masked_bytes != 0;
    info->has_index = h.index_fp != 0; info->kmer_size = h.k; info->spaced = h.spaced; info->n_entries = h.n_entries;
    return 0;
}
int mmgpu_db_load(mmgpu_ctx *c, const char *path, uint64_t sfp, uint64_t ifp, const mmgpu_pf_index *tables) {
    EmuDbHeader h;
    FILE *f = NULL;
    const int rc = emu_read_header(path, &h, &f);
    if (rc != 0) return rc;
    if (h.source_fp != sfp || (ifp != 0 && h.index_fp != ifp) || (ifp != 0 && !tables)) {
        fclose(f);
        return fail(MMGPU_ERR_STATE, "fingerprint differs");
    }
    std::vector<uint64_t> toff(h.n + 1), offsets(ifp ? h.table : 0);
    std::vector<uint8_t> tres(h.res_bytes), masked(h.masked_bytes);
    std::vector<uint32_t> ids(ifp ? h.n_entries : 0);
    std::vector<uint16_t> pos(ifp ? h.n_entries : 0);
    bool ok = fread(toff.data(), 8, h.n + 1, f) == h.n + 1 && (tres.empty() || fread(tres.data(), 1, tres.size(), f) == tres.size()) &&
              (masked.empty() || fread(masked.data(), 1, masked.size(), f) == masked.size());
    if (ok && ifp)
        ok = fread(offsets.data(), 8, offsets.size(), f) == offsets.size() &&
             (ids.empty() || (fread(ids.data(), 4, ids.size(), f) == ids.size() && fread(pos.data(), 2, pos.size(), f) == pos.size()));
    fclose(f);
    if (!ok) return fail(MMGPU_ERR_STATE, "short file");
    mmgpu_load_targets(c, tres.data(), toff.data(), (uint32_t)h.n, (int)h.alphabet);
    if (ifp) {
        c->pf_tres.swap(masked);
        mmgpu_pf_index full = *tables;
        full.offsets = offsets.data();
        full.entry_ids = ids.data();
        full.entry_pos = pos.data();
        full.entries6 = NULL;
        full.n_entries = h.n_entries;
        full.kmer_alphabet = 0;
        return mmgpu_pf_load_index(c, &full);
    }
    return 0;
}

int mmgpu_pf_prepare(mmgpu_ctx *c, const mmgpu_pf_params *p, const mmgpu_pf_query *qs, uint32_t nq, mmgpu_pf_batch_t **out) {
    if (!c->have_index) return fail(MMGPU_ERR_STATE, "no index");
    // test knob: behave like a device whose memory holds at most this many queries per batch (the host must cut the block)
    if (const char *e = getenv("MMGPU_EMU_MAX_BATCH"))
        if (nq > (uint32_t)atoi(e)) return fail(MMGPU_ERR_HIP, "hipMalloc: out of memory (emulated)");
    mmgpu_pf_batch_t *b = new mmgpu_pf_batch_t();
    b->par = *p;
    for (uint32_t i = 0; i < nq; i++) {
        b->q.push_back(std::vector<uint8_t>(qs[i].q, qs[i].q + qs[i].qlen));
        if (qs[i].comp_bias) b->bias.push_back(std::vector<float>(qs[i].comp_bias, qs[i].comp_bias + qs[i].qlen));
        else b->bias.push_back(std::vector<float>(qs[i].qlen, 0.0f));
        b->identity.push_back(qs[i].identity_id);
        if (qs[i].profile) {
            const size_t n = (size_t)qs[i].qlen * qs[i].profile_row;
            b->pscore.push_back(std::vector<int16_t>(qs[i].profile_score, qs[i].profile_score + n));
            b->pindex.push_back(std::vector<uint32_t>(qs[i].profile_index, qs[i].profile_index + n));
            b->paln.push_back(std::vector<int8_t>(qs[i].profile, qs[i].profile + (size_t)20 * qs[i].qlen));
            b->prow.push_back((int)qs[i].profile_row);
        } else {
            b->pscore.push_back(std::vector<int16_t>());
            b->pindex.push_back(std::vector<uint32_t>());
            b->paln.push_back(std::vector<int8_t>());
            b->prow.push_back(0);
        }
    }
    b->hits.resize(nq);
    b->stats.resize(nq);
    *out = b;
    return 0;
}

int mmgpu_pf_run(mmgpu_ctx *c, mmgpu_pf_batch_t *b) {
    mmo_pf_params P;
    memset(&P, 0, sizeof(P));
    P.gen = &c->gen;
    P.alphabet = c->alphabet;
    P.spaced = c->spaced;
    P.kmer_thr = b->par.kmer_thr;
    P.offsets = c->offsets.data();
    P.ids = c->ids.data();
    P.pos = c->pos.data();
    P.tdata = c->pf_res();
    P.toff = c->toff.data();
    P.n_targets = c->n;
    P.ungapped_mat = c->ungapped.data();
    P.bins = b->par.ref_bins;
    P.max_hits = b->par.max_hits;
    P.min_diag_score = b->par.min_diag_score;
    P.exact_kmer = (int)b->par.exact_kmer;
    P.index_base = c->kbase;
    P.nucleotide = (int)b->par.nucleotide;
    P.kmer_score = (int)b->par.kmer_score;
    const size_t cap = (size_t)std::min<uint64_t>(b->par.max_hits, c->n) + 1;
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < b->q.size(); i++) {
        b->hits[i].resize(cap);
        uint64_t nh = 0;
        int rc;
        if (b->prow[i] > 0) {
            mmo_pf_profile pr;
            pr.score = b->pscore[i].data();
            pr.index = b->pindex[i].data();
            pr.row = b->prow[i];
            pr.aln = b->paln[i].data();
            rc = mmo_pf_match_query_profile(&P, b->q[i].data(), (int)b->q[i].size(), &pr, b->identity[i], b->hits[i].data(), cap, &nh,
                                            &b->stats[i], NULL);
        } else {
            rc = mmo_pf_match_query(&P, b->q[i].data(), (int)b->q[i].size(), b->bias[i].data(), b->identity[i], b->hits[i].data(), cap, &nh,
                                    &b->stats[i], NULL);
        }
        if (rc != 0)
            bad = 1;
        b->hits[i].resize(nh);
    }
    return bad ? fail(MMGPU_ERR_ARG, "mmo_pf_match_query failed") : 0;
}

int mmgpu_pf_fetch(mmgpu_ctx *c, mmgpu_pf_batch_t *b, mmgpu_pf_hit *hits, uint32_t stride, uint32_t *counts, int32_t *status,
                   mmgpu_pf_qstat *stats) {
    // sequences of 32768 residues or more are not restated in the oracle (UngappedAlignment::computeLongScore): like the
    // device, hand such queries back to the host - here, conservatively, every query once the database holds a long target
    bool long_target = false;
    for (uint32_t t = 0; t < c->n; t++) long_target |= c->toff[t + 1] - c->toff[t] >= 32768;
    for (size_t i = 0; i < b->q.size(); i++) {
        counts[i] = (uint32_t)b->hits[i].size();
        status[i] = (long_target || b->q[i].size() >= 32768) ? MMGPU_PF_LONG_SEQ : MMGPU_PF_OK;
        // (ties among up to 16 saturated elements: the restatement's stable choice is the reference's, as on the device; beyond, the
        // device replays std::sort - the stand-in hands the query back)
        if (status[i] == MMGPU_PF_OK && ((b->stats[i].sat_tie && b->stats[i].sat_len > 16) || (b->par.nucleotide && b->stats[i].overflow))) status[i] = MMGPU_PF_SAT_TIE;
        if (status[i] == MMGPU_PF_OK && b->par.kmer_score && b->stats[i].overflow) status[i] = MMGPU_PF_OVERFLOW;
        if (status[i] == MMGPU_PF_OK && b->par.kmer_score && b->stats[i].big_list) status[i] = MMGPU_PF_SAT_TIE;
        if (status[i] != MMGPU_PF_OK) counts[i] = 0;
        for (size_t k = 0; k < b->hits[i].size() && k < stride; k++) {
            mmgpu_pf_hit &o = hits[i * (size_t)stride + k];
            o.id = b->hits[i][k].id;
            o.score = b->hits[i][k].score;
            o.diagonal = b->hits[i][k].diagonal;
            o.reserved = 0;
        }
        if (stats) {
            stats[i].db_matches = b->stats[i].db_matches;
            stats[i].kmer_list_len = b->stats[i].kmer_list_len;
            stats[i].double_hits = (uint32_t)b->stats[i].after_keepmax;
            stats[i].diag_thr = b->stats[i].diag_thr;
        }
    }
    return 0;
}
void mmgpu_pf_free(mmgpu_ctx *, mmgpu_pf_batch_t *b) { delete b; }

// nucleotide alignment step: the restatement of BandedNucleotideAligner::align per pair (nucl_oracle.c)
int mmgpu_nucl_align(mmgpu_ctx *c, const mmgpu_nucl_params *par, const mmgpu_nucl_query *qs, uint32_t nq, const mmgpu_nucl_pair *pairs,
                     uint32_t np, mmgpu_nucl_hit *out, char *bt, uint64_t bt_cap, uint64_t *bt_used) {
    if (c->alphabet != 5) {
        g_err = "mmgpu_nucl_align (emu): the resident targets are not nucleotides";
        return MMGPU_ERR_ARG;
    }
    uint64_t used = 0;
    for (uint32_t i = 0; i < np; i++) {
        const mmgpu_nucl_pair &p = pairs[i];
        if (p.query >= nq || p.target >= c->n) {
            g_err = "mmgpu_nucl_align (emu): pair index out of range";
            return MMGPU_ERR_ARG;
        }
        const int qlen = (int)qs[p.query].qlen, tlen = (int)(c->toff[p.target + 1] - c->toff[p.target]);
        const int pq = (p.past_end & 0x80u) ? (int)(p.past_end & 7u) : par->past_end_query;
        const int pt = (p.past_end & 0x80u) ? (int)((p.past_end >> 3) & 7u) : par->past_end_target;
        std::vector<char> s((size_t)qlen + tlen + 2);
        mmo_nucl_result r;
        mmo_nucl_align_wrapped(qs[p.query].q, qlen, c->tres.data() + c->toff[p.target], tlen, par->mat, 5, par->reverse, par->gap_open,
                               par->gap_extend, par->zdrop, p.diagonal, p.reverse, pq, pt, par->wrapped ? 1 : 0, &r, s.data(), (int)s.size());
        mmgpu_nucl_hit &o = out[i];
        o.score = r.score; o.q_start = r.q_start; o.q_end = r.q_end; o.t_start = r.t_start; o.t_end = r.t_end;
        o.ident = r.ident; o.bt_len = (uint32_t)r.bt_len; o.bt_off = used;
        o.status = MMGPU_NUCL_OK;
        if (used + (uint64_t)r.bt_len + 1 > bt_cap) o.status = MMGPU_NUCL_BT_OVERFLOW;
        else {
            memcpy(bt + used, s.data(), (size_t)r.bt_len);
            bt[used + r.bt_len] = '\0';
        }
        used += (uint64_t)r.bt_len + 1;
    }
    if (bt_used) *bt_used = used;
    return 0;
}

// ---- several contexts (mmgpu_init_multi): what the hooks' MMGPU_DEVICES / large-split paths drive.  The contract of the real
// library is that the merged lists of the shards EQUAL the unsplit run's (tests/test_sharded_gpu.py pins that on the device), so
// the stand-in keeps the whole database on context 0 and answers from there; the other contexts exist for the alignment hook,
// which loads the targets on every context and deals its queries to them.
struct mmgpu_multi { std::vector<mmgpu_ctx *> ctx; };
struct mmgpu_multi_pf_batch { mmgpu_pf_batch_t *b; };

int mmgpu_init_multi(mmgpu_multi **out, const int *ids, int n) {
    if (!out || !ids || n < 1) return fail(MMGPU_ERR_ARG, "mmgpu_init_multi (emu): bad argument");
    mmgpu_multi *m = new mmgpu_multi();
    for (int i = 0; i < n; i++) {
        mmgpu_ctx *c = NULL;
        mmgpu_init(&c, ids[i]);
        m->ctx.push_back(c);
    }
    *out = m;
    return 0;
}
void mmgpu_destroy_multi(mmgpu_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); i++) mmgpu_destroy(m->ctx[i]);
    delete m;
}
int mmgpu_multi_size(mmgpu_multi *m) { return m ? (int)m->ctx.size() : 0; }
mmgpu_ctx *mmgpu_multi_ctx(mmgpu_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : NULL; }
int mmgpu_multi_synchronize(mmgpu_multi *) { return 0; }
int mmgpu_comm_info(mmgpu_ctx *, int *rank, int *n_ranks, char *transport, int cap) {
    if (rank) *rank = 0;
    if (n_ranks) *n_ranks = 1;
    if (transport && cap > 0) snprintf(transport, cap, "emulated");
    return 0;
}
int mmgpu_multi_load_targets(mmgpu_multi *m, const uint8_t *res, const uint64_t *off, uint32_t n, int alphabet) {
    return mmgpu_load_targets(m->ctx[0], res, off, n, alphabet);
}
int mmgpu_multi_pf_mask_targets(mmgpu_multi *m, const double *lr, int alphabet, double min_mask_prob, int mask_letter, uint64_t *n_masked) {
    return mmgpu_pf_mask_targets(m->ctx[0], lr, alphabet, min_mask_prob, mask_letter, n_masked);
}
int mmgpu_multi_pf_build_index(mmgpu_multi *m, const mmgpu_pf_index *ix, const int16_t *kmer_submat, int kmer_thr) {
    return mmgpu_pf_build_index(m->ctx[0], ix, kmer_submat, kmer_thr);
}
int mmgpu_multi_pf_prepare(mmgpu_multi *m, const mmgpu_pf_params *p, const mmgpu_pf_query *qs, uint32_t nq, mmgpu_multi_pf_batch **out) {
    mmgpu_pf_batch_t *b = NULL;
    const int rc = mmgpu_pf_prepare(m->ctx[0], p, qs, nq, &b);
    if (rc != 0) return rc;
    *out = new mmgpu_multi_pf_batch();
    (*out)->b = b;
    return 0;
}
int mmgpu_multi_pf_run(mmgpu_multi *m, mmgpu_multi_pf_batch *mb) { return mmgpu_pf_run(m->ctx[0], mb->b); }
int mmgpu_multi_pf_fetch(mmgpu_multi *m, mmgpu_multi_pf_batch *mb, mmgpu_pf_hit *hits, uint32_t stride, uint32_t *counts, int32_t *status) {
    return mmgpu_pf_fetch(m->ctx[0], mb->b, hits, stride, counts, status, NULL);
}
int mmgpu_multi_has_unsplit(mmgpu_multi *) { return 1; }      // (the stand-in answers from the whole database in the first place)
int mmgpu_multi_pf_redone(mmgpu_multi_pf_batch *, uint32_t *n_redone, uint32_t *n_left) {
    if (n_redone) *n_redone = 0;
    if (n_left) *n_left = 0;
    return 0;
}
void mmgpu_multi_pf_free(mmgpu_multi *m, mmgpu_multi_pf_batch *mb) {
    if (!mb) return;
    mmgpu_pf_free(m->ctx[0], mb->b);
    delete mb;
}

}  // extern "C"
