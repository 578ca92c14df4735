// libmmgpu_client.so: the C-ABI entry points the reference-side binding calls (integration/*.cpp), served by a resident
// mmgpu_server process instead of a device context of this process (mmgpu_wire.h; reference counterpart: the client half
// of GPUSharedMemory, src/commons/GpuUtil.cpp, used by ungappedprefilter.cpp:205-260 when --gpu-server is set).
// Same symbol names and argument meaning as libmmgpu.so: a binary linked against libmmgpu.so is switched over with
// LD_PRELOAD=libmmgpu_client.so (or by linking this library instead); MMGPU_SERVER_SOCKET names the server's socket.
// No HIP in this file: the process never opens the device.
#include <stdio.h>
#include <stdlib.h>
#include <sys/socket.h>
#include <sys/un.h>

#include <string>
#include <vector>

#include "../../include/mmgpu.h"
#include "mmgpu_wire.h"

using namespace mmgpu_wire;

struct mmgpu_ctx {
    int fd;
    int cus;
    std::string name;
    uint64_t targets_fp;
    uint64_t mask_fp;      // 0 = the prefilter reads the targets as loaded; else the tantan parameters they were masked with
};
struct mmgpu_sw_batch_t {
    uint64_t handle;
    uint64_t pairs;
};
struct mmgpu_pf_batch_t {
    uint64_t handle;
    uint32_t nq;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &m) {
    g_err = m;
    return code;
}

// one request / reply; on a library error the reply payload is the server's mmgpu_last_error()
int call(mmgpu_ctx *c, uint32_t op, const Buf &req, Buf *rep) {
    if (!c || c->fd < 0) return fail(MMGPU_ERR_STATE, "mmgpu client: not connected");
    WireHdr h;
    Buf local;
    Buf *r = rep ? rep : &local;
    if (!send_msg(c->fd, op, 0, req.d.data(), req.d.size()) || !recv_msg(c->fd, &h, r))
        return fail(MMGPU_ERR_STATE, "mmgpu client: connection to mmgpu_server lost");
    if (h.status != MMGPU_OK) return fail(h.status, std::string(reinterpret_cast<const char *>(r->d.data()), r->d.size()));
    return MMGPU_OK;
}

}  // namespace

extern "C" {

const char *mmgpu_last_error(void) { return g_err.c_str(); }

int mmgpu_init(mmgpu_ctx **out, int device_id) {
    if (!out) return fail(MMGPU_ERR_ARG, "mmgpu_init: NULL argument");
    const char *path = getenv("MMGPU_SERVER_SOCKET");
    if (!path || !*path) return fail(MMGPU_ERR_STATE, "mmgpu client: MMGPU_SERVER_SOCKET is not set");
    sockaddr_un a;
    memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    if (strlen(path) >= sizeof(a.sun_path)) return fail(MMGPU_ERR_ARG, "mmgpu client: socket path too long");
    strcpy(a.sun_path, path);
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0 || connect(fd, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0) {
        if (fd >= 0) close(fd);
        return fail(MMGPU_ERR_STATE, std::string("mmgpu client: cannot connect to mmgpu_server at ") + path + ": " + strerror(errno));
    }
    mmgpu_ctx *c = new mmgpu_ctx();
    c->fd = fd;
    c->cus = 0;
    c->targets_fp = 0;
    Buf req, rep;
    req.put<int32_t>(device_id);
    const int rc = call(c, OP_HELLO, req, &rep);
    if (rc != MMGPU_OK) {
        close(fd);
        delete c;
        return rc;
    }
    c->cus = rep.get<int32_t>();
    size_t n = 0;
    const uint8_t *s = rep.get_bytes(&n);
    c->name.assign(reinterpret_cast<const char *>(s), n);
    *out = c;
    return MMGPU_OK;
}

void mmgpu_destroy(mmgpu_ctx *c) {
    if (!c) return;
    if (c->fd >= 0) close(c->fd);      // the server goes back to accept(); the database stays resident
    delete c;
}

int mmgpu_device_info(mmgpu_ctx *c, int *cus, char *name, int cap) {
    if (!c) return fail(MMGPU_ERR_ARG, "mmgpu_device_info: NULL argument");
    if (cus) *cus = c->cus;
    if (name && cap > 0) snprintf(name, (size_t)cap, "%s (mmgpu_server)", c->name.c_str());
    return MMGPU_OK;
}

int mmgpu_synchronize(mmgpu_ctx *) { return MMGPU_OK; }

int mmgpu_load_targets(mmgpu_ctx *c, const uint8_t *res, const uint64_t *off, uint32_t n, int alphabet) {
    if (!c || !off || (!res && n)) return fail(MMGPU_ERR_ARG, "mmgpu_load_targets: NULL argument");
    uint64_t fp = fingerprint(off, ((size_t)n + 1) * 8, 0x7461726765747321ull ^ (uint64_t)alphabet);
    fp = fingerprint(res, (size_t)off[n], fp);
    c->targets_fp = fp;
    c->mask_fp = 0;
    Buf q, r;
    q.put<uint64_t>(fp);
    q.put<uint32_t>(n);
    q.put<int32_t>(alphabet);
    int rc = call(c, OP_HAS_TARGETS, q, &r);
    if (rc != MMGPU_OK) return rc;
    if (r.get<uint32_t>()) return MMGPU_OK;      // resident: nothing to send
    q.put_bytes(off, ((size_t)n + 1) * 8);
    q.put_bytes(res, (size_t)off[n]);
    return call(c, OP_LOAD_TARGETS, q, nullptr);
}

// tantan masking of the resident targets for the prefilter: the server remembers with which parameters a slot's targets are
// masked, a client that asks for the same again (the next `mmseqs prefilter` of the same database) finds it done
int mmgpu_pf_mask_targets(mmgpu_ctx *c, const double *lr, int alphabet, double min_mask_prob, int mask_letter, uint64_t *n_masked) {
    if (!c || !lr || alphabet < 1 || alphabet > 64) return fail(MMGPU_ERR_ARG, "mmgpu_pf_mask_targets: bad argument");
    uint64_t fp = fingerprint(lr, (size_t)alphabet * alphabet * 8, 0x74616E74616E2121ull ^ (uint64_t)mask_letter);
    fp = fingerprint(&min_mask_prob, 8, fp) | 1ull;
    Buf q, r;
    q.put<uint64_t>(fp);
    q.put<int32_t>(alphabet);
    q.put<int32_t>(mask_letter);
    q.put_bytes(&min_mask_prob, 8);
    q.put_bytes(lr, (size_t)alphabet * alphabet * 8);
    const int rc = call(c, OP_MASK_TARGETS, q, &r);
    if (rc != MMGPU_OK) return rc;
    c->mask_fp = fp;
    const uint64_t masked = r.get<uint64_t>();
    if (n_masked) *n_masked = masked;
    return MMGPU_OK;
}

// nothing to load on this side of the socket
// the persisted device layout is a matter of the process that owns the device: the resident server keeps its databases between
// commands anyway (found by fingerprint), so a client neither saves nor loads files - "no such file" makes the caller build as ever
int mmgpu_db_save(mmgpu_ctx *, const char *, uint64_t, uint64_t) {
    return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_db_save: not served through mmgpu_server (nothing written; the server keeps databases resident itself)");
}
int mmgpu_db_probe(const char *, mmgpu_db_info *) { return fail(MMGPU_ERR_STATE, "mmgpu_db_probe: not served through mmgpu_server"); }
int mmgpu_db_load(mmgpu_ctx *, const char *, uint64_t, uint64_t, const mmgpu_pf_index *) {
    return fail(MMGPU_ERR_STATE, "mmgpu_db_load: not served through mmgpu_server (the server keeps databases resident itself)");
}

int mmgpu_warmup(mmgpu_ctx *) { return MMGPU_OK; }

int mmgpu_sw_block_starts(mmgpu_ctx *, mmgpu_sw_batch_t *, uint32_t *, uint32_t *, uint32_t *) {
    return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_block_starts: not served through mmgpu_server (the drop-in hooks name their pairs: mmgpu_sw_block_backtrace)");
}

int mmgpu_sw_block_growth(mmgpu_ctx *, mmgpu_sw_batch_t *, const uint32_t *, uint32_t, mmgpu_sw_block *, uint32_t *, uint32_t) {
    return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_block_growth: a test aid, not served through mmgpu_server");
}

int mmgpu_sw_block_tiers(const mmgpu_sw_batch_t *, uint32_t *first_tier, uint32_t *second_tier) {
    if (first_tier) *first_tier = 0;
    if (second_tier) *second_tier = 0;
    return MMGPU_OK;
}

int mmgpu_pf_load_index(mmgpu_ctx *c, const mmgpu_pf_index *ix) {
    if (!c || !ix) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: NULL argument");
    if (!ix->offsets || !ix->ungapped_mat) return fail(MMGPU_ERR_ARG, "mmgpu_pf_load_index: NULL table");
    const bool three = ix->score3 && ix->index3;      // (absent: an index for exact k-mer matching)
    const size_t kalph = (size_t)ix->alphabet - 1;
    const size_t n3 = kalph * kalph * kalph, n2 = kalph * kalph;
    const size_t kbase = ix->kmer_alphabet > 0 ? (size_t)ix->kmer_alphabet : kalph;      // (the index's own alphabet: profile targets)
    size_t table = 1;
    for (int i = 0; i < ix->kmer_size; i++) table *= kbase;
    const bool two = three && ix->score2 && ix->index2;
    uint64_t fp = c->targets_fp ^ 0x696E646578212121ull;
    const int32_t scal[5] = {ix->kmer_size, ix->alphabet, ix->spaced, two ? 1 : 0, (int32_t)kbase};
    fp = fingerprint(scal, sizeof(scal), fp);
    if (three) {
        fp = fingerprint(ix->score3, n3 * ix->row3 * 2, fp);
        fp = fingerprint(ix->index3, n3 * ix->row3 * 4, fp);
    }
    if (two) {
        fp = fingerprint(ix->score2, n2 * ix->row2 * 2, fp);
        fp = fingerprint(ix->index2, n2 * ix->row2 * 4, fp);
    }
    fp = fingerprint(ix->offsets, (table + 1) * 8, fp);
    if (ix->entries6) fp = fingerprint(ix->entries6, (size_t)ix->n_entries * 6, fp);
    else {
        fp = fingerprint(ix->entry_ids, (size_t)ix->n_entries * 4, fp);
        fp = fingerprint(ix->entry_pos, (size_t)ix->n_entries * 2, fp);
    }
    fp = fingerprint(ix->ungapped_mat, (size_t)ix->alphabet * ix->alphabet, fp);
    Buf q, r;
    q.put<uint64_t>(fp);
    q.put<uint64_t>(c->mask_fp);      // the view this client's prefilter reads (0 = as loaded): the server drops another client's masked view
    int rc = call(c, OP_HAS_INDEX, q, &r);
    if (rc != MMGPU_OK) return rc;
    if (r.get<uint32_t>()) return MMGPU_OK;
    q.put<int32_t>(ix->kmer_size);
    q.put<int32_t>(ix->alphabet);
    q.put<int32_t>(ix->spaced);
    q.put<int32_t>((int32_t)kbase);
    q.put<uint64_t>((uint64_t)(three ? ix->row3 : 0));
    q.put<uint64_t>((uint64_t)(two ? ix->row2 : 0));
    q.put<uint64_t>(ix->n_entries);
    q.put_bytes(three ? ix->score3 : nullptr, three ? n3 * ix->row3 * 2 : 0);
    q.put_bytes(three ? ix->index3 : nullptr, three ? n3 * ix->row3 * 4 : 0);
    q.put_bytes(two ? ix->score2 : nullptr, two ? n2 * ix->row2 * 2 : 0);
    q.put_bytes(two ? ix->index2 : nullptr, two ? n2 * ix->row2 * 4 : 0);
    q.put_bytes(ix->offsets, (table + 1) * 8);
    q.put_bytes(ix->entries6, ix->entries6 ? (size_t)ix->n_entries * 6 : 0);
    q.put_bytes(ix->entries6 ? nullptr : ix->entry_ids, ix->entries6 ? 0 : (size_t)ix->n_entries * 4);
    q.put_bytes(ix->entries6 ? nullptr : ix->entry_pos, ix->entries6 ? 0 : (size_t)ix->n_entries * 2);
    q.put_bytes(ix->ungapped_mat, (size_t)ix->alphabet * ix->alphabet);
    return call(c, OP_LOAD_INDEX, q, nullptr);
}

// the index built on the device from the resident targets (the hook's default): resident already if this database, these
// tables and this threshold were seen before
int mmgpu_pf_build_index(mmgpu_ctx *c, const mmgpu_pf_index *ix, const int16_t *kmer_submat, int kmer_thr) {
    if (!c || !ix || !kmer_submat || !ix->ungapped_mat) return fail(MMGPU_ERR_ARG, "mmgpu_pf_build_index: NULL argument");
    const size_t kalph = (size_t)ix->alphabet - 1;
    const size_t n3 = kalph * kalph * kalph, n2 = kalph * kalph, a2 = (size_t)ix->alphabet * ix->alphabet;
    const bool three = ix->score3 && ix->index3, two = ix->score2 && ix->index2;
    uint64_t fp = c->targets_fp ^ 0x6275696C64212121ull ^ c->mask_fp;      // (an index over masked targets is another index)
    const int32_t scal[6] = {ix->kmer_size, ix->alphabet, ix->spaced, three ? 1 : 0, two ? 1 : 0, kmer_thr};
    fp = fingerprint(scal, sizeof(scal), fp);
    if (three) fp = fingerprint(ix->score3, n3 * ix->row3 * 2, fp);
    fp = fingerprint(kmer_submat, a2 * 2, fp);
    fp = fingerprint(ix->ungapped_mat, a2, fp);
    Buf q, r;
    q.put<uint64_t>(fp);
    q.put<uint64_t>(c->mask_fp);      // the view this client's prefilter reads (0 = as loaded): the server drops another client's masked view
    int rc = call(c, OP_HAS_INDEX, q, &r);
    if (rc != MMGPU_OK) return rc;
    if (r.get<uint32_t>()) return MMGPU_OK;
    q.put<int32_t>(ix->kmer_size);
    q.put<int32_t>(ix->alphabet);
    q.put<int32_t>(ix->spaced);
    q.put<int32_t>(kmer_thr);
    q.put<uint64_t>((uint64_t)(three ? ix->row3 : 0));
    q.put<uint64_t>((uint64_t)(two ? ix->row2 : 0));
    q.put_bytes(three ? ix->score3 : nullptr, three ? n3 * ix->row3 * 2 : 0);
    q.put_bytes(three ? ix->index3 : nullptr, three ? n3 * ix->row3 * 4 : 0);
    q.put_bytes(two ? ix->score2 : nullptr, two ? n2 * ix->row2 * 2 : 0);
    q.put_bytes(two ? ix->index2 : nullptr, two ? n2 * ix->row2 * 4 : 0);
    q.put_bytes(kmer_submat, a2 * 2);
    q.put_bytes(ix->ungapped_mat, a2);
    return call(c, OP_BUILD_INDEX, q, nullptr);
}

// ---- prefilter ---------------------------------------------------------------------------------------------------
int mmgpu_pf_prepare(mmgpu_ctx *c, const mmgpu_pf_params *p, const mmgpu_pf_query *qs, uint32_t nq, mmgpu_pf_batch_t **out) {
    if (!c || !p || !out || (!qs && nq)) return fail(MMGPU_ERR_ARG, "mmgpu_pf_prepare: NULL argument");
    Buf q, r;
    q.put(*p);
    q.put<uint32_t>(nq);
    for (uint32_t i = 0; i < nq; i++) {
        q.put<uint32_t>(qs[i].identity_id);
        q.put_bytes(qs[i].q, qs[i].qlen);
        q.put_bytes(qs[i].comp_bias, qs[i].comp_bias ? (size_t)qs[i].qlen * 4 : 0);
        const bool prof = qs[i].profile && qs[i].profile_score && qs[i].profile_index;
        q.put<uint32_t>(prof ? qs[i].profile_row : 0u);
        q.put_bytes(qs[i].profile_score, prof ? (size_t)qs[i].qlen * qs[i].profile_row * 2 : 0);
        q.put_bytes(qs[i].profile_index, prof ? (size_t)qs[i].qlen * qs[i].profile_row * 4 : 0);
        q.put_bytes(qs[i].profile, prof ? (size_t)20 * qs[i].qlen : 0);
    }
    const int rc = call(c, OP_PF_PREPARE, q, &r);
    if (rc != MMGPU_OK) return rc;
    mmgpu_pf_batch_t *b = new mmgpu_pf_batch_t();
    b->handle = r.get<uint64_t>();
    b->nq = nq;
    *out = b;
    return MMGPU_OK;
}

int mmgpu_pf_run(mmgpu_ctx *c, mmgpu_pf_batch_t *b) {
    if (!b) return fail(MMGPU_ERR_ARG, "mmgpu_pf_run: NULL argument");
    Buf q;
    q.put<uint64_t>(b->handle);
    return call(c, OP_PF_RUN, q, nullptr);
}

int mmgpu_pf_fetch(mmgpu_ctx *c, mmgpu_pf_batch_t *b, mmgpu_pf_hit *hits, uint32_t stride, uint32_t *counts, int32_t *status,
                   mmgpu_pf_qstat *stats) {
    if (!b || !hits || !counts || !status) return fail(MMGPU_ERR_ARG, "mmgpu_pf_fetch: NULL argument");
    Buf q, r;
    q.put<uint64_t>(b->handle);
    q.put<uint32_t>(stride);
    q.put<uint32_t>(stats ? 1u : 0u);
    const int rc = call(c, OP_PF_FETCH, q, &r);
    if (rc != MMGPU_OK) return rc;
    size_t n = 0;
    const uint8_t *p = r.get_bytes(&n);
    if (n != (size_t)b->nq * stride * sizeof(mmgpu_pf_hit)) return fail(MMGPU_ERR_STATE, "mmgpu client: malformed PF_FETCH reply");
    if (n) memcpy(hits, p, n);
    p = r.get_bytes(&n);
    if (n) memcpy(counts, p, n);
    p = r.get_bytes(&n);
    if (n) memcpy(status, p, n);
    p = r.get_bytes(&n);
    if (stats && n) memcpy(stats, p, n);
    return r.bad ? fail(MMGPU_ERR_STATE, "mmgpu client: malformed PF_FETCH reply") : MMGPU_OK;
}

void mmgpu_pf_free(mmgpu_ctx *c, mmgpu_pf_batch_t *b) {
    if (!b) return;
    Buf q;
    q.put<uint64_t>(b->handle);
    call(c, OP_PF_FREE, q, nullptr);
    delete b;
}

// ---- alignment ---------------------------------------------------------------------------------------------------
int mmgpu_sw_prepare(mmgpu_ctx *c, const mmgpu_sw_params *p, const mmgpu_sw_query *qs, uint32_t nq, int mode, mmgpu_sw_batch_t **out) {
    if (!c || !p || !out || (!qs && nq)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: NULL argument");
    Buf q, r;
    q.put<int32_t>(p->alphabet);
    q.put<int32_t>(p->gap_open);
    q.put<int32_t>(p->gap_extend);
    q.put<int32_t>(mode);
    q.put_bytes(p->mat, (size_t)p->alphabet * p->alphabet);
    q.put<uint32_t>(nq);
    for (uint32_t i = 0; i < nq; i++) {
        q.put<int32_t>(qs[i].min_start_score);
        q.put<uint32_t>(qs[i].profile ? qs[i].profile_letters : 0u);
        q.put_bytes(qs[i].q, qs[i].qlen);
        q.put_bytes(qs[i].comp_bias, qs[i].comp_bias ? qs[i].qlen : 0);
        q.put_bytes(qs[i].target_ids, (size_t)qs[i].n_targets * 4);
        q.put_bytes(qs[i].profile, qs[i].profile ? (size_t)qs[i].profile_letters * qs[i].qlen : 0);
    }
    const int rc = call(c, OP_SW_PREPARE, q, &r);
    if (rc != MMGPU_OK) return rc;
    mmgpu_sw_batch_t *b = new mmgpu_sw_batch_t();
    b->handle = r.get<uint64_t>();
    b->pairs = r.get<uint64_t>();
    *out = b;
    return MMGPU_OK;
}

int mmgpu_sw_run(mmgpu_ctx *c, mmgpu_sw_batch_t *b) {
    if (!b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_run: NULL argument");
    Buf q;
    q.put<uint64_t>(b->handle);
    return call(c, OP_SW_RUN, q, nullptr);
}

int mmgpu_sw_fetch(mmgpu_ctx *c, mmgpu_sw_batch_t *b, mmgpu_sw_hit *out) {
    if (!b || (!out && b->pairs)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_fetch: NULL argument");
    Buf q, r;
    q.put<uint64_t>(b->handle);
    const int rc = call(c, OP_SW_FETCH, q, &r);
    if (rc != MMGPU_OK) return rc;
    size_t n = 0;
    const uint8_t *p = r.get_bytes(&n);
    if (n != (size_t)b->pairs * sizeof(mmgpu_sw_hit)) return fail(MMGPU_ERR_STATE, "mmgpu client: malformed SW_FETCH reply");
    if (n) memcpy(out, p, n);
    return MMGPU_OK;
}

int mmgpu_sw_traceback(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *idx, uint32_t n, mmgpu_sw_bt *info, char *bt, size_t cap,
                       size_t *used) {
    if (!b || (!idx && n) || (!info && n)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_traceback: NULL argument");
    Buf q, r;
    q.put<uint64_t>(b->handle);
    q.put<uint64_t>(bt ? (uint64_t)cap : 0ull);
    q.put_bytes(idx, (size_t)n * 4);
    WireHdr h;
    // the sizing call of the C-ABI (bt == NULL) fails with MMGPU_ERR_ARG by design and still reports the size: the reply
    // of this op therefore carries the library's return code in its payload, not in the header
    if (!c || c->fd < 0 || !send_msg(c->fd, OP_SW_TRACEBACK, 0, q.d.data(), q.d.size()) || !recv_msg(c->fd, &h, &r))
        return fail(MMGPU_ERR_STATE, "mmgpu client: connection to mmgpu_server lost");
    if (h.status != MMGPU_OK) return fail(h.status, std::string(reinterpret_cast<const char *>(r.d.data()), r.d.size()));
    const int32_t rc = r.get<int32_t>();
    const uint64_t u = r.get<uint64_t>();
    if (used) *used = (size_t)u;
    size_t nb = 0;
    const uint8_t *p = r.get_bytes(&nb);
    if (nb && info) memcpy(info, p, nb);
    p = r.get_bytes(&nb);
    if (nb && bt && nb <= cap) memcpy(bt, p, nb);
    p = r.get_bytes(&nb);
    if (rc != MMGPU_OK) return fail(rc, std::string(reinterpret_cast<const char *>(p), nb));
    return MMGPU_OK;
}

// same conversation as mmgpu_sw_traceback: sizing call and real call both travel, the library's code rides in the payload
int mmgpu_sw_block_backtrace(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *idx, uint32_t n, mmgpu_sw_block *out, char *bt, size_t cap,
                             size_t *used) {
    if (!b || (!idx && n) || (!out && n)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_backtrace: NULL argument");
    Buf q, r;
    q.put<uint64_t>(b->handle);
    // all ones: no strings wanted; all ones but the lowest bit: start positions only
    q.put<uint64_t>(bt ? (uint64_t)cap : (cap == MMGPU_BLOCK_NO_STRINGS ? ~0ull : (cap == MMGPU_BLOCK_STARTS_ONLY ? ~1ull : 0ull)));
    q.put_bytes(idx, (size_t)n * 4);
    WireHdr h;
    if (!c || c->fd < 0 || !send_msg(c->fd, OP_SW_BLOCK_BACKTRACE, 0, q.d.data(), q.d.size()) || !recv_msg(c->fd, &h, &r))
        return fail(MMGPU_ERR_STATE, "mmgpu client: connection to mmgpu_server lost");
    if (h.status != MMGPU_OK) return fail(h.status, std::string(reinterpret_cast<const char *>(r.d.data()), r.d.size()));
    const int32_t rc = r.get<int32_t>();
    const uint64_t u = r.get<uint64_t>();
    if (used) *used = (size_t)u;
    size_t nb = 0;
    const uint8_t *p = r.get_bytes(&nb);
    if (nb && out) memcpy(out, p, nb);
    p = r.get_bytes(&nb);
    if (nb && bt && nb <= cap) memcpy(bt, p, nb);
    p = r.get_bytes(&nb);
    if (rc != MMGPU_OK) return fail(rc, std::string(reinterpret_cast<const char *>(p), nb));
    return MMGPU_OK;
}

int mmgpu_sw_reverse_pairs(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *idx, uint32_t n, mmgpu_sw_hit *out) {
    if (!b || (!idx && n)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_reverse_pairs: NULL argument");
    if (n == 0) return MMGPU_OK;
    Buf q, r;
    q.put<uint64_t>(b->handle);
    q.put_bytes(idx, (size_t)n * 4);
    const int rc = call(c, OP_SW_REVERSE_PAIRS, q, &r);
    if (rc != MMGPU_OK) return rc;
    size_t nb = 0;
    const uint8_t *p = r.get_bytes(&nb);
    if (nb != (size_t)n * sizeof(mmgpu_sw_hit)) return fail(MMGPU_ERR_STATE, "mmgpu client: malformed SW_REVERSE_PAIRS reply");
    if (out) memcpy(out, p, nb);
    return MMGPU_OK;
}

void mmgpu_sw_free(mmgpu_ctx *c, mmgpu_sw_batch_t *b) {
    if (!b) return;
    Buf q;
    q.put<uint64_t>(b->handle);
    call(c, OP_SW_FREE, q, nullptr);
    delete b;
}

// test / monitoring hook (not part of include/mmgpu.h): the server's counters
int mmgpu_client_server_stats(mmgpu_ctx *c, uint64_t out[6]) {
    Buf q, r;
    const int rc = call(c, OP_STATS, q, &r);
    if (rc != MMGPU_OK) return rc;
    for (int i = 0; i < 6; i++) out[i] = r.get<uint64_t>();
    return MMGPU_OK;
}

}  // extern "C"
