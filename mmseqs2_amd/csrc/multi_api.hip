// One process, several GPUs: the contexts that hold the shards of one target database, their communicator, and the
// prefilter -> exchange -> merge -> owned alignment -> gather sequence driven over all of them from one host thread
// ("mmgpu_ctx: one per process, owns N devices + RCCL comm"; section 8e).  This is the form the patched
// `mmseqs` binary uses (integration/): the reference's own multi-node analogue is Prefiltering::runMpiSplits +
// mergeTargetSplits (Prefiltering.cpp:605-689,412-526), which exchanges result files; here the shards exchange 16-byte
// records over xGMI and the merged lists equal the UNSPLIT run.
//
// Every step is a loop over the contexts that only enqueues work on each context's stream; the collectives of a step are
// issued for all contexts inside one RCCL group (ncclGroupStart / ncclGroupEnd), which is how one thread drives several
// communicator ranks.  Transport "copy" (device-to-device copies ordered by events, hipMemcpyPeerAsync between devices)
// stands in where RCCL cannot be used: several contexts on ONE device (the shard logic is tested that way on a 1-GPU box),
// or MMGPU_MULTI_TRANSPORT=copy.
#include <cstring>

#include "mmgpu_internal.h"

using mmgpu::fail;
using mmgpu::XchgBlock;

struct mmgpu_multi {
    std::vector<mmgpu_ctx *> ctx;
    std::vector<hipEvent_t> ready;      // per context: "this step's send buffers are complete"
    bool copy_transport = false;
    // shard description of the resident database
    std::vector<uint32_t> shard_of, local_id, shard_sizes;
    std::vector<std::vector<uint32_t>> global_ids;
    uint32_t n_global = 0;
    // the WHOLE database once more, in a context of its own on the first device (when it fits beside the shard): queries whose merged
    // list comes back flagged inexact run against it (mmgpu_multi_pf_run), so that no query is handed back for the way it was sharded
    mmgpu_ctx *full = nullptr;
};

// what mmgpu_multi_pf_run needs of the queries to run some of them again (the caller's arrays may be gone by then)
struct MultiQueryCopy {
    std::vector<uint8_t> q;
    std::vector<float> comp_bias;
    std::vector<int16_t> profile_score;
    std::vector<uint32_t> profile_index;
    std::vector<int8_t> profile;
};

struct mmgpu_multi_pf_batch {
    std::vector<mmgpu_pf_batch_t *> b;
    uint32_t nq = 0, stride = 0;
    std::vector<uint32_t> identity_global;
    bool has_identity = false;
    mmgpu_pf_params par;
    std::vector<MultiQueryCopy> copies;
    std::vector<mmgpu_pf_query> queries;      // global identity ids, pointing into `copies`
    uint32_t n_redone = 0, n_left = 0;        // of the last run
    bool redo_pending = false;
};

namespace {

int all_gather_step(mmgpu_multi *m, const std::vector<std::array<XchgBlock, 2>> &blk) {
    const int n = (int)m->ctx.size();
    if (!m->copy_transport && n > 1) {
        if (int e = mmgpu::comm_group_start()) return e;
        int err = MMGPU_OK;
        for (int i = 0; i < n && !err; i++) {
            (void)hipSetDevice(m->ctx[i]->device);
            for (int k = 0; k < 2 && !err; k++) err = mmgpu::comm_allgather(m->ctx[i], blk[i][k].send, blk[i][k].recv, blk[i][k].bytes);
        }
        const int e2 = mmgpu::comm_group_end();
        return err ? err : e2;
    }
    if (n == 1) {
        for (int k = 0; k < 2; k++)
            if (int e = mmgpu::comm_allgather(m->ctx[0], blk[0][k].send, blk[0][k].recv, blk[0][k].bytes)) return e;
        return MMGPU_OK;
    }
    // copy transport: context i pulls every rank's block once that rank's stream has produced it
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->ctx[i]->device));
        HIP_TRY(hipEventRecord(m->ready[i], m->ctx[i]->stream));
    }
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->ctx[i]->device));
        for (int j = 0; j < n; j++) {
            if (j != i) HIP_TRY(hipStreamWaitEvent(m->ctx[i]->stream, m->ready[j], 0));
            for (int k = 0; k < 2; k++) {
                if (blk[j][k].bytes == 0) continue;
                char *dst = (char *)blk[i][k].recv + (size_t)j * blk[j][k].bytes;
                if (m->ctx[i]->device == m->ctx[j]->device)
                    HIP_TRY(hipMemcpyAsync(dst, blk[j][k].send, blk[j][k].bytes, hipMemcpyDeviceToDevice, m->ctx[i]->stream));
                else
                    HIP_TRY(hipMemcpyPeerAsync(dst, m->ctx[i]->device, blk[j][k].send, m->ctx[j]->device, blk[j][k].bytes, m->ctx[i]->stream));
            }
        }
    }
    // a context must not start its next step (and overwrite its send buffers) before everyone has pulled them
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->ctx[i]->device));
        HIP_TRY(hipEventRecord(m->ready[i], m->ctx[i]->stream));
    }
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->ctx[i]->device));
        for (int j = 0; j < n; j++)
            if (j != i) HIP_TRY(hipStreamWaitEvent(m->ctx[i]->stream, m->ready[j], 0));
    }
    return MMGPU_OK;
}

}  // namespace

extern "C" int mmgpu_init_multi(mmgpu_multi **out, const int *device_ids, int n_devices) {
    if (!out || !device_ids) return fail(MMGPU_ERR_ARG, "mmgpu_init_multi: NULL argument");
    if (n_devices < 1 || n_devices > 64) return fail(MMGPU_ERR_ARG, "mmgpu_init_multi: 1..64 devices");
    mmgpu_multi *m = new mmgpu_multi();
    bool repeats = false;
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++) repeats |= device_ids[i] == device_ids[j];
    m->copy_transport = repeats;
    for (int i = 0; i < n_devices; i++) {
        mmgpu_ctx *c = nullptr;
        if (int e = mmgpu_init(&c, device_ids[i])) { mmgpu_destroy_multi(m); return e; }
        m->ctx.push_back(c);
        // each context works on its own stream, so that the devices (or the contexts sharing one) run side by side
        hipStream_t s = nullptr;
        hipEvent_t ev = nullptr;
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            mmgpu_destroy_multi(m);
            return fail(MMGPU_ERR_HIP, "mmgpu_init_multi: stream / event creation failed");
        }
        c->stream = s;
        c->owns_stream = true;
        m->ready.push_back(ev);
    }
    if (m->copy_transport) {
        for (int i = 0; i < n_devices; i++) {
            mmgpu::Comm *cm = new mmgpu::Comm();
            cm->rank = i;
            cm->n_ranks = n_devices;
            cm->transport = "copy";
            m->ctx[i]->comm = cm;
            for (int j = 0; j < n_devices; j++)       // peer access for the pulls; "already enabled" is fine
                if (device_ids[i] != device_ids[j]) { (void)hipSetDevice(device_ids[i]); (void)hipDeviceEnablePeerAccess(device_ids[j], 0); (void)hipGetLastError(); }
        }
    } else if (n_devices > 1) {
        if (int e = mmgpu::comm_init_all(m->ctx.data(), n_devices)) { mmgpu_destroy_multi(m); return e; }
    }
    *out = m;
    return MMGPU_OK;
}

extern "C" void mmgpu_destroy_multi(mmgpu_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        (void)hipSetDevice(m->ctx[i]->device);
        (void)hipDeviceSynchronize();
        if (i < m->ready.size() && m->ready[i]) (void)hipEventDestroy(m->ready[i]);
        mmgpu_destroy(m->ctx[i]);
    }
    if (m->full) mmgpu_destroy(m->full);
    delete m;
}

extern "C" int mmgpu_multi_size(mmgpu_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" mmgpu_ctx *mmgpu_multi_ctx(mmgpu_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

extern "C" int mmgpu_multi_synchronize(mmgpu_multi *m) {
    if (!m) return fail(MMGPU_ERR_ARG, "mmgpu_multi_synchronize: NULL argument");
    for (mmgpu_ctx *c : m->ctx)
        if (int e = mmgpu_synchronize(c)) return e;
    return MMGPU_OK;
}

// The database dealt to the contexts by length bucket (mmgpu_host_partition_targets), shard i resident on context i with its
// shard description set (global ids kept: every shard answers for the whole database).
extern "C" int mmgpu_multi_load_targets(mmgpu_multi *m, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int alphabet) {
    if (!m || !offsets || (!residues && n)) return fail(MMGPU_ERR_ARG, "mmgpu_multi_load_targets: NULL argument");
    const uint32_t ns = (uint32_t)m->ctx.size();
    m->shard_of.assign(std::max<uint32_t>(n, 1), 0);
    m->local_id.assign(std::max<uint32_t>(n, 1), 0);
    m->shard_sizes.assign(ns, 0);
    if (int e = mmgpu_host_partition_targets(offsets, n, ns, m->shard_of.data(), m->local_id.data(), m->shard_sizes.data(), nullptr)) return e;
    m->n_global = n;
    m->global_ids.assign(ns, std::vector<uint32_t>());
    std::vector<std::vector<uint64_t>> soff(ns);
    std::vector<std::vector<uint8_t>> sres(ns);
    for (uint32_t s = 0; s < ns; s++) {
        m->global_ids[s].reserve(m->shard_sizes[s]);
        soff[s].reserve((size_t)m->shard_sizes[s] + 1);
        soff[s].push_back(0);
    }
    std::vector<uint64_t> bytes(ns, 0);
    for (uint32_t i = 0; i < n; i++) bytes[m->shard_of[i]] += offsets[i + 1] - offsets[i];
    for (uint32_t s = 0; s < ns; s++) sres[s].reserve(bytes[s]);
    for (uint32_t i = 0; i < n; i++) {          // ascending global id inside a shard = ascending local id
        const uint32_t s = m->shard_of[i];
        m->global_ids[s].push_back(i);
        sres[s].insert(sres[s].end(), residues + offsets[i], residues + offsets[i + 1]);
        soff[s].push_back(sres[s].size());
    }
    for (uint32_t s = 0; s < ns; s++) {
        if (int e = mmgpu_load_targets(m->ctx[s], sres[s].empty() ? residues : sres[s].data(), soff[s].data(), m->shard_sizes[s], alphabet)) return e;
        mmgpu_pf_shard sh;
        sh.n_shards = ns;
        sh.shard = s;
        sh.global_db_size = n;
        sh.global_ids = m->global_ids[s].data();
        sh.shard_of = m->shard_of.data();
        sh.local_id = m->local_id.data();
        if (int e = mmgpu_pf_set_shard(m->ctx[s], &sh)) return e;
    }
    // the unsplit copy: residues twice (masked view), 8-byte index entries, offset and score tables - when the first device has
    // the room beside its shard (a database that is sharded because it does NOT fit one device keeps handing such queries back)
    if (m->full) { mmgpu_destroy(m->full); m->full = nullptr; }
    if (ns > 1 && n > 0) {
        size_t free_b = 0, total_b = 0;
        (void)hipSetDevice(m->ctx[0]->device);
        const uint64_t need = 12ull * offsets[n] + (3ull << 30);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)free_b > 2 * need) {
            if (int e = mmgpu_init(&m->full, m->ctx[0]->device)) return e;
            if (int e = mmgpu_load_targets(m->full, residues, offsets, n, alphabet)) return e;
        }
    }
    return MMGPU_OK;
}

// tantan masking shard by shard (mmgpu_pf_mask_targets): the model runs over one sequence at a time, so the masked shards
// together are the masked database
extern "C" int mmgpu_multi_pf_mask_targets(mmgpu_multi *m, const double *likelihood_ratios, int alphabet, double min_mask_prob,
                                           int mask_letter, uint64_t *n_masked) {
    if (!m) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_mask_targets: NULL argument");
    uint64_t total = 0;
    for (mmgpu_ctx *c : m->ctx) {
        uint64_t n = 0;
        if (int e = mmgpu_pf_mask_targets(c, likelihood_ratios, alphabet, min_mask_prob, mask_letter, &n)) return e;
        total += n;
    }
    if (m->full)
        if (int e = mmgpu_pf_mask_targets(m->full, likelihood_ratios, alphabet, min_mask_prob, mask_letter, nullptr)) return e;
    if (n_masked) *n_masked = total;
    return MMGPU_OK;
}

// IndexBuilder::fillDatabase of every shard on its own device (mmgpu_pf_build_index)
extern "C" int mmgpu_multi_pf_build_index(mmgpu_multi *m, const mmgpu_pf_index *ix, const int16_t *kmer_submat, int kmer_thr) {
    if (!m) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_build_index: NULL argument");
    for (mmgpu_ctx *c : m->ctx)
        if (int e = mmgpu_pf_build_index(c, ix, kmer_submat, kmer_thr)) return e;
    if (m->full)
        if (int e = mmgpu_pf_build_index(m->full, ix, kmer_submat, kmer_thr)) return e;
    return MMGPU_OK;
}

// queries[i].identity_id is the GLOBAL id of the query's own target (UINT32_MAX none): every shard gets its local id
extern "C" int mmgpu_multi_pf_prepare(mmgpu_multi *m, const mmgpu_pf_params *par, const mmgpu_pf_query *qs, uint32_t nq, mmgpu_multi_pf_batch **out) {
    if (!m || !par || !out || (!qs && nq)) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_prepare: NULL argument");
    mmgpu_multi_pf_batch *mb = new mmgpu_multi_pf_batch();
    mb->nq = nq;
    mb->identity_global.assign(nq, 0xFFFFFFFFu);
    std::vector<mmgpu_pf_query> local(qs, qs + nq);
    for (uint32_t i = 0; i < nq; i++)
        if (qs[i].identity_id != 0xFFFFFFFFu) {
            if (qs[i].identity_id >= m->n_global) { delete mb; return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_prepare: identity_id beyond the database"); }
            mb->identity_global[i] = qs[i].identity_id;
            mb->has_identity = true;
        }
    for (size_t s = 0; s < m->ctx.size(); s++) {
        for (uint32_t i = 0; i < nq; i++) {
            const uint32_t g = mb->identity_global[i];
            local[i].identity_id = (g != 0xFFFFFFFFu && m->shard_of[g] == s) ? m->local_id[g] : 0xFFFFFFFFu;
        }
        mmgpu_pf_batch_t *b = nullptr;
        if (int e = mmgpu_pf_prepare(m->ctx[s], par, local.data(), nq, &b)) { mmgpu_multi_pf_free(m, mb); return e; }
        mb->b.push_back(b);
    }
    mb->par = *par;
    if (m->full) {      // (deep copies: mmgpu_pf_prepare copied what the shards need, a re-run needs the queries again)
        mb->copies.resize(nq);
        mb->queries.assign(qs, qs + nq);
        for (uint32_t i = 0; i < nq; i++) {
            MultiQueryCopy &cp = mb->copies[i];
            mmgpu_pf_query &d = mb->queries[i];
            cp.q.assign(qs[i].q, qs[i].q + qs[i].qlen);
            d.q = cp.q.data();
            if (qs[i].comp_bias) { cp.comp_bias.assign(qs[i].comp_bias, qs[i].comp_bias + qs[i].qlen); d.comp_bias = cp.comp_bias.data(); }
            if (qs[i].profile_score && qs[i].profile_index && qs[i].profile) {
                const size_t rows = (size_t)qs[i].qlen * qs[i].profile_row;
                cp.profile_score.assign(qs[i].profile_score, qs[i].profile_score + rows);
                cp.profile_index.assign(qs[i].profile_index, qs[i].profile_index + rows);
                cp.profile.assign(qs[i].profile, qs[i].profile + (size_t)20 * qs[i].qlen);
                d.profile_score = cp.profile_score.data();
                d.profile_index = cp.profile_index.data();
                d.profile = cp.profile.data();
            }
        }
    }
    *out = mb;
    return MMGPU_OK;
}

// prefilter of every shard, all-gather of the exchange records, merge on every context: nothing but enqueues
extern "C" int mmgpu_multi_pf_run(mmgpu_multi *m, mmgpu_multi_pf_batch *mb) {
    if (!m || !mb || mb->b.size() != m->ctx.size()) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_run: bad argument");
    const int n = (int)m->ctx.size();
    for (int i = 0; i < n; i++)
        if (int e = mmgpu_pf_run(m->ctx[i], mb->b[i])) return e;
    std::vector<std::array<XchgBlock, 2>> blk(n);
    for (int i = 0; i < n; i++)
        if (int e = mmgpu::pf_xchg_begin(m->ctx[i], mb->b[i], n, blk[i].data())) return e;
    if (int e = all_gather_step(m, blk)) return e;
    for (int i = 0; i < n; i++)
        if (int e = mmgpu::pf_xchg_merge(m->ctx[i], mb->b[i], n, mb->has_identity ? mb->identity_global.data() : nullptr)) return e;
    mb->n_redone = mb->n_left = 0;
    mb->redo_pending = m->full != nullptr && mb->nq != 0;
    return MMGPU_OK;
}

// Queries whose merged list is flagged (every context computed the same flags): once more against the whole database, the rows
// applied to every context's merged lists - the alignment step reads them there.  Runs where the batch's lists are first read
// (mmgpu_multi_pf_fetch, mmgpu_multi_sw_from_pf): it reads the flags back, and mmgpu_multi_pf_run only enqueues.
static int multi_redo_unsplit(mmgpu_multi *m, mmgpu_multi_pf_batch *mb) {
    if (!mb->redo_pending) return MMGPU_OK;
    mb->redo_pending = false;
    std::vector<uint32_t> flagged;
    if (int e = mmgpu::pf_redo_flagged(m->ctx[0], mb->b[0], flagged)) return e;
    if (flagged.empty()) return MMGPU_OK;
    mmgpu::PfRedoRows rows;
    if (int e = mmgpu::pf_redo_run(m->full, &mb->par, mb->queries.data(), flagged, rows)) return e;
    for (size_t i = 0; i < m->ctx.size(); i++)
        if (int e = mmgpu::pf_redo_apply(m->ctx[i], mb->b[i], rows)) return e;
    mb->n_redone = (uint32_t)flagged.size();
    for (int32_t st : rows.status) mb->n_left += st != MMGPU_PF_OK;
    return MMGPU_OK;
}

extern "C" int mmgpu_multi_pf_redone(mmgpu_multi_pf_batch *mb, uint32_t *n_redone, uint32_t *n_left) {
    if (!mb) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_redone: NULL batch");
    if (n_redone) *n_redone = mb->n_redone;
    if (n_left) *n_left = mb->n_left;
    return MMGPU_OK;
}

extern "C" int mmgpu_multi_has_unsplit(mmgpu_multi *m) { return m && m->full ? 1 : 0; }

// merged lists (global ids, the unsplit run's order) from context 0; status[q] = MMGPU_PF_X_INEXACT (4) where a shard's
// element took the reference's overflow path or scores were not computed on the device: re-run such queries unsplit
extern "C" int mmgpu_multi_pf_fetch(mmgpu_multi *m, mmgpu_multi_pf_batch *mb, mmgpu_pf_hit *hits, uint32_t hit_stride, uint32_t *counts, int32_t *status) {
    if (!m || !mb || mb->b.empty() || ((!hits || !counts) && mb->nq)) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_fetch: NULL argument");
    const mmgpu_pf_hit *dh = nullptr;
    const uint32_t *dc = nullptr;
    uint32_t stride = 0, nq = 0;
    if (!mmgpu::pf_batch_merged_lists(mb->b[0], &dh, &dc, &stride, &nq)) return fail(MMGPU_ERR_STATE, "mmgpu_multi_pf_fetch: batch was never run");
    if (hit_stride < stride) return fail(MMGPU_ERR_ARG, "mmgpu_multi_pf_fetch: hit_stride smaller than min(max_hits, dbSize)");
    if (nq == 0) return MMGPU_OK;
    if (int e = multi_redo_unsplit(m, mb)) return e;
    mmgpu_ctx *c = m->ctx[0];
    HIP_TRY(hipSetDevice(c->device));
    const void *df = nullptr;
    if (int e = mmgpu::pf_batch_merged_flags(mb->b[0], &df)) return e;
    std::vector<uint32_t> flags(nq);
    HIP_TRY(hipMemcpy2DAsync(hits, (size_t)hit_stride * sizeof(mmgpu_pf_hit), dh, (size_t)stride * sizeof(mmgpu_pf_hit), (size_t)stride * sizeof(mmgpu_pf_hit),
                             nq, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(counts, dc, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(flags.data(), df, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t q = 0; q < nq; q++) {
        const bool inexact = (flags[q] & 1u) != 0;
        int32_t st = inexact ? MMGPU_PF_SHARD_INEXACT : MMGPU_PF_OK;
        // what a shard's mmgpu_pf_run decided on the host never reaches the merged flag word: a query of 32768 residues or more
        // (MMGPU_PF_LONG_SEQ) or with more list segments than the device handles (MMGPU_PF_OVERFLOW) contributed nothing on that
        // shard - the merged list would silently miss its hits.  The worst status of any shard is the query's.
        for (size_t s = 0; s < mb->b.size(); s++) {
            const int32_t *hs = mmgpu::pf_batch_host_status(mb->b[s]);
            if (hs && hs[q] != MMGPU_PF_OK && (st == MMGPU_PF_OK || st == MMGPU_PF_SHARD_INEXACT)) st = hs[q];
        }
        // ... unless the query ran once more against the whole database: that run's word counts
        const int32_t *rs = mmgpu::pf_batch_redo_status(mb->b[0]);
        if (rs && rs[q] >= 0) st = rs[q];
        if (status) status[q] = st;
        if (st != MMGPU_PF_OK) counts[q] = 0;
    }
    return MMGPU_OK;
}

extern "C" uint32_t mmgpu_multi_pf_stride(mmgpu_multi_pf_batch *mb) {
    const mmgpu_pf_hit *dh = nullptr;
    const uint32_t *dc = nullptr;
    uint32_t stride = 0, nq = 0;
    if (!mb || mb->b.empty() || !mmgpu::pf_batch_merged_lists(mb->b[0], &dh, &dc, &stride, &nq)) return 0;
    return stride;
}

extern "C" void mmgpu_multi_pf_free(mmgpu_multi *m, mmgpu_multi_pf_batch *mb) {
    if (!m || !mb) return;
    for (size_t i = 0; i < mb->b.size(); i++) mmgpu_pf_free(m->ctx[i], mb->b[i]);
    delete mb;
}

// Alignment of the merged lists of a batch that has been run: every context aligns the pairs whose target its shard holds,
// the records are gathered over the communicator, `out` [nq * stride] receives them in merged-list order from context 0.
// queries as for mmgpu_sw_prepare_from_pf.  kernel_ms (may be NULL): slowest context's alignment kernels.
extern "C" int mmgpu_multi_sw_from_pf(mmgpu_multi *m, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq, int mode,
                                      mmgpu_multi_pf_batch *mb, mmgpu_sw_hit *out, uint64_t *cells, float *kernel_ms) {
    if (!m || !mb || mb->b.size() != m->ctx.size()) return fail(MMGPU_ERR_ARG, "mmgpu_multi_sw_from_pf: bad argument");
    const int n = (int)m->ctx.size();
    if (int e = multi_redo_unsplit(m, mb)) return e;
    std::vector<mmgpu_sw_batch_t *> sb(n, nullptr);
    int err = MMGPU_OK;
    for (int i = 0; i < n && !err; i++) err = mmgpu_sw_prepare_owned(m->ctx[i], par, qs, nq, mode, mb->b[i], &sb[i]);
    for (int i = 0; i < n && !err; i++) err = mmgpu_sw_run(m->ctx[i], sb[i]);
    for (int attempt = 0; attempt < 2 && !err; attempt++) {
        std::vector<std::array<XchgBlock, 2>> blk(n);
        for (int i = 0; i < n && !err; i++) err = mmgpu::sw_gather_begin(m->ctx[i], sb[i], n, blk[i].data());
        if (!err) err = all_gather_step(m, blk);
        for (int i = 0; i < n && !err; i++) err = mmgpu::sw_gather_finish(m->ctx[i], sb[i], n);
        // a shard that owns more of the hits than its send buffer holds: once more with buffers for every slot
        bool again = false;
        for (int i = 0; i < n && !err && attempt == 0 && n > 1; i++) {
            bool a = false;
            err = mmgpu::sw_gather_overflowed(m->ctx[i], sb[i], &a);
            again |= a;
        }
        if (!again) break;
    }
    uint32_t records = 0;
    if (!err) err = mmgpu_sw_fetch_owned(m->ctx[0], sb[0], out, &records);
    if (!err && (cells || kernel_ms)) {
        uint64_t tot = 0;
        float worst = 0.0f;
        for (int i = 0; i < n && !err; i++) {
            uint64_t c1 = 0, p1 = 0;
            float ms = 0.0f;
            err = mmgpu_sw_batch_stats(sb[i], &c1, &p1);
            if (!err) err = mmgpu_sw_last_kernel_ms(m->ctx[i], sb[i], &ms);
            tot += c1;
            worst = std::max(worst, ms);
        }
        if (cells) *cells = tot;
        if (kernel_ms) *kernel_ms = worst;
    }
    std::string keep = mmgpu::g_last_error;
    for (int i = 0; i < n; i++) {
        if (!sb[i]) continue;
        (void)mmgpu_synchronize(m->ctx[i]);
        mmgpu_sw_free(m->ctx[i], sb[i]);
    }
    if (err) mmgpu::g_last_error = keep;
    return err;
}
