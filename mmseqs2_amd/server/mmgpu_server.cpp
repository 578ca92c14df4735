// mmgpu_server: resident server mode (reference counterpart src/util/gpuserver.cpp:23-100).
// Owns ONE mmgpu_ctx (include/mmgpu.h) with the target database and the k-mer index resident in HBM and serves the
// blocks of queries that libmmgpu_client.so forwards (mmgpu_wire.h).  Like the reference's server it runs until
// SIGINT / SIGTERM, serves one client at a time, and leaves the database resident between clients; unlike it, the database
// arrives from the first client (the patched `mmseqs` hands over its own - masked - SequenceLookup and IndexTable,
// INTEGRATION.md section 2) and is recognised by later clients through its fingerprint.
//
// usage: mmgpu_server --socket PATH [--device N] [--databases N (resident databases, default 4)]
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <new>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/mmgpu.h"
#include "mmgpu_wire.h"

using namespace mmgpu_wire;

namespace {

volatile sig_atomic_t g_run = 1;
int g_listen = -1;
void on_signal(int) {
    g_run = 0;
    if (g_listen >= 0) shutdown(g_listen, SHUT_RDWR);   // wakes accept()
}

struct PfRec {
    mmgpu_pf_batch_t *b;
    uint32_t nq;
};
struct SwRec {
    mmgpu_sw_batch_t *b;
    uint64_t pairs;
};

// One resident database = one library context (a context holds one target set + one index).  The two seams hand over
// different residues for the same database (the prefilter's SequenceLookup is masked, Alignment::run maps the plain
// sequences), so a search keeps two slots busy; slots are reused least-recently-used first.
struct Slot {
    mmgpu_ctx *ctx = nullptr;
    uint64_t targets_fp = 0, index_fp = 0, mask_fp = 0;      // mask_fp: tantan parameters the prefilter's view is masked with (0: none)
    bool have_targets = false, have_index = false;
    uint64_t last_use = 0;
};

struct Server {
    std::vector<Slot> slots;
    size_t max_slots = 4;
    int cur = 0;                 // slot of the connected client
    uint64_t clock = 0;
    mmgpu_ctx *ctx = nullptr;    // == slots[cur].ctx
    int device = 0;
    std::map<uint64_t, PfRec> pf;
    std::map<uint64_t, SwRec> sw;
    uint64_t next_handle = 1;
    ServerStats st = {0, 0, 0, 0, 0, 0};
};

bool reply(int fd, uint32_t op, int rc, const Buf &b) {
    if (rc != MMGPU_OK) {
        const char *e = mmgpu_last_error();
        return send_msg(fd, op, rc, e, strlen(e));
    }
    return send_msg(fd, op, MMGPU_OK, b.d.data(), b.d.size());
}
bool reply_err(int fd, uint32_t op, int rc, const char *msg) { return send_msg(fd, op, rc, msg, strlen(msg)); }

// batches of a client that went away without freeing them
void drop_batches(Server &S) {
    for (auto &kv : S.pf) mmgpu_pf_free(S.ctx, kv.second.b);
    for (auto &kv : S.sw) mmgpu_sw_free(S.ctx, kv.second.b);
    S.pf.clear();
    S.sw.clear();
}

// The prefilter view of the connected client's slot: `want` = 0 (residues as loaded) or the tantan parameters the client masked
// with.  A slot is found by its unmasked targets, so it may carry the masked view (and the index over it) of an earlier client.
int sync_mask_view(Server &S, uint64_t want) {
    Slot &sl = S.slots[S.cur];
    if (!sl.have_targets || sl.mask_fp == want) return MMGPU_OK;
    if (want != 0) return MMGPU_OK;      // (OP_MASK_TARGETS sets the view before an index is asked for; nothing to undo here)
    drop_batches(S);
    sl.have_index = false;
    const int rc = mmgpu_pf_mask_targets(S.ctx, nullptr, 0, 0.0, 0, nullptr);
    if (rc == MMGPU_OK) sl.mask_fp = 0;
    return rc;
}

bool handle(Server &S, int fd, const WireHdr &h, Buf &in) {
    Buf out;
    S.st.requests++;
    switch (h.op) {
        case OP_HELLO: {
            (void)in.get<int32_t>();     // the client's device wish: the server's device is the one it was started on
            int cus = 0;
            char name[256] = {0};
            const int rc = mmgpu_device_info(S.ctx, &cus, name, sizeof(name));
            out.put<int32_t>(cus);
            out.put_bytes(name, strlen(name));
            return reply(fd, h.op, rc, out);
        }
        case OP_HAS_TARGETS: {
            const uint64_t fp = in.get<uint64_t>();
            uint32_t yes = 0;
            for (size_t k = 0; k < S.slots.size(); k++)
                if (S.slots[k].have_targets && S.slots[k].targets_fp == fp) {
                    drop_batches(S);
                    S.cur = (int)k;
                    S.ctx = S.slots[k].ctx;
                    S.slots[k].last_use = ++S.clock;
                    yes = 1;
                }
            out.put<uint32_t>(yes);
            return reply(fd, h.op, MMGPU_OK, out);
        }
        case OP_LOAD_TARGETS: {
            const uint64_t fp = in.get<uint64_t>();
            const uint32_t n = in.get<uint32_t>();
            const int32_t alphabet = in.get<int32_t>();
            size_t nb = 0, nr = 0;
            const uint8_t *off = in.get_bytes(&nb);
            const uint8_t *res = in.get_bytes(&nr);
            if (in.bad || nb != ((size_t)n + 1) * 8) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed LOAD_TARGETS");
            std::vector<uint64_t> o((size_t)n + 1);
            memcpy(o.data(), off, nb);
            // the offsets must describe exactly the residues that came with them (mmgpu_load_targets reads residues[o[i] .. o[i+1]))
            bool ok = o[0] == 0 && o[n] == nr;
            for (uint32_t i = 0; i < n && ok; i++) ok = o[i] <= o[i + 1];
            if (!ok) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: LOAD_TARGETS offsets do not match the residues sent");
            drop_batches(S);
            // an empty slot, else a new one while there is room, else the least recently used
            int pick = -1;
            for (size_t k = 0; k < S.slots.size() && pick < 0; k++)
                if (!S.slots[k].have_targets) pick = (int)k;
            if (pick < 0 && S.slots.size() < S.max_slots) {
                Slot sl;
                if (mmgpu_init(&sl.ctx, S.device) != MMGPU_OK) return reply(fd, h.op, MMGPU_ERR_HIP, out);
                S.slots.push_back(sl);
                pick = (int)S.slots.size() - 1;
            }
            if (pick < 0) {
                pick = 0;
                for (size_t k = 1; k < S.slots.size(); k++)
                    if (S.slots[k].last_use < S.slots[pick].last_use) pick = (int)k;
            }
            Slot &sl = S.slots[pick];
            S.cur = pick;
            S.ctx = sl.ctx;
            sl.have_targets = sl.have_index = false;      // mmgpu_load_targets drops the index of the previous database
            sl.mask_fp = 0;
            const int rc = mmgpu_load_targets(S.ctx, res, o.data(), n, alphabet);
            if (rc == MMGPU_OK) {
                sl.have_targets = true;
                sl.targets_fp = fp;
                sl.last_use = ++S.clock;
                S.st.target_uploads++;
            }
            return reply(fd, h.op, rc, out);
        }
        case OP_MASK_TARGETS: {
            const uint64_t fp = in.get<uint64_t>();
            const int32_t alphabet = in.get<int32_t>();
            const int32_t mask_letter = in.get<int32_t>();
            size_t nb = 0, nl = 0;
            const uint8_t *pp = in.get_bytes(&nb);
            const uint8_t *lp = in.get_bytes(&nl);
            if (in.bad || nb != 8 || alphabet < 1 || alphabet > 64 || nl != (size_t)alphabet * alphabet * 8)
                return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed MASK_TARGETS");
            Slot &sl = S.slots[S.cur];
            if (!sl.have_targets) return reply_err(fd, h.op, MMGPU_ERR_STATE, "mmgpu_server: MASK_TARGETS without targets");
            uint64_t masked = 0;
            if (sl.mask_fp != fp) {      // (masked like this already: the index over it stays as well)
                double prob;
                memcpy(&prob, pp, 8);
                std::vector<double> lr((size_t)alphabet * alphabet);
                memcpy(lr.data(), lp, nl);
                drop_batches(S);
                sl.have_index = false;      // mmgpu_pf_mask_targets frees the index
                const int rc = mmgpu_pf_mask_targets(S.ctx, lr.data(), alphabet, prob, mask_letter, &masked);
                if (rc != MMGPU_OK) return reply_err(fd, h.op, rc, mmgpu_last_error());
                sl.mask_fp = fp;
            }
            out.put<uint64_t>(masked);
            return reply(fd, h.op, MMGPU_OK, out);
        }
        case OP_HAS_INDEX: {
            const uint64_t fp = in.get<uint64_t>();
            const uint64_t want_mask = in.get<uint64_t>();
            if (in.bad) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed HAS_INDEX");
            {   // a slot found by its (unmasked) targets may still carry the masked view of an earlier client: a client that runs
                // without masking (or never asked for this masking) gets the residues as loaded, and no index built over the masked ones
                const int rc = sync_mask_view(S, want_mask);
                if (rc != MMGPU_OK) return reply_err(fd, h.op, rc, mmgpu_last_error());
            }
            out.put<uint32_t>(S.slots[S.cur].have_index && fp == S.slots[S.cur].index_fp ? 1u : 0u);
            return reply(fd, h.op, MMGPU_OK, out);
        }
        case OP_LOAD_INDEX: {
            const uint64_t fp = in.get<uint64_t>();
            {
                const uint64_t want_mask = in.get<uint64_t>();
                const int rc = in.bad ? MMGPU_OK : sync_mask_view(S, want_mask);
                if (rc != MMGPU_OK) return reply_err(fd, h.op, rc, mmgpu_last_error());
            }
            mmgpu_pf_index ix;
            memset(&ix, 0, sizeof(ix));
            ix.kmer_size = in.get<int32_t>();
            ix.alphabet = in.get<int32_t>();
            ix.spaced = in.get<int32_t>();
            ix.kmer_alphabet = in.get<int32_t>();
            ix.row3 = (size_t)in.get<uint64_t>();
            ix.row2 = (size_t)in.get<uint64_t>();
            ix.n_entries = in.get<uint64_t>();
            size_t n = 0;
            // the wire payload is byte-packed: typed arrays are copied to aligned storage
            std::vector<int16_t> s3, s2;
            std::vector<uint32_t> i3, i2, ids;
            std::vector<uint64_t> offs;
            std::vector<uint16_t> pos;
            const uint8_t *p;
            p = in.get_bytes(&n); s3.resize(n / 2); if (n) memcpy(s3.data(), p, n);
            p = in.get_bytes(&n); i3.resize(n / 4); if (n) memcpy(i3.data(), p, n);
            p = in.get_bytes(&n); s2.resize(n / 2); if (n) memcpy(s2.data(), p, n);
            p = in.get_bytes(&n); i2.resize(n / 4); if (n) memcpy(i2.data(), p, n);
            p = in.get_bytes(&n); offs.resize(n / 8); if (n) memcpy(offs.data(), p, n);
            size_t n6 = 0;
            const uint8_t *e6 = in.get_bytes(&n6);
            p = in.get_bytes(&n); ids.resize(n / 4); if (n) memcpy(ids.data(), p, n);
            p = in.get_bytes(&n); pos.resize(n / 2); if (n) memcpy(pos.data(), p, n);
            size_t nm = 0;
            const uint8_t *um = in.get_bytes(&nm);
            if (in.bad) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed LOAD_INDEX");
            ix.score3 = s3.empty() ? nullptr : s3.data();
            ix.index3 = i3.empty() ? nullptr : i3.data();
            ix.score2 = s2.empty() ? nullptr : s2.data();
            ix.index2 = i2.empty() ? nullptr : i2.data();
            ix.offsets = offs.data();
            ix.entries6 = n6 ? e6 : nullptr;
            ix.entry_ids = ids.empty() ? nullptr : ids.data();
            ix.entry_pos = pos.empty() ? nullptr : pos.data();
            ix.ungapped_mat = reinterpret_cast<const int8_t *>(um);
            drop_batches(S);
            S.slots[S.cur].have_index = false;
            const int rc = mmgpu_pf_load_index(S.ctx, &ix);
            if (rc == MMGPU_OK) {
                S.slots[S.cur].have_index = true;
                S.slots[S.cur].index_fp = fp;
                S.st.index_uploads++;
            }
            return reply(fd, h.op, rc, out);
        }
        case OP_BUILD_INDEX: {
            const uint64_t fp = in.get<uint64_t>();
            {
                const uint64_t want_mask = in.get<uint64_t>();
                const int rc = in.bad ? MMGPU_OK : sync_mask_view(S, want_mask);
                if (rc != MMGPU_OK) return reply_err(fd, h.op, rc, mmgpu_last_error());
            }
            mmgpu_pf_index ix;
            memset(&ix, 0, sizeof(ix));
            ix.kmer_size = in.get<int32_t>();
            ix.alphabet = in.get<int32_t>();
            ix.spaced = in.get<int32_t>();
            const int32_t kmer_thr = in.get<int32_t>();
            ix.row3 = (size_t)in.get<uint64_t>();
            ix.row2 = (size_t)in.get<uint64_t>();
            size_t n = 0;
            std::vector<int16_t> s3, s2, km;
            std::vector<uint32_t> i3, i2;
            const uint8_t *p;
            p = in.get_bytes(&n); s3.resize(n / 2); if (n) memcpy(s3.data(), p, n);
            p = in.get_bytes(&n); i3.resize(n / 4); if (n) memcpy(i3.data(), p, n);
            p = in.get_bytes(&n); s2.resize(n / 2); if (n) memcpy(s2.data(), p, n);
            p = in.get_bytes(&n); i2.resize(n / 4); if (n) memcpy(i2.data(), p, n);
            p = in.get_bytes(&n); km.resize(n / 2); if (n) memcpy(km.data(), p, n);
            const uint8_t *um = in.get_bytes(&n);
            if (in.bad || km.empty()) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed BUILD_INDEX");
            ix.score3 = s3.empty() ? nullptr : s3.data();
            ix.index3 = i3.empty() ? nullptr : i3.data();
            ix.score2 = s2.empty() ? nullptr : s2.data();
            ix.index2 = i2.empty() ? nullptr : i2.data();
            ix.ungapped_mat = reinterpret_cast<const int8_t *>(um);
            drop_batches(S);
            S.slots[S.cur].have_index = false;
            const int rc = mmgpu_pf_build_index(S.ctx, &ix, km.data(), kmer_thr);
            if (rc == MMGPU_OK) {
                S.slots[S.cur].have_index = true;
                S.slots[S.cur].index_fp = fp;
                S.st.index_uploads++;
            }
            return reply(fd, h.op, rc, out);
        }
        case OP_PF_PREPARE: {
            const mmgpu_pf_params par = in.get<mmgpu_pf_params>();
            const uint32_t nq = in.get<uint32_t>();
            std::vector<mmgpu_pf_query> qs(nq);
            std::vector<std::vector<float> > cb(nq);
            std::vector<std::vector<int16_t> > ps(nq);
            std::vector<std::vector<uint32_t> > pi(nq);
            for (uint32_t i = 0; i < nq && !in.bad; i++) {
                memset(&qs[i], 0, sizeof(qs[i]));
                qs[i].identity_id = in.get<uint32_t>();
                size_t n = 0;
                qs[i].q = in.get_bytes(&n);
                qs[i].qlen = (uint32_t)n;
                const uint8_t *b = in.get_bytes(&n);
                cb[i].resize(n / 4);
                if (n) memcpy(cb[i].data(), b, n);
                qs[i].comp_bias = n ? cb[i].data() : nullptr;
                qs[i].profile_row = in.get<uint32_t>();
                b = in.get_bytes(&n);
                ps[i].resize(n / 2);
                if (n) memcpy(ps[i].data(), b, n);
                b = in.get_bytes(&n);
                pi[i].resize(n / 4);
                if (n) memcpy(pi[i].data(), b, n);
                qs[i].profile = reinterpret_cast<const int8_t *>(in.get_bytes(&n));
                qs[i].profile_score = ps[i].empty() ? nullptr : ps[i].data();
                qs[i].profile_index = pi[i].empty() ? nullptr : pi[i].data();
            }
            if (in.bad) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed PF_PREPARE");
            mmgpu_pf_batch_t *b = nullptr;
            const int rc = mmgpu_pf_prepare(S.ctx, &par, qs.data(), nq, &b);
            if (rc == MMGPU_OK) {
                PfRec r;
                r.b = b;
                r.nq = nq;
                S.pf[S.next_handle] = r;
                out.put<uint64_t>(S.next_handle++);
                S.st.pf_batches++;
            }
            return reply(fd, h.op, rc, out);
        }
        case OP_PF_RUN: case OP_PF_FETCH: case OP_PF_FREE: {
            const uint64_t hd = in.get<uint64_t>();
            auto it = S.pf.find(hd);
            if (it == S.pf.end()) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: unknown prefilter batch");
            if (h.op == OP_PF_RUN) return reply(fd, h.op, mmgpu_pf_run(S.ctx, it->second.b), out);
            if (h.op == OP_PF_FREE) {
                mmgpu_pf_free(S.ctx, it->second.b);
                S.pf.erase(it);
                return reply(fd, h.op, MMGPU_OK, out);
            }
            const uint32_t stride = in.get<uint32_t>();
            const uint32_t want_stats = in.get<uint32_t>();
            const uint32_t nq = it->second.nq;
            std::vector<mmgpu_pf_hit> hits((size_t)nq * stride);
            std::vector<uint32_t> counts(nq);
            std::vector<int32_t> status(nq);
            std::vector<mmgpu_pf_qstat> stats(want_stats ? nq : 0);
            const int rc = mmgpu_pf_fetch(S.ctx, it->second.b, hits.data(), stride, counts.data(), status.data(), want_stats ? stats.data() : nullptr);
            out.put_bytes(hits.data(), hits.size() * sizeof(mmgpu_pf_hit));
            out.put_bytes(counts.data(), counts.size() * 4);
            out.put_bytes(status.data(), status.size() * 4);
            out.put_bytes(stats.data(), stats.size() * sizeof(mmgpu_pf_qstat));
            return reply(fd, h.op, rc, out);
        }
        case OP_SW_PREPARE: {
            mmgpu_sw_params par;
            par.alphabet = in.get<int32_t>();
            par.gap_open = in.get<int32_t>();
            par.gap_extend = in.get<int32_t>();
            const int32_t mode = in.get<int32_t>();
            size_t n = 0;
            par.mat = reinterpret_cast<const int8_t *>(in.get_bytes(&n));
            const uint32_t nq = in.get<uint32_t>();
            std::vector<mmgpu_sw_query> qs(nq);
            std::vector<std::vector<uint32_t> > ids(nq);
            uint64_t pairs = 0;
            for (uint32_t i = 0; i < nq && !in.bad; i++) {
                memset(&qs[i], 0, sizeof(qs[i]));
                qs[i].min_start_score = in.get<int32_t>();
                qs[i].profile_letters = in.get<uint32_t>();
                qs[i].q = in.get_bytes(&n);
                qs[i].qlen = (uint32_t)n;
                qs[i].comp_bias = reinterpret_cast<const int8_t *>(in.get_bytes(&n));
                const uint8_t *t = in.get_bytes(&n);
                ids[i].resize(n / 4);
                if (n) memcpy(ids[i].data(), t, n);
                qs[i].target_ids = ids[i].data();
                qs[i].n_targets = (uint32_t)ids[i].size();
                qs[i].profile = reinterpret_cast<const int8_t *>(in.get_bytes(&n));
                pairs += qs[i].n_targets;
            }
            if (in.bad) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed SW_PREPARE");
            mmgpu_sw_batch_t *b = nullptr;
            const int rc = mmgpu_sw_prepare(S.ctx, &par, qs.data(), nq, mode, &b);
            if (rc == MMGPU_OK) {
                SwRec r;
                r.b = b;
                r.pairs = pairs;
                S.sw[S.next_handle] = r;
                out.put<uint64_t>(S.next_handle++);
                out.put<uint64_t>(pairs);
                S.st.sw_batches++;
            }
            return reply(fd, h.op, rc, out);
        }
        case OP_SW_RUN: case OP_SW_FETCH: case OP_SW_FREE: case OP_SW_TRACEBACK: case OP_SW_BLOCK_BACKTRACE: case OP_SW_REVERSE_PAIRS: {
            const uint64_t hd = in.get<uint64_t>();
            auto it = S.sw.find(hd);
            if (it == S.sw.end()) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: unknown alignment batch");
            if (h.op == OP_SW_RUN) return reply(fd, h.op, mmgpu_sw_run(S.ctx, it->second.b), out);
            if (h.op == OP_SW_FREE) {
                mmgpu_sw_free(S.ctx, it->second.b);
                S.sw.erase(it);
                return reply(fd, h.op, MMGPU_OK, out);
            }
            if (h.op == OP_SW_FETCH) {
                std::vector<mmgpu_sw_hit> res((size_t)it->second.pairs);
                const int rc = mmgpu_sw_fetch(S.ctx, it->second.b, res.data());
                out.put_bytes(res.data(), res.size() * sizeof(mmgpu_sw_hit));
                return reply(fd, h.op, rc, out);
            }
            if (h.op == OP_SW_REVERSE_PAIRS) {
                size_t nr = 0;
                const uint8_t *rp = in.get_bytes(&nr);
                if (in.bad || nr % 4) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed SW_REVERSE_PAIRS");
                std::vector<uint32_t> ridx(nr / 4);
                if (nr) memcpy(ridx.data(), rp, nr);
                std::vector<mmgpu_sw_hit> res(ridx.size());
                const int rc = mmgpu_sw_reverse_pairs(S.ctx, it->second.b, ridx.data(), (uint32_t)ridx.size(), res.data());
                out.put_bytes(res.data(), res.size() * sizeof(mmgpu_sw_hit));
                return reply(fd, h.op, rc, out);
            }
            const uint64_t cap = in.get<uint64_t>();
            size_t n = 0;
            const uint8_t *ip = in.get_bytes(&n);
            std::vector<uint32_t> idx(n / 4);
            if (n) memcpy(idx.data(), ip, n);
            const bool starts_only = h.op == OP_SW_BLOCK_BACKTRACE && cap == ~1ull;      // MMGPU_BLOCK_STARTS_ONLY on the client side
            const bool no_strings = starts_only || (h.op == OP_SW_BLOCK_BACKTRACE && cap == ~0ull);      // MMGPU_BLOCK_NO_STRINGS
            if (in.bad || (!no_strings && cap > MMGPU_WIRE_MAX_MSG)) return reply_err(fd, h.op, MMGPU_ERR_ARG, "mmgpu_server: malformed SW_TRACEBACK");
            if (h.op == OP_SW_BLOCK_BACKTRACE) {
                std::vector<mmgpu_sw_block> blk(idx.size());
                std::vector<char> bts(no_strings ? 0 : (size_t)cap);
                size_t used_b = 0;
                const int rcb = no_strings ? mmgpu_sw_block_backtrace(S.ctx, it->second.b, idx.data(), (uint32_t)idx.size(), blk.data(), nullptr,
                                                                      starts_only ? MMGPU_BLOCK_STARTS_ONLY : MMGPU_BLOCK_NO_STRINGS, &used_b)
                                           : mmgpu_sw_block_backtrace(S.ctx, it->second.b, idx.data(), (uint32_t)idx.size(), blk.data(),
                                                                      cap ? bts.data() : nullptr, (size_t)cap, &used_b);
                out.put<int32_t>(rcb);
                out.put<uint64_t>((uint64_t)used_b);
                out.put_bytes(blk.data(), blk.size() * sizeof(mmgpu_sw_block));
                out.put_bytes(bts.data(), (rcb == MMGPU_OK && !no_strings) ? std::min<size_t>(used_b, (size_t)cap) : 0);
                const char *eb = rcb == MMGPU_OK ? "" : mmgpu_last_error();
                out.put_bytes(eb, strlen(eb));
                return reply(fd, h.op, MMGPU_OK, out);
            }
            std::vector<mmgpu_sw_bt> info(idx.size());
            std::vector<char> bt((size_t)cap);
            size_t used = 0;
            const int rc = mmgpu_sw_traceback(S.ctx, it->second.b, idx.data(), (uint32_t)idx.size(), info.data(), cap ? bt.data() : nullptr,
                                              (size_t)cap, &used);
            // the sizing call (cap == 0) returns an error code by design: the code travels in the payload
            out.put<int32_t>(rc);
            out.put<uint64_t>((uint64_t)used);
            out.put_bytes(info.data(), info.size() * sizeof(mmgpu_sw_bt));
            out.put_bytes(bt.data(), rc == MMGPU_OK ? std::min<size_t>(used, (size_t)cap) : 0);
            const char *e = rc == MMGPU_OK ? "" : mmgpu_last_error();
            out.put_bytes(e, strlen(e));
            return reply(fd, h.op, MMGPU_OK, out);
        }
        case OP_STATS: {
            out.put(S.st.requests);
            out.put(S.st.clients);
            out.put(S.st.target_uploads);
            out.put(S.st.index_uploads);
            out.put(S.st.pf_batches);
            out.put(S.st.sw_batches);
            return reply(fd, h.op, MMGPU_OK, out);
        }
        case OP_SHUTDOWN:
            g_run = 0;
            return reply(fd, h.op, MMGPU_OK, out);
        default: break;
    }
    return reply_err(fd, h.op, MMGPU_ERR_UNSUPPORTED, "mmgpu_server: unknown request");
}

}  // namespace

int main(int argc, char **argv) {
    std::string path;
    Server S;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--socket") && i + 1 < argc) path = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) S.device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--databases") && i + 1 < argc) S.max_slots = (size_t)std::max(1, atoi(argv[++i]));
        else {
            fprintf(stderr, "usage: mmgpu_server --socket PATH [--device N] [--databases N]\n");
            return 2;
        }
    }
    if (path.empty()) {
        fprintf(stderr, "usage: mmgpu_server --socket PATH [--device N] [--databases N]\n");
        return 2;
    }
    {
        Slot first;
        if (mmgpu_init(&first.ctx, S.device) != MMGPU_OK) {     // no device, no server: there is nothing to fall back to
            fprintf(stderr, "mmgpu_server: %s\n", mmgpu_last_error());
            return 1;
        }
        S.slots.push_back(first);
        S.ctx = first.ctx;
    }
    sockaddr_un a;
    memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    if (path.size() >= sizeof(a.sun_path)) {
        fprintf(stderr, "mmgpu_server: socket path too long\n");
        return 2;
    }
    strcpy(a.sun_path, path.c_str());
    unlink(path.c_str());
    g_listen = socket(AF_UNIX, SOCK_STREAM, 0);
    umask(077);      // the socket is created private (chmod below only narrows what an odd umask left)
    if (g_listen < 0 || bind(g_listen, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || listen(g_listen, 16) != 0) {
        fprintf(stderr, "mmgpu_server: cannot listen on %s: %s\n", path.c_str(), strerror(errno));
        return 1;
    }
    chmod(path.c_str(), 0600);
    struct sigaction act;
    memset(&act, 0, sizeof(act));
    act.sa_handler = on_signal;
    sigaction(SIGINT, &act, NULL);
    sigaction(SIGTERM, &act, NULL);
    signal(SIGPIPE, SIG_IGN);
    int cus = 0;
    char name[256] = {0};
    mmgpu_device_info(S.ctx, &cus, name, sizeof(name));
    fprintf(stderr, "mmgpu_server: device %d (%s, %d CUs) listening on %s\n", S.device, name, cus, path.c_str());
    fflush(stderr);
    while (g_run) {
        const int fd = accept(g_listen, NULL, NULL);     // one client at a time (the reference's RESERVED state)
        if (fd < 0) {
            if (errno == EINTR && g_run) continue;
            break;
        }
        S.st.clients++;
        WireHdr h;
        Buf in;
        try {
            while (g_run && recv_msg(fd, &h, &in)) {
                if (!handle(S, fd, h, in)) break;
            }
        } catch (const std::bad_alloc &) {      // a request too large for this host: this client loses its connection, the server stays
            fprintf(stderr, "mmgpu_server: out of host memory while serving a request; connection closed\n");
        }
        close(fd);
        drop_batches(S);       // the database stays resident, the client's batches do not
    }
    drop_batches(S);
    for (Slot &sl : S.slots) mmgpu_destroy(sl.ctx);
    close(g_listen);
    unlink(path.c_str());
    fprintf(stderr, "mmgpu_server: %llu requests from %llu clients, %llu target uploads, %llu index uploads\n",
            (unsigned long long)S.st.requests, (unsigned long long)S.st.clients, (unsigned long long)S.st.target_uploads,
            (unsigned long long)S.st.index_uploads);
    return 0;
}
