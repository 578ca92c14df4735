// Host side of the nucleotide alignment step (include/mmgpu.h: mmgpu_nucl_align): uploads the queries and pairs,
// orders the pairs longest first, sizes the per-group scratch and runs nucl_align_kernel.
#include "mmgpu_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <numeric>

using namespace mmgpu;


extern "C" int mmgpu_nucl_align(mmgpu_ctx *c, const mmgpu_nucl_params *par, const mmgpu_nucl_query *qs, uint32_t nq,
                                const mmgpu_nucl_pair *pairs, uint32_t n_pairs, mmgpu_nucl_hit *out, char *bt, uint64_t bt_cap,
                                uint64_t *bt_used) {
    if (!c || !par || !par->mat || !par->reverse || (!qs && nq) || (!pairs && n_pairs) || (!out && n_pairs))
        return fail(MMGPU_ERR_ARG, "mmgpu_nucl_align: NULL argument");
    if (!c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_nucl_align: no targets loaded");
    if (c->db.alphabet != 5) return fail(MMGPU_ERR_ARG, "mmgpu_nucl_align: the resident targets are not nucleotides (alphabet 5)");
    if (par->gap_open < 0 || par->gap_extend < 0 || par->gap_open + par->gap_extend > 60)
        return fail(MMGPU_ERR_ARG, "mmgpu_nucl_align: gap penalties out of the 8-bit range of the extension");
    if (bt_used) *bt_used = 0;
    if (n_pairs == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    const bool trace = getenv("MMGPU_TRACE") != nullptr;     // debugging aid: where the call's time goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    const auto t_begin = now();
    std::vector<uint32_t> qoff(nq + 1, 0);
    for (uint32_t i = 0; i < nq; i++) {
        if (!qs[i].q || qs[i].qlen == 0 || qs[i].qlen > 0x3FFFFFFFu) return fail(MMGPU_ERR_ARG, "mmgpu_nucl_align: bad query");
        qoff[i + 1] = qoff[i] + qs[i].qlen;
    }
    std::vector<uint8_t> qres(qoff[nq]);
    for (uint32_t i = 0; i < nq; i++) {
        for (uint32_t k = 0; k < qs[i].qlen; k++)
            if (qs[i].q[k] > 4) return fail(MMGPU_ERR_ARG, "mmgpu_nucl_align: query residue code > 4");
        memcpy(qres.data() + qoff[i], qs[i].q, qs[i].qlen);
    }
    // Anti-diagonals one extension can visit: the band (w = 64) leaves the shorter sequence after 2 * min + w of them.
    uint64_t longest = 0, most_rows = 0;
    std::vector<uint64_t> work(n_pairs);
    for (uint32_t i = 0; i < n_pairs; i++) {
        if (pairs[i].query >= nq || pairs[i].target >= c->db.n) return fail(MMGPU_ERR_ARG, "mmgpu_nucl_align: pair index out of range");
        const uint64_t ql = qs[pairs[i].query].qlen, tl = c->h_len[pairs[i].target];
        const uint64_t rows = std::min<uint64_t>(ql + tl, 2 * std::min(ql, tl) + 66);
        work[i] = rows;
        longest = std::max(longest, ql + tl);
        most_rows = std::max(most_rows, rows);
    }
    std::vector<uint32_t> order(n_pairs);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return work[a] > work[b]; });

    // one slice of direction bytes (6 blocks of 16 per anti-diagonal) and of backtrack letters per alignment in flight:
    // a wavefront per alignment (nucl_wave.h; MMGPU_NUCL_LANES=16 selects the 16-lane LDS formulation of nucl_core.h,
    // kept as a cross-check); the number in flight is what the chip holds and what fits the scratch budget (a wavefront
    // that finishes takes the next pair)
    const uint64_t p_stride = (most_rows * 6 + 2) * 16, w_stride = (longest + 16 + 15) / 16 * 16;
    static const int lanes = getenv("MMGPU_NUCL_LANES") && atoi(getenv("MMGPU_NUCL_LANES")) == 16 ? 16 : 64;
    const uint64_t gpb = 256 / lanes;                                                     // alignments per workgroup
    uint64_t groups = ((uint64_t)n_pairs + gpb - 1) / gpb * gpb;
    groups = std::min<uint64_t>(groups, (uint64_t)c->compute_units * 4 * gpb);          // 4 workgroups per CU
    const uint64_t budget = 16ull << 30;
    while (groups > gpb && groups * (p_stride + w_stride) > budget) groups -= gpb;
    if (groups * (p_stride + w_stride) > (96ull << 30)) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_nucl_align: sequences too long for the direction scratch");
    const unsigned blocks = (unsigned)(groups / gpb);

    hipStream_t s = c->stream;
    DevBuf d_pairs, d_order, d_qres, d_qoff, d_p, d_w, d_out, d_bt, d_ctr;
    // through the context's block cache: a process aligns batch after batch, the (large) scratch is allocated once
    // (MMGPU_NO_BLOCK_CACHE=1 switches the cache off)
    for (DevBuf *d : {&d_pairs, &d_order, &d_qres, &d_qoff, &d_p, &d_w, &d_out, &d_bt, &d_ctr}) d->bind(c->cache);
    std::vector<mmgpu_nucl_pair> pv(pairs, pairs + n_pairs);
    HIP_TRY(upload(d_pairs, pv, s));
    HIP_TRY(upload(d_order, order, s));
    HIP_TRY(upload(d_qres, qres, s));
    HIP_TRY(upload(d_qoff, qoff, s));
    HIP_TRY(d_p.alloc(groups * p_stride));
    HIP_TRY(d_w.alloc(groups * w_stride));
    HIP_TRY(d_out.alloc((size_t)n_pairs * sizeof(mmgpu_nucl_hit)));
    HIP_TRY(d_bt.alloc(std::max<uint64_t>(bt_cap, 16)));
    HIP_TRY(d_ctr.alloc(16));
    HIP_TRY(hipMemsetAsync(d_ctr.p, 0, 16, s));
    HIP_TRY(hipMemsetAsync(d_out.p, 0, (size_t)n_pairs * sizeof(mmgpu_nucl_hit), s));

    NuclLaunch L;
    L.pairs = d_pairs.as<mmgpu_nucl_pair>();
    L.order = d_order.as<uint32_t>();
    L.n_pairs = n_pairs;
    L.q_res = d_qres.as<uint8_t>();
    L.q_off = d_qoff.as<uint32_t>();
    L.t_res = c->db.res;
    L.t_off4 = c->db.off4;
    L.t_len = c->db.len;
    for (int i = 0; i < 25; i++) L.mat[i] = par->mat[i];
    for (int i = 0; i < 8; i++) L.rev_lookup[i] = i < 5 ? par->reverse[i] : (uint8_t)4;
    L.gapo = par->gap_open;
    L.gape = par->gap_extend;
    L.zdrop = par->zdrop;
    L.wrapped = par->wrapped ? 1 : 0;
    L.past_end_q = par->past_end_query;
    L.past_end_t = par->past_end_target;
    L.pscratch = d_p.as<uint8_t>();
    L.pscratch_stride = p_stride;
    L.wscratch = d_w.as<char>();
    L.wscratch_stride = w_stride;
    L.out = d_out.as<mmgpu_nucl_hit>();
    L.bt = d_bt.as<char>();
    L.bt_cursor = d_ctr.as<unsigned long long>();
    L.bt_cap = bt_cap;
    L.next_pair = reinterpret_cast<uint32_t *>(d_ctr.as<unsigned long long>() + 1);
    double t_setup = 0, t_kernel = 0;
    if (trace) { HIP_TRY(hipStreamSynchronize(s)); t_setup = ms_since(t_begin); }
    const auto t_k = now();
    HIP_TRY(lanes == 64 ? launch_nucl_align_wave(L, blocks, s) : launch_nucl_align(L, blocks, s));
    if (trace) { HIP_TRY(hipStreamSynchronize(s)); t_kernel = ms_since(t_k); }
    HIP_TRY(hipMemcpyAsync(out, d_out.p, (size_t)n_pairs * sizeof(mmgpu_nucl_hit), hipMemcpyDeviceToHost, s));
    unsigned long long used = 0;
    HIP_TRY(hipMemcpyAsync(&used, d_ctr.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (bt && bt_cap) HIP_TRY(hipMemcpy(bt, d_bt.p, (size_t)std::min<uint64_t>(used, bt_cap), hipMemcpyDeviceToHost));
    if (bt_used) *bt_used = used;
    if (trace)
        fprintf(stderr, "[nucl_align] %u pairs, %u workgroups x %d lanes/alignment: host prep + upload + alloc %.2f ms, kernel %.2f ms, total %.2f ms "
                "(scratch %.2f GB, strings %.1f MB)\n", n_pairs, blocks, lanes, t_setup, t_kernel, ms_since(t_begin),
                (double)(groups * (p_stride + w_stride)) / 1073741824.0, (double)used / 1048576.0);
    return MMGPU_OK;
}
