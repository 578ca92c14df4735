// Host side of libmmgpu: context, target database residency, batch scheduling, launches.
// Everything here is plumbing around the kernels in sw_kernel.hip; see include/mmgpu.h for the contract.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "mmgpu_internal.h"

#include <chrono>

using namespace mmgpu;

namespace mmgpu {
thread_local std::string g_last_error;
}
using mmgpu::g_last_error;

extern "C" const char *mmgpu_last_error(void) { return g_last_error.c_str(); }

extern "C" int mmgpu_init(mmgpu_ctx **out, int device_id) {
    if (!out) return fail(MMGPU_ERR_ARG, "mmgpu_init: ctx is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(MMGPU_ERR_ARG, "mmgpu_init: no such device");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    mmgpu_ctx *c = new mmgpu_ctx();
    c->device = device_id;
    c->compute_units = prop.multiProcessorCount;
    c->name = prop.name;
    *out = c;
    return MMGPU_OK;
}

static void free_db(DeviceDb &db) {
    dev_free(db.res);
    dev_free(db.off4);
    dev_free(db.len);
    db = DeviceDb();
}

namespace mmgpu {
// what mmgpu_load_targets drops before it loads: the resident targets, their masked view, the index over them, the shard description
void db_release(mmgpu_ctx *c) {
    pf_index_free(c);
    c->shard.on = false;
    if (c->pf_masked_res) { dev_free(c->pf_masked_res); c->pf_masked_res = nullptr; }
    free_db(c->db);
    c->h_len.clear();
    c->mean_len = 0;
}
}  // namespace mmgpu

extern "C" void mmgpu_destroy(mmgpu_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_db(c->db);
    if (c->pf_masked_res) { dev_free(c->pf_masked_res); c->pf_masked_res = nullptr; }
    mmgpu::pf_index_free(c);
    (void)hipDeviceSynchronize();
    mmgpu::comm_free(c);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    for (auto &st : c->side) if (st) (void)hipStreamDestroy(st);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (auto &e : c->join) if (e) (void)hipEventDestroy(e);
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->cache->trim();
    c->cache->closed = true;
    delete c;
}

extern "C" int mmgpu_set_stream(mmgpu_ctx *c, void *s) {
    if (!c) return fail(MMGPU_ERR_ARG, "ctx is NULL");
    c->stream = reinterpret_cast<hipStream_t>(s);
    return MMGPU_OK;
}

extern "C" int mmgpu_synchronize(mmgpu_ctx *c) {
    if (!c) return fail(MMGPU_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MMGPU_OK;
}

extern "C" int mmgpu_device_info(mmgpu_ctx *c, int *cus, char *name, int cap) {
    if (!c) return fail(MMGPU_ERR_ARG, "ctx is NULL");
    if (cus) *cus = c->compute_units;
    if (name && cap > 0) {
        strncpy(name, c->name.c_str(), (size_t)cap - 1);
        name[cap - 1] = 0;
    }
    return MMGPU_OK;
}

extern "C" int mmgpu_device_memory(mmgpu_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!c) return fail(MMGPU_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MMGPU_OK;
}

extern "C" int mmgpu_load_targets(mmgpu_ctx *c, const uint8_t *residues, const uint64_t *offsets, uint32_t n,
                                  int alphabet) {
    if (!c || !residues || !offsets) return fail(MMGPU_ERR_ARG, "mmgpu_load_targets: NULL argument");
    if (alphabet < 2 || alphabet > 254) return fail(MMGPU_ERR_ARG, "mmgpu_load_targets: bad alphabet size");
    HIP_TRY(hipSetDevice(c->device));
    // a resident prefilter index belongs to the database it was built / loaded for: it goes with it
    mmgpu::pf_index_free(c);
    c->shard.on = false;
    if (c->pf_masked_res) { dev_free(c->pf_masked_res); c->pf_masked_res = nullptr; }
    free_db(c->db);
    std::vector<uint32_t> off4(std::max<uint32_t>(n, 1)), len(std::max<uint32_t>(n, 1));
    uint64_t cur4 = 16;      // 64 bytes of padding in front: the reverse scan reads up to 3 bytes before a target (sw_kernel.hip)
    uint32_t max_len = 0;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(MMGPU_ERR_ARG, "mmgpu_load_targets: offsets not monotone");
        uint64_t l = offsets[i + 1] - offsets[i];
        if (l > 65535) return fail(MMGPU_ERR_ARG, "mmgpu_load_targets: sequence longer than 65535 (Parameters.h:271)");
        if (cur4 > 0xFFFFFFFFull) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_load_targets: more than 16 GiB of residues per shard");
        off4[i] = (uint32_t)cur4;
        len[i] = (uint32_t)l;
        max_len = std::max<uint32_t>(max_len, (uint32_t)l);
        total += l;
        cur4 += (l + 3) / 4;
    }
    const size_t bytes = (size_t)cur4 * 4 + max_len + 64;
    // Layout in HBM: 64 B of padding, every target at a 4-byte boundary (pad letters = the "no letter" code), max_len + 64 B of
    // padding behind the last.  The device buffer is filled with the pad code, then the targets travel in chunks: the host threads
    // pack chunk j + 1 into one of two pinned staging buffers while the copy engine moves chunk j (a pageable 280 MB buffer packed
    // first and copied then cost 0.14 s for the 1 M targets of configs[2], 2 GB/s).
    DeviceDb db;
    uint8_t *stage[2] = {nullptr, nullptr};
    hipEvent_t moved[2] = {nullptr, nullptr};
    hipStream_t up = nullptr;
    auto drop = [&]() {
        for (int k = 0; k < 2; k++) {
            if (stage[k]) (void)hipHostFree(stage[k]);
            if (moved[k]) (void)hipEventDestroy(moved[k]);
        }
        if (up) (void)hipStreamDestroy(up);
    };
#define DB_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { drop(); free_db(db); return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    DB_TRY(dev_malloc_ctx(c, (void **)&db.res, bytes));
    DB_TRY(dev_malloc_ctx(c, (void **)&db.off4, off4.size() * sizeof(uint32_t)));
    DB_TRY(dev_malloc_ctx(c, (void **)&db.len, len.size() * sizeof(uint32_t)));
    DB_TRY(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    DB_TRY(hipMemsetAsync(db.res, alphabet, bytes, up));
    DB_TRY(hipMemcpyAsync(db.off4, off4.data(), off4.size() * sizeof(uint32_t), hipMemcpyHostToDevice, up));
    DB_TRY(hipMemcpyAsync(db.len, len.data(), len.size() * sizeof(uint32_t), hipMemcpyHostToDevice, up));
    const size_t chunk_bytes = (size_t)std::min<uint64_t>(32ull << 20, std::max<uint64_t>((uint64_t)(cur4 - 16) * 4, 4096));
    std::atomic<bool> bad_res(false);
    for (uint32_t first = 0, j = 0; first < n; j++) {
        // targets [first, last): at most chunk_bytes of packed residues (one target alone may exceed it by < 64 KB: staged is sized for that)
        const uint64_t base = (uint64_t)off4[first] * 4;
        uint32_t last = first + 1;
        {
            uint32_t lo = first + 1, hi = n;        // largest last with packed end <= base + chunk_bytes
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo + 1) / 2;
                const uint64_t end_mid = mid < n ? (uint64_t)off4[mid] * 4 : (uint64_t)cur4 * 4;
                if (end_mid - base <= chunk_bytes) lo = mid; else hi = mid - 1;
            }
            last = lo;
        }
        const uint64_t end = last < n ? (uint64_t)off4[last] * 4 : (uint64_t)cur4 * 4;
        const int k = (int)(j & 1);
        if (!stage[k]) {
            DB_TRY(hipHostMalloc((void **)&stage[k], chunk_bytes + 65536 + 64, hipHostMallocDefault));
            DB_TRY(hipEventCreateWithFlags(&moved[k], hipEventDisableTiming));
        } else {
            DB_TRY(hipEventSynchronize(moved[k]));      // the copy out of this buffer two chunks ago
        }
        uint8_t *dst = stage[k];
        parallel_for((size_t)(last - first), [&, first, base, dst](size_t a, size_t b) {
            for (size_t i = first + a; i < first + b; i++) {
                const uint8_t *src = residues + offsets[i];
                uint8_t *out = dst + ((uint64_t)off4[i] * 4 - base);
                const uint32_t l = len[i];
                uint8_t top = 0;
                for (uint32_t x = 0; x < l; x++) top = std::max(top, src[x]);
                if (top >= alphabet) bad_res = true;
                memcpy(out, src, l);
                for (uint32_t x = l; x < ((l + 3) & ~3u); x++) out[x] = (uint8_t)alphabet;
            }
        });
        DB_TRY(hipMemcpyAsync(db.res + base, dst, end - base, hipMemcpyHostToDevice, up));
        DB_TRY(hipEventRecord(moved[k], up));
        first = last;
    }
    DB_TRY(hipStreamSynchronize(up));
#undef DB_TRY
    drop();
    if (bad_res) { free_db(db); return fail(MMGPU_ERR_ARG, "mmgpu_load_targets: residue code >= alphabet"); }
    db.n = n;
    db.res_bytes = bytes;
    db.max_len = max_len;
    db.total_residues = total;
    db.alphabet = alphabet;
    c->db = db;
    c->h_len.assign(len.begin(), len.begin() + n);
    uint64_t res_total = 0;
    for (uint32_t i = 0; i < n; i++) res_total += len[i];
    c->mean_len = n ? (uint32_t)(res_total / n) : 0;
    return MMGPU_OK;
}

// Loads the code objects of the hot kernels now instead of at their first launch (the runtime loads a translation unit's code
// object lazily: 0.1 - 0.2 s in total, paid inside the first prefilter block and the first alignment batch of a process).  A
// caller that has something else to do meanwhile - the fused search masks / reads its databases on the host - calls this on a
// helper thread right after mmgpu_init.
extern "C" int mmgpu_warmup(mmgpu_ctx *c) {
    if (!c) return fail(MMGPU_ERR_ARG, "mmgpu_warmup: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    mmgpu::warm_tantan();
    mmgpu::warm_ix();
    mmgpu::warm_pf();
    mmgpu::warm_sw();
    mmgpu::warm_block();
    mmgpu::warm_block4();
    mmgpu::warm_bt();
    return MMGPU_OK;
}

// tantan masking of the resident targets for the prefilter (tantan_kernel.hip): what IndexBuilder::fillDatabase does to every
// target before it counts k-mers (IndexBuilder.cpp:148, Masker.cpp:14-57 with maskTantan only).  The alignment kernels keep
// reading the unmasked residues.
extern "C" int mmgpu_pf_mask_targets(mmgpu_ctx *c, const double *likelihood_ratios, int alphabet, double min_mask_prob, int mask_letter,
                                     uint64_t *n_masked) {
    if (!c) return fail(MMGPU_ERR_ARG, "mmgpu_pf_mask_targets: NULL argument");
    if (!c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_pf_mask_targets: no targets loaded");
    if (!likelihood_ratios) {      // back to the unmasked view (--mask 0 after --mask 1 on a resident database): the masked copy and its index go
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        mmgpu::pf_index_free(c);
        if (c->pf_masked_res) dev_free(c->pf_masked_res);
        c->pf_masked_res = nullptr;
        if (n_masked) *n_masked = 0;
        return MMGPU_OK;
    }
    if (alphabet != c->db.alphabet || alphabet > 32) return fail(MMGPU_ERR_ARG, "mmgpu_pf_mask_targets: alphabet differs from the loaded targets (or exceeds 32)");
    if (mask_letter < 0 || mask_letter >= alphabet) return fail(MMGPU_ERR_ARG, "mmgpu_pf_mask_targets: mask letter outside the alphabet");
    HIP_TRY(hipSetDevice(c->device));
    mmgpu::pf_index_free(c);      // an index built from the unmasked residues does not describe the masked ones
    hipStream_t s = c->stream;
    const uint32_t n = c->db.n;
    if (!c->pf_masked_res) HIP_TRY(dev_malloc_ctx(c, (void **)&c->pf_masked_res, c->db.res_bytes));
    HIP_TRY(hipMemcpyAsync(c->pf_masked_res, c->db.res, c->db.res_bytes, hipMemcpyDeviceToDevice, s));
    // targets in order of length (longest first): 64 consecutive ones share a wavefront and end together; counting sort
    std::vector<uint32_t> order(std::max<uint32_t>(n, 1));
    {
        std::vector<uint32_t> first(65536 + 2, 0);
        for (uint32_t i = 0; i < n; i++) first[65535 - std::min<uint32_t>(c->h_len[i], 65535) + 1]++;
        for (size_t k = 1; k < first.size(); k++) first[k] += first[k - 1];
        for (uint32_t i = 0; i < n; i++) order[first[65535 - std::min<uint32_t>(c->h_len[i], 65535)]++] = i;
    }
    const uint32_t n_waves = (n + 63) / 64;
    std::vector<uint64_t> pbase(std::max<uint32_t>(n_waves, 1)), sbase(std::max<uint32_t>(n_waves, 1));
    uint64_t pcur = 0, scur = 0;
    for (uint32_t w = 0; w < n_waves; w++) {
        const uint64_t longest = c->h_len[order[(size_t)w * 64]];      // the first of a wavefront is its longest
        pbase[w] = pcur;
        sbase[w] = scur;
        pcur += longest * 64;
        scur += (longest / 16 + 1) * 64;
    }
    // Tantan's constructor (tantan.cpp:94-131) with the Masker's constants (Masker.cpp:22-31)
    const double repeat_prob = 0.005, repeat_end_prob = 0.05, decay = 0.9;
    const int max_offset = 50;
    std::vector<double> b2f(max_offset);
    {
        const double first_prob = (decay < 1 || decay > 1) ? (1 - decay) / (1 - std::pow(decay, max_offset)) : 1.0 / max_offset;   // :38-44
        double p = repeat_prob * first_prob;
        for (int i = 0; i < max_offset; i++) {
            b2f[i] = p;
            p *= decay;
        }
    }
    DevBuf d_order, d_pbase, d_sbase, d_lr, d_b2f, d_probs, d_scales, d_count;
    for (DevBuf *d : {&d_order, &d_pbase, &d_sbase, &d_lr, &d_b2f, &d_probs, &d_scales, &d_count}) d->bind(c->cache);
    HIP_TRY(upload(d_order, order, s));
    HIP_TRY(upload(d_pbase, pbase, s));
    HIP_TRY(upload(d_sbase, sbase, s));
    std::vector<double> lr(likelihood_ratios, likelihood_ratios + (size_t)alphabet * alphabet);
    HIP_TRY(upload(d_lr, lr, s));
    HIP_TRY(upload(d_b2f, b2f, s));
    // The forward probabilities (4 B per residue) and scale factors (0.5 B) are scratch: wavefronts run in chunks whose scratch stays
    // below MMGPU_TANTAN_SCRATCH_MB (default 8 GB; a quarter of the free memory if that is less) - one chunk for 1 M targets, several
    // for a shard near the 16 G-residue limit, which would otherwise ask for 70 GB on top of the masked copy.
    uint64_t budget = (getenv("MMGPU_TANTAN_SCRATCH_MB") ? strtoull(getenv("MMGPU_TANTAN_SCRATCH_MB"), nullptr, 10) : 8192ull) << 20;
    {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) budget = std::min<uint64_t>(budget, std::max<uint64_t>(mem_free / 4, 64ull << 20));
    }
    std::vector<uint32_t> chunk_first(1, 0);      // wavefronts [chunk_first[k], chunk_first[k + 1])
    uint64_t chunk_p = 0, chunk_s = 0;            // the largest chunk's scratch (floats, doubles)
    auto p_end = [&](uint32_t w) { return w < n_waves ? pbase[w] : pcur; };
    auto s_end = [&](uint32_t w) { return w < n_waves ? sbase[w] : scur; };
    for (uint32_t w0 = 0; w0 < n_waves;) {
        uint32_t w1 = w0 + 1;      // (a single wavefront over the budget runs alone)
        while (w1 < n_waves && (p_end(w1 + 1) - pbase[w0]) * 4 + (s_end(w1 + 1) - sbase[w0]) * 8 <= budget) w1++;
        chunk_p = std::max(chunk_p, p_end(w1) - pbase[w0]);
        chunk_s = std::max(chunk_s, s_end(w1) - sbase[w0]);
        chunk_first.push_back(w1);
        w0 = w1;
    }
    HIP_TRY(d_probs.alloc(std::max<uint64_t>(chunk_p, 1) * sizeof(float)));
    HIP_TRY(d_scales.alloc(std::max<uint64_t>(chunk_s, 1) * sizeof(double)));
    HIP_TRY(d_count.alloc(8));
    HIP_TRY(hipMemsetAsync(d_count.p, 0, 8, s));
    TantanArgs A;
    A.t_res = c->db.res;
    A.out_res = c->pf_masked_res;
    A.t_off4 = c->db.off4;
    A.t_len = c->db.len;
    A.order = d_order.as<uint32_t>();
    A.n = n;
    A.lr = d_lr.as<double>();
    A.b2f = d_b2f.as<double>();
    A.alphabet = alphabet;
    A.repeat_prob = repeat_prob;
    A.repeat_end_prob = repeat_end_prob;
    A.min_mask_prob = min_mask_prob;
    A.mask_letter = (uint8_t)mask_letter;
    A.n_masked = d_count.as<unsigned long long>();
    for (size_t k = 0; k + 1 < chunk_first.size(); k++) {      // (same stream: a chunk's scratch is free again when the next one starts)
        const uint32_t w0 = chunk_first[k], w1 = chunk_first[k + 1];
        A.order = d_order.as<uint32_t>() + (size_t)w0 * 64;
        A.n = std::min<uint32_t>(n - w0 * 64, (w1 - w0) * 64);
        // the bases stay absolute: the scratch pointers are moved back by the chunk's first base instead
        A.probs = d_probs.as<float>() - pbase[w0];
        A.scales = d_scales.as<double>() - sbase[w0];
        A.wave_prob_base = d_pbase.as<uint64_t>() + w0;
        A.wave_scale_base = d_sbase.as<uint64_t>() + w0;
        HIP_TRY(launch_tantan_mask(A, s));
    }
    unsigned long long masked = 0;
    HIP_TRY(hipMemcpyAsync(&masked, d_count.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));      // the host vectors above die with this scope
    if (n_masked) *n_masked = masked;
    return MMGPU_OK;
}

// test hook: the prefilter's (masked) view of the resident targets back on the host, in the caller's layout
extern "C" int mmgpu_pf_debug_masked_targets(mmgpu_ctx *c, const uint64_t *offsets, uint32_t n, uint8_t *residues) {
    if (!c || !offsets || !residues) return fail(MMGPU_ERR_ARG, "mmgpu_pf_debug_masked_targets: NULL argument");
    if (!c->db.res || n != c->db.n) return fail(MMGPU_ERR_STATE, "mmgpu_pf_debug_masked_targets: not the resident target set");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint8_t> packed(c->db.res_bytes);
    HIP_TRY(hipMemcpy(packed.data(), c->pf_res(), packed.size(), hipMemcpyDeviceToHost));
    std::vector<uint32_t> off4(std::max<uint32_t>(n, 1));
    HIP_TRY(hipMemcpy(off4.data(), c->db.off4, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) memcpy(residues + offsets[i], packed.data() + (size_t)off4[i] * 4, (size_t)(offsets[i + 1] - offsets[i]));
    return MMGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------------------
// SubstitutionMatrix::calcLocalAaBiasCorrection, src/commons/SubstitutionMatrix.cpp:79-112.  The reference mixes
// float and double (`deltaS_i /= -1.0 * (float)windowLength`, `deltaS_i += pBack[a] * (float)subMat[a]`); each
// statement below rounds where the reference's expression rounds.
extern "C" int mmgpu_host_comp_bias(const int16_t *submat, const double *pback, int alphabet, const uint8_t *seq,
                                    uint32_t len, float scale, float *out) {
    if (!submat || !pback || (!seq && len) || (!out && len)) return fail(MMGPU_ERR_ARG, "mmgpu_host_comp_bias: NULL argument");
    const int n = (int)len, half_window = 20;
    for (int i = 0; i < n; i++) {
        if (seq[i] >= alphabet) return fail(MMGPU_ERR_ARG, "mmgpu_host_comp_bias: residue code >= alphabet");
        const int lo = std::max(0, i - half_window), hi = std::min(n, i + half_window);
        const int16_t *row = submat + (int)seq[i] * alphabet;
        int sum = 0;
        for (int j = lo; j < hi; j++) sum += row[seq[j]];
        sum -= row[seq[i]];
        float delta = (float)sum;
        delta = (float)((double)delta / (-1.0 * (double)(float)(hi - lo)));
        for (int a = 0; a < alphabet; a++) delta = (float)((double)delta + pback[a] * (double)(float)row[a]);
        out[i] = scale * delta;
    }
    return MMGPU_OK;
}

// The same correction for a whole block of queries (what Prefiltering::runSplit / Alignment::run compute per query inside
// their OpenMP loops, QueryMatcher.cpp:108-116, Matcher::initQuery -> ssw_init StripedSmithWaterman.cpp:1371-1381): sequences
// are dealt to n_threads host threads.  out_float (may be NULL) receives the float bias, out_round (may be NULL) ssw_init's int8
// rounding of it; both are indexed like `residues`.
extern "C" int mmgpu_host_comp_bias_batch(const int16_t *submat, const double *pback, int alphabet, const uint8_t *residues,
                                          const uint64_t *offsets, uint32_t n, float scale, float *out_float, int8_t *out_round,
                                          int n_threads) {
    if (!submat || !pback || !offsets || (!residues && n && offsets[n])) return fail(MMGPU_ERR_ARG, "mmgpu_host_comp_bias_batch: NULL argument");
    if (n_threads < 1) n_threads = 1;
    n_threads = (int)std::min<uint32_t>((uint32_t)n_threads, std::max<uint32_t>(n, 1u));
    std::atomic<uint32_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        std::vector<float> tmp;
        for (;;) {
            const uint32_t b = next.fetch_add(64);
            if (b >= n) return;
            for (uint32_t i = b; i < std::min(n, b + 64); i++) {
                const uint64_t o = offsets[i];
                const uint32_t len = (uint32_t)(offsets[i + 1] - o);
                float *dst = out_float ? out_float + o : (tmp.resize(len), tmp.data());
                if (mmgpu_host_comp_bias(submat, pback, alphabet, residues + o, len, scale, dst) != MMGPU_OK) { bad = 1; return; }
                if (out_round) mmgpu_host_round_comp_bias(dst, len, out_round + o);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return bad ? fail(MMGPU_ERR_ARG, "mmgpu_host_comp_bias_batch: residue code >= alphabet") : MMGPU_OK;
}

// Length-bucket sharding of a target database over n_shards devices (the reference balances
// residues over its splits, DBReader::decomposeDomainByAminoAcid, src/commons/DBReader.cpp:1108-1150, and libmarv deals
// length partitions to devices).  Targets are binned by length (boundaries below), the bins are walked from the longest
// to the shortest and their members dealt round-robin, the deal continuing from bin to bin: every shard receives the same
// length distribution (+-1 sequence per bin) and therefore the same number of residues, index entries and alignment
// cells to within a fraction of a percent.  Inside a shard the local ids follow the global ids (ascending), so that
// "same score -> smaller id first" orders of the shard and of the whole database agree.
extern "C" int mmgpu_host_partition_targets(const uint64_t *offsets, uint32_t n, uint32_t n_shards, uint32_t *shard_of,
                                            uint32_t *local_id, uint32_t *shard_sizes, uint64_t *shard_residues) {
    if (!offsets || !shard_of || !local_id || !shard_sizes || n_shards == 0) return fail(MMGPU_ERR_ARG, "mmgpu_host_partition_targets: bad argument");
    static const uint32_t bound[] = {32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 448, 512, 576, 640, 768, 896, 1024,
                                     1280, 1536, 2048, 2560, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536};
    const int nb = (int)(sizeof(bound) / sizeof(bound[0]));
    std::vector<uint8_t> bin(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t len = offsets[i + 1] - offsets[i];
        int b = 0;
        while (b < nb - 1 && len > bound[b]) b++;
        bin[i] = (uint8_t)b;
    }
    uint32_t next = 0;
    for (int b = nb - 1; b >= 0; b--)
        for (uint32_t i = 0; i < n; i++)
            if (bin[i] == b) {
                shard_of[i] = next;
                next = next + 1 == n_shards ? 0 : next + 1;
            }
    for (uint32_t s = 0; s < n_shards; s++) {
        shard_sizes[s] = 0;
        if (shard_residues) shard_residues[s] = 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        local_id[i] = shard_sizes[shard_of[i]]++;
        if (shard_residues) shard_residues[shard_of[i]] += offsets[i + 1] - offsets[i];
    }
    return MMGPU_OK;
}

extern "C" int mmgpu_host_round_comp_bias(const float *bias, uint32_t len, int8_t *out) {
    if ((!bias || !out) && len) return fail(MMGPU_ERR_ARG, "mmgpu_host_round_comp_bias: NULL argument");
    for (uint32_t i = 0; i < len; i++) out[i] = (int8_t)((bias[i] < 0.0f) ? bias[i] - 0.5 : bias[i] + 0.5);
    return MMGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Smith-Waterman batches
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t JOB_HITS = 256;   // most hits per workgroup (8 rounds of 32); the reverse pass packs a job with one thread per hit (sw_kernel.hip)
static_assert(JOB_HITS <= 256, "a job may hold at most one hit per thread of the 256-thread workgroup");
constexpr uint32_t JOB_ROUND = 32;   // targets a workgroup has in flight
constexpr uint64_t JOB_CELLS = 60000000ull;   // cut a job once it holds this many forward cells
constexpr uint32_t LONG_QUERY = 1024;  // multi-tile queries from this length on are scheduled per wave (8 targets)
}  // namespace

struct mmgpu_sw_batch_t {
    int mode = 0;
    int alphabet = 0, gap_open = 0, gap_extend = 0;
    uint64_t cells = 0, pairs = 0;
    uint64_t valid_pairs = 0;   // from_pf: slots actually holding a hit (pairs counts all slots)
    uint32_t n_queries = 0;
    // all jobs of the batch, by kernel group (sw_kernel.hip), longest first inside a group; SwJob::shape picks the body
    uint32_t n_jobs = 0, n_multi_jobs = 0;
    uint32_t n_rev_jobs = 0;      // reverse-scan jobs of the multi-tile queries (mode START), behind the forward jobs in d_jobs
    uint32_t group_begin[SW_GROUPS + 1] = {};
    size_t group_lds[SW_GROUPS] = {};   // largest profile of any shape present in the group
    DevBuf d_jobs;
    DevBuf d_scratch;             // multi-tile jobs: [scratch slot][4 waves][4 groups][2 buffers][scratch_cols] x uint2
    DevBuf d_scratch_busy;        // one flag per slot of the pool (sw_kernel claims / releases)
    uint32_t scratch_slots = 0;
    DevBuf d_pf_counts, d_slot_target;   // from_pf: list lengths and slot -> target id, copied out of the prefilter batch
    DevBuf d_qout_off, d_out_target;     // mmgpu_sw_block_starts: h_qout_off / h_out_target on the device (uploaded on first use)
    DevBuf d_qres, d_qcb, d_qoff, d_qbias, d_qminstart, d_hit_target, d_hit_out, d_out, d_mat;
    DevBuf d_qprof, d_qprof_off;   // profile queries only (empty otherwise)
    bool any_profile = false;
    // fused hand-over from a prefilter batch (mmgpu_sw_prepare_from_pf): lists, counts and statistics live on the device
    bool from_pf = false;
    uint32_t pf_stride = 0, slot_stride = 0;
    DevBuf d_stats;                        // [2] unsigned long long: cells, pairs
    std::vector<uint32_t> h_out_target;   // target id of every result slot (kept for mmgpu_sw_traceback, mode >= START)
    std::vector<uint32_t> h_qout_off;     // [nq + 1] first result slot of every query
    std::vector<uint32_t> h_qoff;         // [nq + 1] residue offsets
    std::vector<uint8_t> h_query_is_profile;   // profile queries of the batch (empty: none)
    DevBuf d_bt_scratch, d_bt_jobs, d_bt_info, d_bt_str, d_bt_cursor;
    uint32_t scratch_cols = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // one pair per mmgpu_sw_run since prepare
    bool ran = false;
    // mmgpu_sw_traceback's host copies of the last run's results
    std::vector<mmgpu_sw_hit> h_res;
    std::vector<uint32_t> h_slot_target;
    bool h_res_valid = false;
    uint32_t block_pairs_tier[3] = {0, 0, 0};   // last mmgpu_sw_block_backtrace call: pairs decided with blocks <= 512 / 2048 / 4096 rows
    uint32_t block_pairs_fast = 0;              // ... by block4_kernel.hip's first launch (four pairs per wavefront, blocks <= 256 rows)
    uint32_t block_pairs_skew = 0;              // ... and by its skewed launches (one pair per wavefront, blocks <= 1024 / 4096 rows)
    // pairs this rank owns of a sharded run's merged lists (mmgpu_sw_prepare_owned / mmgpu_sw_gather_owned)
    bool owned = false;
    uint32_t o_stride = 0, o_cap = 0;
    int o_ranks = 0;
    bool o_dense = false;          // a rank's send buffer overflowed once: every rank's buffer holds all slots from now on
    DevBuf o_lhits, o_lcounts, o_lslot;        // merged lists restricted to this shard's targets (local ids) + their list positions
    DevBuf o_send, o_counter, o_recv, o_recv_counters, o_full, o_status;
};

// which kernel body serves a query of this length: 16 lanes x R rows per tile (any R since round 5: the padding of a tile is
// below 16 rows, it was below 32 with even R only), at most 16 * SW_MAX_R rows per tile; longer queries are cut into equal tiles
// (multi-tile bodies).
static void pick_class(uint32_t qlen, int *rows_per_lane, bool *multi) {
    constexpr uint32_t max_rows = 16u * (uint32_t)SW_MAX_R;
    const uint32_t n_tiles = (qlen + max_rows - 1) / max_rows;
    const uint32_t rows = (qlen + n_tiles - 1) / n_tiles;       // rows per tile before rounding
    *rows_per_lane = (int)std::max<uint32_t>(1u, (rows + 15) / 16);
    *multi = n_tiles > 1;
}

// Slots of a job that should hold about `rounds` rounds of `round` hits: whole workgroup rounds (4 waves x 8), so
// that no wave idles while its neighbours run a second round; only jobs below one workgroup round (long queries cut
// per wave) may hold 8 or 16.
static uint32_t job_slots(uint32_t round, uint64_t rounds) {
    const uint64_t slots = std::max<uint64_t>(1, rounds) * round;
    if (slots >= JOB_ROUND) return (uint32_t)std::min<uint64_t>(JOB_HITS, slots / JOB_ROUND * JOB_ROUND);
    return slots >= 16 ? 16u : round;
}

// device-resident hit lists an alignment batch is built from (the fused hand-over and the multi-GPU path)
struct DeviceLists {
    const mmgpu_pf_hit *hits = nullptr;
    const uint32_t *counts = nullptr;
    uint32_t stride = 0;
};

static int sw_prepare_impl(mmgpu_ctx *c, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq, int mode,
                           const DeviceLists *pf, mmgpu_sw_batch_t **out) {
    if (!c || !par || !out || (!qs && nq)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: NULL argument");
    const mmgpu_pf_hit *pf_hits = nullptr;
    const uint32_t *pf_counts = nullptr;
    uint32_t pf_stride = 0;
    if (pf) {
        pf_hits = pf->hits;
        pf_counts = pf->counts;
        pf_stride = pf->stride;
        if (pf_stride > (uint32_t)SW_PF_MAX_LIST) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_prepare_from_pf: lists longer than 16384");
    }
    if (!c->db.res) return fail(MMGPU_ERR_STATE, "mmgpu_sw_prepare: no targets loaded");
    static const bool prep_trace = getenv("MMGPU_TRACE") != nullptr;      // where the host side of a batch's preparation goes
    auto prep_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double prep_mark = prep_now();
    auto prep_lap = [&](const char *what) {
        if (!prep_trace) return;
        const double t = prep_now();
        fprintf(stderr, "[mmgpu sw_prepare] %s %.3f s\n", what, t - prep_mark);
        prep_mark = t;
    };
    if (par->alphabet != c->db.alphabet) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: alphabet differs from the loaded targets");
    if (mode != MMGPU_SW_SCORE_END && mode != MMGPU_SW_START && mode != MMGPU_SW_START_NOT_WORD) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_prepare: unknown mode");
    if (par->gap_open < par->gap_extend || par->gap_extend < 0 || par->gap_open > 32767)
        return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: need 0 <= gap_extend <= gap_open");
    // The cell-by-cell recurrence equals the reference's striped lazy-F result only if opening a gap right after
    // a gap in the other direction never beats a substitution (StripedSmithWaterman.cpp:205).
    int minp = 0;
    for (int i = 0; i < par->alphabet * par->alphabet; i++) minp = std::min<int>(minp, par->mat[i]);
    HIP_TRY(hipSetDevice(c->device));

    mmgpu_sw_batch_t *b = new mmgpu_sw_batch_t();
    b->mode = mode;
    b->alphabet = par->alphabet;
    b->gap_open = par->gap_open;
    b->gap_extend = par->gap_extend;
    b->n_queries = nq;
    for (DevBuf *d : {&b->d_qres, &b->d_qcb, &b->d_qoff, &b->d_qbias, &b->d_qminstart, &b->d_hit_target, &b->d_hit_out, &b->d_out,
                      &b->d_mat, &b->d_stats, &b->d_bt_scratch, &b->d_bt_jobs, &b->d_bt_info, &b->d_bt_str, &b->d_bt_cursor, &b->d_scratch_busy,
                      &b->d_pf_counts, &b->d_slot_target, &b->d_qprof, &b->d_qprof_off, &b->d_qout_off, &b->d_out_target})
        d->bind(c->cache);

    std::vector<uint8_t> qres;
    std::vector<int8_t> qcb;
    std::vector<uint32_t> qoff(nq + 1, 0);
    std::vector<int32_t> qbias(std::max<uint32_t>(nq, 1), 0), qminstart(std::max<uint32_t>(nq, 1), 0);
    std::vector<int8_t> qprof;                       // profile queries: [alphabet][qlen] blocks, concatenated
    std::vector<uint32_t> qprof_off(std::max<uint32_t>(nq, 1), 0xFFFFFFFFu);
    uint64_t total_hits = 0;
    for (uint32_t i = 0; i < nq; i++) {
        // a query without targets may come without residues (Alignment::run never maps the query of an empty list, :322)
        const bool empty_ok = !pf && qs[i].n_targets == 0 && qs[i].qlen == 0;
        if (!empty_ok && (qs[i].qlen == 0 || qs[i].qlen > 65535 || !qs[i].q)) { delete b; return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: bad query"); }
        qoff[i + 1] = qoff[i] + qs[i].qlen;
        total_hits += pf ? pf_stride : qs[i].n_targets;
    }
    if (total_hits > 0xFFFFFFF0ull) { delete b; return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_prepare: more than 2^32 pairs in one batch"); }
    qres.resize(qoff[nq]);
    qcb.assign(qoff[nq], 0);
    std::vector<uint32_t> hit_target(pf ? 0 : (size_t)total_hits), hit_out(pf ? 0 : (size_t)total_hits);
    b->d_jobs.bind(c->cache);
    b->d_scratch.bind(c->cache);
    std::vector<SwJob> jobs;
    std::vector<SwJob> rev_jobs;          // multi-tile queries, mode START: the reverse scan runs per query (sw_rev_multi_kernel)
    std::vector<uint64_t> rev_cells;
    std::vector<uint64_t> job_cells;
    uint32_t n_multi = 0;
    struct Deferred { uint32_t query, hit_cursor, out_cursor, shape, round; bool multi; };   // caller-supplied lists, scheduled below
    std::vector<Deferred> deferred;
    uint32_t hit_cursor = 0, out_cursor = 0;
    b->h_qout_off.assign(nq + 1, 0);
    if (mode >= MMGPU_SW_START && !pf) b->h_out_target.resize((size_t)total_hits);
    uint32_t max_tlen = 0;
    bool any_multi = false;
    // Reverse-scan jobs of a multi-tile query: consecutive slots of its list, as many as hold about six forward jobs'
    // worth of cells (one pair in six reaches the start-score threshold on hit lists), whole workgroup rounds, at most
    // SW_REV_JOB_MAX; the kernel packs the live pairs of a job before it deals them to its waves.
    auto add_rev_jobs = [&](uint32_t query, uint32_t first, uint32_t n, uint32_t shape, uint64_t cells_per_hit) {
        uint64_t per = 6 * JOB_CELLS / std::max<uint64_t>(cells_per_hit, 1);
        per = std::min<uint64_t>(std::max<uint64_t>(per / JOB_ROUND * JOB_ROUND, JOB_ROUND), (uint64_t)SW_REV_JOB_MAX);
        for (uint32_t k = 0; k < n; k += (uint32_t)per) {
            SwJob j;
            j.query = query;
            j.hit_begin = first + k;
            j.hit_end = first + std::min<uint32_t>(k + (uint32_t)per, n);
            j.shape = shape;
            rev_jobs.push_back(j);
            rev_cells.push_back(cells_per_hit * (j.hit_end - j.hit_begin) * 65536u + (n - k));
        }
    };
    for (uint32_t i = 0; i < nq; i++) {
        const mmgpu_sw_query &Q = qs[i];
        if (Q.qlen == 0) {      // empty list, no residues: no jobs, no result slots
            b->h_qout_off[i + 1] = out_cursor;
            continue;
        }
        memcpy(qres.data() + qoff[i], Q.q, Q.qlen);
        int mincb = 0;
        int qminp = minp;
        for (uint32_t k = 0; k < Q.qlen; k++) {
            if (Q.q[k] >= par->alphabet) { delete b; return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: query residue code >= alphabet"); }
            if (Q.comp_bias && !Q.profile) { qcb[qoff[i] + k] = Q.comp_bias[k]; mincb = std::min<int>(mincb, Q.comp_bias[k]); }
        }
        if (Q.profile) {
            // ssw_init with a profile query (StripedSmithWaterman.cpp:1386-1406): the first PROFILE_AA_SIZE letter rows are the
            // profile, the X row scores 0, no composition bias; bias = |min| over the profile rows
            if (Q.profile_letters == 0 || (int)Q.profile_letters > par->alphabet) {
                delete b;
                return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: profile_letters must be in [1, alphabet]");
            }
            if (qprof.size() + (size_t)par->alphabet * Q.qlen > 0xFFFFFFF0ull) { delete b; return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_prepare: more than 4 GB of query profiles in one batch"); }
            qprof_off[i] = (uint32_t)qprof.size();
            qprof.resize(qprof.size() + (size_t)par->alphabet * Q.qlen, 0);
            const uint32_t rows = std::min<uint32_t>(Q.profile_letters, (uint32_t)par->alphabet - 1);   // :1389-1390 zeroes the last letter (X)
            memcpy(qprof.data() + qprof_off[i], Q.profile, (size_t)rows * Q.qlen);
            qminp = 0;
            for (size_t k = 0; k < (size_t)Q.profile_letters * Q.qlen; k++) qminp = std::min<int>(qminp, Q.profile[k]);
            b->any_profile = true;
            b->h_query_is_profile.resize(nq, 0);
            b->h_query_is_profile[i] = 1;
        }
        if (!(qminp + mincb + par->gap_extend > -par->gap_open)) {
            delete b;
            return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_prepare: gap penalties too small for this matrix (adjacent insertion+deletion could win)");
        }
        qbias[i] = std::abs(qminp) + std::abs(mincb);   // ssw_init :1397-1406
        qminstart[i] = Q.min_start_score;
        int rpl; bool multi;
        pick_class(Q.qlen, &rpl, &multi);
        any_multi |= multi;
        const uint32_t shape = (multi ? 32u : 0u) + (uint32_t)rpl - 1;   // [0,32): single tile R = 1..32, [32,64): multi-tile
        const int grp = sw_shape_group(shape);
        b->group_lds[grp] = std::max(b->group_lds[grp], sw_lds_bytes(rpl, par->alphabet));
        // jobs are cut at multiples of one workgroup round (4 waves x 8 targets); for queries of several tiles a
        // single wave's 8 targets already run for milliseconds, so those are cut per wave to shorten the tail
        const uint32_t round = (multi && Q.qlen >= LONG_QUERY) ? JOB_ROUND / 4 : JOB_ROUND;
        if (pf) {
            // the list is on the device: fixed jobs over the query's slots, the kernel clips them to the list length
            // (SwLaunch::q_hit_count); order and statistics come from sw_from_pf_kernel.  Hits per job: as many rounds as fit JOB_CELLS at the length the hits will probably have (prefilter hits are
            // mostly about as long as the query; the database mean otherwise)
            const uint64_t est_cells = (uint64_t)Q.qlen * ((Q.qlen + c->mean_len) / 2 + 1) * round;
            const uint32_t per_job = job_slots(round, JOB_CELLS / est_cells);
            for (uint32_t k = 0; k < pf_stride; k += per_job) {
                SwJob j;
                j.query = i;
                j.hit_begin = hit_cursor + k;
                j.hit_end = hit_cursor + std::min<uint32_t>(k + per_job, pf_stride);
                j.shape = shape | (multi ? n_multi++ << 8 : 0u);
                jobs.push_back(j);
                // stand-in for the cell count the host cannot see: query length x slots; the lists are sorted by
                // target length on the device, so among equals a query's earlier jobs hold the longer targets
                job_cells.push_back((uint64_t)Q.qlen * (j.hit_end - j.hit_begin) * 65536u + (pf_stride - k));
            }
            if (multi && mode >= MMGPU_SW_START)
                add_rev_jobs(i, hit_cursor, pf_stride, shape, (uint64_t)Q.qlen * ((Q.qlen + c->mean_len) / 2 + 1));
            max_tlen = c->db.max_len;
            hit_cursor += pf_stride;
            out_cursor += pf_stride;
            b->h_qout_off[i + 1] = out_cursor;
            continue;
        }
        // caller-supplied list: the sort by target length and the job cuts are done for all queries in parallel below
        deferred.push_back(Deferred{i, hit_cursor, out_cursor, shape, round, multi});
        hit_cursor += Q.n_targets;
        out_cursor += Q.n_targets;
        b->h_qout_off[i + 1] = out_cursor;
    }
    prep_lap("queries copied, buffers sized");
    if (!deferred.empty()) {
        // Per query: sort the prefilter list by target length (longest first) so the 8 targets a wave runs together end
        // together, cut it into jobs.  3 M pairs of a 10 000-query block cost 0.2 s on one thread (the sort's comparisons
        // read the length table at random) - the queries are independent, so threads take contiguous ranges of them; the jobs
        // are concatenated in query order afterwards, so the result does not depend on the number of threads.
        struct PerQuery { std::vector<SwJob> jobs; std::vector<uint64_t> cells; uint64_t sum_cells = 0; uint32_t max_tlen = 0; uint32_t rev_mid_len = 0; bool bad = false; };
        std::vector<PerQuery> pq(deferred.size());
        auto work = [&](size_t from, size_t to) {
            std::vector<uint32_t> ord;
            for (size_t d = from; d < to; d++) {
                const Deferred &D = deferred[d];
                const mmgpu_sw_query &Q = qs[D.query];
                PerQuery &P = pq[d];
                const bool same_list = d > from && Q.target_ids == qs[deferred[d - 1].query].target_ids && Q.n_targets == qs[deferred[d - 1].query].n_targets;
                if (!same_list) {   // all-vs-all callers hand the same list to every query: sort it once (per thread)
                    ord.resize(Q.n_targets);
                    std::iota(ord.begin(), ord.end(), 0u);
                    for (uint32_t k = 0; k < Q.n_targets; k++)
                        if (Q.target_ids[k] >= c->db.n) { P.bad = true; break; }
                    if (P.bad) continue;
                    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t bb) {
                        return c->h_len[Q.target_ids[a]] > c->h_len[Q.target_ids[bb]];
                    });
                }
                for (uint32_t k = 0; k < Q.n_targets; k++) {
                    const uint32_t t = Q.target_ids[ord[k]];
                    hit_target[D.hit_cursor + k] = t;
                    hit_out[D.hit_cursor + k] = D.out_cursor + ord[k];
                    P.sum_cells += (uint64_t)Q.qlen * c->h_len[t];
                    P.max_tlen = std::max(P.max_tlen, c->h_len[t]);
                }
                // Jobs: consecutive hits of the (length-sorted) list, cut at multiples of one workgroup round (32 targets)
                // once a job holds JOB_CELLS forward cells, at the latest after JOB_HITS hits - a 5000-residue query against
                // 300 long targets must not become one 7e9-cell workgroup.
                for (uint32_t k = 0; k < Q.n_targets;) {
                    uint64_t jc = 0;
                    uint32_t e = k;
                    while (e < Q.n_targets && e - k < JOB_HITS) {
                        const uint32_t stop = std::min<uint32_t>(e + D.round, Q.n_targets);
                        for (; e < stop; e++) jc += (uint64_t)Q.qlen * c->h_len[hit_target[D.hit_cursor + e]];
                        // cut at 8, 16 (long queries) or whole workgroup rounds: no wave idles while another runs a second round
                        const uint32_t held = e - k;
                        if (jc >= JOB_CELLS && (held <= 16 || held % JOB_ROUND == 0)) break;
                    }
                    SwJob j;
                    j.query = D.query;
                    j.hit_begin = D.hit_cursor + k;
                    j.hit_end = D.hit_cursor + e;
                    j.shape = D.shape;      // (multi-tile jobs get their scratch slot number when the lists are joined)
                    P.jobs.push_back(j);
                    P.cells.push_back(jc);
                    k = e;
                }
                if (Q.n_targets) P.rev_mid_len = c->h_len[hit_target[D.hit_cursor + Q.n_targets / 2]];
                if (mode >= MMGPU_SW_START)
                    for (uint32_t k = 0; k < Q.n_targets; k++) b->h_out_target[D.out_cursor + k] = Q.target_ids[k];
            }
        };
        static const unsigned host_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        const size_t n_thr = std::max<size_t>(1, std::min<size_t>(host_threads, total_hits / 65536 + 1));
        if (n_thr <= 1) {
            work(0, deferred.size());
        } else {
            // ranges of about equal numbers of pairs
            std::vector<std::thread> pool;
            size_t from = 0;
            uint64_t acc = 0;
            const uint64_t share = total_hits / n_thr + 1;
            for (size_t d = 0; d < deferred.size(); d++) {
                acc += qs[deferred[d].query].n_targets;
                if (acc >= share || d + 1 == deferred.size()) {
                    pool.emplace_back(work, from, d + 1);
                    from = d + 1;
                    acc = 0;
                }
            }
            for (std::thread &t : pool) t.join();
        }
        prep_lap("lists sorted by target length, jobs cut (threads)");
        for (size_t d = 0; d < deferred.size(); d++) {
            const Deferred &D = deferred[d];
            PerQuery &P = pq[d];
            if (P.bad) { delete b; return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare: target id out of range"); }
            b->cells += P.sum_cells;
            max_tlen = std::max(max_tlen, P.max_tlen);
            for (size_t z = 0; z < P.jobs.size(); z++) {
                SwJob j = P.jobs[z];
                if (D.multi) j.shape |= n_multi++ << 8;
                jobs.push_back(j);
                job_cells.push_back(P.cells[z]);
            }
            const uint32_t n_t = qs[D.query].n_targets;
            if (D.multi && mode >= MMGPU_SW_START && n_t)
                add_rev_jobs(D.query, D.hit_cursor, n_t, D.shape, (uint64_t)qs[D.query].qlen * (P.rev_mid_len + 1));
        }
    }
    prep_lap("jobs joined");
    b->pairs = total_hits;
    b->h_qoff = qoff;
    b->from_pf = pf != nullptr;
    b->pf_stride = pf_stride;
    b->slot_stride = pf_stride;

    hipStream_t s = c->stream;
    std::vector<int8_t> mat(par->mat, par->mat + par->alphabet * par->alphabet);
#define B_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { delete b; return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    // the fused hand-over (pf): the stream still holds the prefilter batch - the uploads go through pinned staging so that this
    // thread is not parked behind it (mmgpu_ctx::pinned); sized here, once, for everything uploaded below
    void *pin = nullptr;
    size_t pin_cap = 0, pin_used = 0;
    if (pf) {
        size_t need = upload_pinned_need(qres.size()) + upload_pinned_need(qcb.size()) + upload_pinned_need(qoff.size() * 4) +
                      upload_pinned_need(qbias.size() * 4) + upload_pinned_need(qminstart.size() * 4) + upload_pinned_need(mat.size()) +
                      upload_pinned_need((jobs.size() + rev_jobs.size()) * sizeof(SwJob)) + upload_pinned_need(qprof.size()) +
                      upload_pinned_need(qprof_off.size() * 4);
        if (need > c->pinned_cap) {      // (nothing of an earlier batch is in flight from it: every prepare ends with the stream drained)
            if (c->pinned) (void)hipHostFree(c->pinned);
            c->pinned = nullptr;
            c->pinned_cap = 0;
            if (hipHostMalloc(&c->pinned, need + need / 4, hipHostMallocDefault) == hipSuccess) c->pinned_cap = need + need / 4;
            else (void)hipGetLastError();   // pageable copies then
        }
        pin = c->pinned;
        pin_cap = c->pinned_cap;
    }
#define UPLOAD(buf, vec) upload_pinned(buf, vec, s, pin, pin_cap, pin_used)
    B_TRY(UPLOAD(b->d_qres, qres));
    B_TRY(UPLOAD(b->d_qcb, qcb));
    B_TRY(UPLOAD(b->d_qoff, qoff));
    B_TRY(UPLOAD(b->d_qbias, qbias));
    B_TRY(UPLOAD(b->d_qminstart, qminstart));
    if (b->any_profile) {
        B_TRY(UPLOAD(b->d_qprof, qprof));
        B_TRY(UPLOAD(b->d_qprof_off, qprof_off));
    }
    if (pf) {
        B_TRY(b->d_hit_target.alloc(std::max<size_t>((size_t)total_hits, 1) * 4));
        B_TRY(b->d_hit_out.alloc(std::max<size_t>((size_t)total_hits, 1) * 4));
        B_TRY(b->d_stats.alloc(SW_FROM_PF_STAT_SLOTS * 24));      // (cells, pairs, longest target) x slots: sw_from_pf_kernel
        B_TRY(hipMemsetAsync(b->d_stats.p, 0, SW_FROM_PF_STAT_SLOTS * 24, s));
        B_TRY(b->d_pf_counts.alloc(std::max<size_t>(nq, 1) * 4));
        B_TRY(b->d_slot_target.alloc(std::max<size_t>((size_t)total_hits, 1) * 4));
    } else {
        B_TRY(upload(b->d_hit_target, hit_target, s));
        B_TRY(upload(b->d_hit_out, hit_out, s));
    }
    B_TRY(UPLOAD(b->d_mat, mat));
    B_TRY(b->d_out.alloc(std::max<size_t>((size_t)total_hits, 1) * sizeof(mmgpu_sw_hit)));
    std::vector<SwJob> sorted(jobs.size());   // uploaded asynchronously: must live until the stream is drained below
    {
        // longest job first: the dispatcher hands out workgroups in blockIdx order, so the tail is the shortest jobs
        std::vector<uint32_t> ord(jobs.size());
        std::iota(ord.begin(), ord.end(), 0u);
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t bb) {
            const int ga = sw_shape_group(jobs[a].shape & 0xFFu), gb = sw_shape_group(jobs[bb].shape & 0xFFu);
            return ga != gb ? ga < gb : job_cells[a] > job_cells[bb];
        });
        for (size_t z = 0; z < ord.size(); z++) {
            sorted[z] = jobs[ord[z]];
            b->group_begin[sw_shape_group(sorted[z].shape & 0xFFu) + 1]++;
        }
        for (int g = 0; g < SW_GROUPS; g++) b->group_begin[g + 1] += b->group_begin[g];
        b->n_jobs = (uint32_t)sorted.size();
        b->n_multi_jobs = n_multi;
        {
            std::vector<uint32_t> ro(rev_jobs.size());
            std::iota(ro.begin(), ro.end(), 0u);
            std::stable_sort(ro.begin(), ro.end(), [&](uint32_t a, uint32_t bb) { return rev_cells[a] > rev_cells[bb]; });
            for (uint32_t z : ro) sorted.push_back(rev_jobs[z]);
            b->n_rev_jobs = (uint32_t)rev_jobs.size();
        }
        B_TRY(UPLOAD(b->d_jobs, sorted));
    }
#undef UPLOAD
    prep_lap("uploads enqueued, jobs ordered");
    // column scratch of the multi-tile jobs: a pool with one slot per workgroup that can be resident at once (not one
    // per job: a batch of long queries against one very long target would ask for 100+ GB), sized by the longest
    // target any list holds
    auto alloc_scratch = [&](uint32_t longest) -> hipError_t {
        b->scratch_cols = longest + 16;
        b->scratch_slots = std::min<uint32_t>(std::max<uint32_t>(std::max<uint32_t>(n_multi, (uint32_t)rev_jobs.size()), 1),
                                              sw_multi_resident_blocks(b->group_lds[SW_GROUPS - 1], mode >= MMGPU_SW_START, c->compute_units));
        hipError_t e = b->d_scratch.alloc((size_t)b->scratch_slots * 4 * 4 * 2 * (size_t)b->scratch_cols * sizeof(uint2));
        if (e != hipSuccess) return e;
        e = b->d_scratch_busy.alloc((size_t)b->scratch_slots * 4);
        if (e != hipSuccess) return e;
        return hipMemsetAsync(b->d_scratch_busy.p, 0, (size_t)b->scratch_slots * 4, s);
    };
    if (any_multi && !pf) B_TRY(alloc_scratch(max_tlen));
    if (pf) {
        B_TRY(hipMemsetAsync(b->d_out.p, 0, std::max<size_t>((size_t)total_hits, 1) * sizeof(mmgpu_sw_hit), s));
        SwFromPfArgs F;
        F.pf_hits = pf_hits;
        F.pf_stride = pf_stride;
        F.hit_count = pf_counts;
        F.stride = pf_stride;
        F.q_off = b->d_qoff.as<uint32_t>();
        F.t_len = c->db.len;
        F.hit_target = b->d_hit_target.as<uint32_t>();
        F.hit_out = b->d_hit_out.as<uint32_t>();
        F.cells = b->d_stats.as<unsigned long long>();
        F.pairs = b->d_stats.as<unsigned long long>() + 1;
        F.count_copy = b->d_pf_counts.as<uint32_t>();
        F.slot_target = b->d_slot_target.as<uint32_t>();
        B_TRY(launch_sw_from_pf(F, nq, s));
    }
    B_TRY(hipStreamSynchronize(s));   // the host vectors above die with this scope
    if (pf) {
        unsigned long long st[3] = {0, 0, 0}, slots[SW_FROM_PF_STAT_SLOTS * 3];
        B_TRY(hipMemcpy(slots, b->d_stats.p, sizeof(slots), hipMemcpyDeviceToHost));
        for (int z = 0; z < SW_FROM_PF_STAT_SLOTS; z++) {
            st[0] += slots[z * 3];
            st[1] += slots[z * 3 + 1];
            st[2] = std::max(st[2], slots[z * 3 + 2]);
        }
        b->cells = st[0];
        b->valid_pairs = st[1];
        if (any_multi) {
            B_TRY(alloc_scratch((uint32_t)st[2]));
            B_TRY(hipStreamSynchronize(s));
        }
    }
#undef B_TRY
    prep_lap("scratch + stream drained");
    *out = b;
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_prepare(mmgpu_ctx *c, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq,
                                int mode, mmgpu_sw_batch_t **out) {
    return sw_prepare_impl(c, par, qs, nq, mode, nullptr, out);
}

extern "C" int mmgpu_sw_prepare_from_pf(mmgpu_ctx *c, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq,
                                        int mode, mmgpu_pf_batch_t *pf, mmgpu_sw_batch_t **out) {
    if (!pf) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare_from_pf: NULL prefilter batch");
    DeviceLists L;
    uint32_t pf_nq = 0;
    if (!pf_batch_device_lists(pf, &L.hits, &L.counts, &L.stride, &pf_nq))
        return fail(MMGPU_ERR_STATE, "mmgpu_sw_prepare_from_pf: the prefilter batch was never run, or is an exchange batch of a sharded run");
    if (pf_nq != nq) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare_from_pf: query count differs from the prefilter batch");
    return sw_prepare_impl(c, par, qs, nq, mode, &L, out);
}

extern "C" int mmgpu_sw_prepare_from_lists(mmgpu_ctx *c, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq, int mode,
                                           const void *d_hits, const void *d_counts, uint32_t stride, mmgpu_sw_batch_t **out) {
    if ((!d_hits || !d_counts) && nq) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare_from_lists: NULL list pointers");
    if (stride == 0) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare_from_lists: stride must be >= 1");
    DeviceLists L;
    L.hits = (const mmgpu_pf_hit *)d_hits;
    L.counts = (const uint32_t *)d_counts;
    L.stride = stride;
    return sw_prepare_impl(c, par, qs, nq, mode, &L, out);
}

// ---- multi-GPU: every rank aligns the pairs of the merged lists whose target it holds; the records are gathered over the
// communicator in merged-list order ("run each pair on the GPU owning t; gather mmgpu_sw_hits per query") ----
extern "C" int mmgpu_sw_prepare_owned(mmgpu_ctx *c, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq, int mode,
                                      mmgpu_pf_batch_t *merged, mmgpu_sw_batch_t **out) {
    if (!c || !merged || !out) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare_owned: NULL argument");
    if (!c->shard.on) return fail(MMGPU_ERR_STATE, "mmgpu_sw_prepare_owned: no shard set (mmgpu_pf_set_shard)");
    const mmgpu_pf_hit *mh = nullptr;
    const uint32_t *mc = nullptr;
    uint32_t stride = 0, mnq = 0;
    if (!mmgpu::pf_batch_merged_lists(merged, &mh, &mc, &stride, &mnq)) return fail(MMGPU_ERR_STATE, "mmgpu_sw_prepare_owned: the prefilter batch holds no merged lists (mmgpu_pf_exchange_merge first)");
    if (mnq != nq) return fail(MMGPU_ERR_ARG, "mmgpu_sw_prepare_owned: query count differs from the prefilter batch");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf lh, lc, ls;
    lh.bind(c->cache); lc.bind(c->cache); ls.bind(c->cache);
    HIP_TRY(lh.alloc(std::max<size_t>((size_t)nq * stride, 1) * sizeof(mmgpu_pf_hit)));
    HIP_TRY(lc.alloc(std::max<size_t>(nq, 1) * 4));
    HIP_TRY(ls.alloc(std::max<size_t>((size_t)nq * stride, 1) * 4));
    PfLocalizeArgs A;
    A.hits = mh;
    A.counts = mc;
    A.nq = nq;
    A.stride = stride;
    A.shard = c->shard.shard;
    A.shard_of = c->shard.d_shard_of.as<uint32_t>();
    A.local_id = c->shard.d_local_id.as<uint32_t>();
    A.local_hits = lh.as<mmgpu_pf_hit>();
    A.local_counts = lc.as<uint32_t>();
    A.local_slot = ls.as<uint32_t>();
    HIP_TRY(launch_pf_localize(A, c->stream));
    DeviceLists L;
    L.hits = lh.as<mmgpu_pf_hit>();
    L.counts = lc.as<uint32_t>();
    L.stride = stride;
    if (int e = sw_prepare_impl(c, par, qs, nq, mode, &L, out)) return e;
    mmgpu_sw_batch_t *b = *out;
    b->owned = true;
    b->o_stride = stride;
    b->o_lhits = std::move(lh);
    b->o_lcounts = std::move(lc);
    b->o_lslot = std::move(ls);
    return MMGPU_OK;
}

namespace mmgpu {

// phase 1: pack the owned records (compacted with a device counter) and name the two all-gathers of the step.  The send
// buffer holds 1.5 x the even share of the slots (+ 4096): the deal by length bucket gives every rank its share to within
// a few percent; a rank that packs more reports it in its counter and the scatter phase raises the batch's overflow status.
int sw_gather_begin(mmgpu_ctx *c, mmgpu_sw_batch_t *b, int n_ranks, XchgBlock blocks[2]) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_gather_owned: NULL argument");
    if (!b->owned || !b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_gather_owned: not a batch of mmgpu_sw_prepare_owned that has been run");
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t slots = (uint64_t)b->n_queries * b->o_stride;
    static const char *dense = getenv("MMGPU_SW_GATHER_DENSE");      // every rank may own everything (no overflow possible)
    const uint64_t cap64 = (dense || b->o_dense || n_ranks == 1) ? slots : std::min<uint64_t>(slots, slots / (uint64_t)n_ranks * 3 / 2 + 4096);
    const uint32_t cap = (uint32_t)std::max<uint64_t>(cap64, 1);
    if (b->o_cap != cap || b->o_ranks != n_ranks) {
        for (DevBuf *d : {&b->o_send, &b->o_counter, &b->o_recv, &b->o_recv_counters, &b->o_full, &b->o_status}) d->bind(c->cache);
        HIP_TRY(b->o_send.alloc((size_t)cap * sizeof(SwOwnedRec)));
        HIP_TRY(b->o_counter.alloc(8));
        HIP_TRY(b->o_recv.alloc((size_t)cap * sizeof(SwOwnedRec) * n_ranks));
        HIP_TRY(b->o_recv_counters.alloc((size_t)8 * n_ranks));
        HIP_TRY(b->o_full.alloc(std::max<size_t>(slots, 1) * sizeof(mmgpu_sw_hit)));
        HIP_TRY(b->o_status.alloc(8));
        b->o_cap = cap;
        b->o_ranks = n_ranks;
    }
    HIP_TRY(hipMemsetAsync(b->o_counter.p, 0, 8, c->stream));
    SwOwnedPackArgs P;
    P.res = b->d_out.as<mmgpu_sw_hit>();
    P.local_counts = b->o_lcounts.as<uint32_t>();
    P.local_slot = b->o_lslot.as<uint32_t>();
    P.nq = b->n_queries;
    P.stride = b->o_stride;
    P.send = b->o_send.as<SwOwnedRec>();
    P.cap = cap;
    P.counter = b->o_counter.as<uint32_t>();
    HIP_TRY(launch_sw_owned_pack(P, c->stream));
    blocks[0] = XchgBlock{b->o_send.p, b->o_recv.p, (size_t)cap * sizeof(SwOwnedRec)};
    blocks[1] = XchgBlock{b->o_counter.p, b->o_recv_counters.p, 8};
    return MMGPU_OK;
}

// after phase 3: did a rank pack more than its send buffer holds (hits clustered in one shard)?  Every rank sees all counters, so
// every rank answers the same; the caller then repeats the three phases with buffers that hold every slot (o_dense).
int sw_gather_overflowed(mmgpu_ctx *c, mmgpu_sw_batch_t *b, bool *overflowed) {
    HIP_TRY(hipSetDevice(c->device));
    uint32_t st[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(st, b->o_status.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *overflowed = st[1] != 0 && !b->o_dense;
    if (*overflowed) b->o_dense = true;
    return MMGPU_OK;
}

// phase 3: scatter every rank's records into the merged-list order
int sw_gather_finish(mmgpu_ctx *c, mmgpu_sw_batch_t *b, int n_ranks) {
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t slots = (uint64_t)b->n_queries * b->o_stride;
    HIP_TRY(hipMemsetAsync(b->o_full.p, 0, std::max<size_t>(slots, 1) * sizeof(mmgpu_sw_hit), c->stream));
    HIP_TRY(hipMemsetAsync(b->o_status.p, 0, 8, c->stream));
    SwOwnedScatterArgs S;
    S.recv = b->o_recv.as<SwOwnedRec>();
    S.counters = b->o_recv_counters.as<uint32_t>();
    S.n_ranks = (uint32_t)n_ranks;
    S.cap = b->o_cap;
    S.n_slots = (uint32_t)slots;
    S.full = b->o_full.as<mmgpu_sw_hit>();
    S.status = b->o_status.as<uint32_t>();
    HIP_TRY(launch_sw_owned_scatter(S, c->stream));
    return MMGPU_OK;
}

}  // namespace mmgpu

extern "C" int mmgpu_sw_gather_owned(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const void **d_full, const void **d_status) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_gather_owned: NULL argument");
    const int n = c->comm ? c->comm->n_ranks : 1;
    for (int attempt = 0; attempt < 2; attempt++) {
        mmgpu::XchgBlock blk[2];
        if (int e = mmgpu::sw_gather_begin(c, b, n, blk)) return e;
        for (int k = 0; k < 2; k++)
            if (int e = mmgpu::comm_allgather(c, blk[k].send, blk[k].recv, blk[k].bytes)) return e;
        if (int e = mmgpu::sw_gather_finish(c, b, n)) return e;
        // uneven shards: one more round with send buffers that hold every slot (all ranks decide alike: they see all counters)
        bool again = false;
        if (n > 1 && attempt == 0)
            if (int e = mmgpu::sw_gather_overflowed(c, b, &again)) return e;
        if (!again) break;
    }
    if (d_full) *d_full = b->o_full.p;
    if (d_status) *d_status = b->o_status.p;
    return MMGPU_OK;
}

// host copy of the gathered records (synchronises the context's stream); MMGPU_ERR_STATE if a rank's send buffer overflowed
// (cannot happen after mmgpu_sw_gather_owned's own second round; kept as the check it is)
extern "C" int mmgpu_sw_fetch_owned(mmgpu_ctx *c, mmgpu_sw_batch_t *b, mmgpu_sw_hit *out, uint32_t *records) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_fetch_owned: NULL argument");
    if (!b->owned || !b->o_full.p) return fail(MMGPU_ERR_STATE, "mmgpu_sw_fetch_owned: nothing gathered (mmgpu_sw_gather_owned first)");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t st[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(st, b->o_status.p, 8, hipMemcpyDeviceToHost, c->stream));
    if (out) HIP_TRY(hipMemcpyAsync(out, b->o_full.p, (size_t)b->n_queries * b->o_stride * sizeof(mmgpu_sw_hit), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (records) *records = st[0];
    if (st[1]) return fail(MMGPU_ERR_STATE, "mmgpu_sw_fetch_owned: the send buffer of a rank overflowed (uneven shards); rerun the gather with MMGPU_SW_GATHER_DENSE=1");
    return MMGPU_OK;
}

// The kernel groups of a batch, forked from / joined to the context's stream.  rev_only: the forward results are in d_out, only
// the reverse scan of the pairs flagged in rev_force runs (mmgpu_sw_reverse_pairs).
static int sw_launch_groups(mmgpu_ctx *c, mmgpu_sw_batch_t *b, bool rev_only, const uint8_t *rev_force) {
    // one launch per kernel group (jobs longest first); the groups run concurrently on side streams forked from /
    // joined to the context's stream.  With start positions asked for, a workgroup runs the reverse scan of its
    // pairs right after their forward scan (it reads the forward results of its own pairs only).
    if (!c->fork) {
        // the group with the long multi-tile jobs gets the highest stream priority: its forward + reverse chain is
        // the critical path, the other groups fill the CUs it leaves
        int prio_low = 0, prio_high = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));   // numerically lower = higher priority
        for (int g = 0; g < SW_GROUPS; g++) {
            const int prio = g == SW_GROUPS - 1 ? prio_high : (g == 0 ? prio_low : (prio_low + prio_high) / 2);
            HIP_TRY(hipStreamCreateWithPriority(&c->side[g], hipStreamNonBlocking, prio));
        }
        HIP_TRY(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
        for (auto &e : c->join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(c->fork, c->stream));
    for (int g = 0; g < SW_GROUPS; g++) HIP_TRY(hipStreamWaitEvent(c->side[g], c->fork, 0));
    for (int g = SW_GROUPS - 1; g >= 0; g--) {
        hipStream_t st = c->side[g];
        {
            SwLaunch L;
            L.jobs = b->d_jobs.as<SwJob>() + b->group_begin[g];
            L.n_jobs = b->group_begin[g + 1] - b->group_begin[g];
            L.q_res = b->d_qres.as<uint8_t>();
            L.q_cb = b->d_qcb.as<int8_t>();
            L.q_off = b->d_qoff.as<uint32_t>();
            L.q_bias = b->d_qbias.as<int32_t>();
            L.q_minstart = b->d_qminstart.as<int32_t>();
            L.q_prof = b->any_profile ? b->d_qprof.as<int8_t>() : nullptr;
            L.q_prof_off = b->any_profile ? b->d_qprof_off.as<uint32_t>() : nullptr;
            L.t_res = c->db.res;
            L.t_off4 = c->db.off4;
            L.t_len = c->db.len;
            L.hit_target = b->d_hit_target.as<uint32_t>();
            L.hit_out = b->d_hit_out.as<uint32_t>();
            L.out = b->d_out.as<mmgpu_sw_hit>();
            L.mat = b->d_mat.as<int8_t>();
            L.alphabet = b->alphabet;
            L.gap_open = b->gap_open;
            L.gap_extend = b->gap_extend;
            L.q_hit_count = b->from_pf ? b->d_pf_counts.as<uint32_t>() : nullptr;
            L.hit_stride = b->slot_stride;
            L.scratch = b->d_scratch.as<uint2>();
            L.scratch_cols = b->scratch_cols;
            L.scratch_busy = b->d_scratch_busy.as<uint32_t>();
            L.scratch_slots = std::max<uint32_t>(b->scratch_slots, 1);
            L.rev_mode = rev_only ? 2 : (b->mode == MMGPU_SW_START_NOT_WORD ? 1 : 0);
            L.rev_only = rev_only ? 1 : 0;
            L.rev_force = rev_force;
            HIP_TRY(launch_sw(L, g, b->group_lds[g], b->mode >= MMGPU_SW_START, st));
            if (g == SW_GROUPS - 1 && b->n_rev_jobs && b->mode >= MMGPU_SW_START) {   // reads the forward results of this stream's kernel
                SwLaunch Rv = L;
                Rv.jobs = b->d_jobs.as<SwJob>() + b->n_jobs;
                Rv.n_jobs = b->n_rev_jobs;
                HIP_TRY(launch_sw_rev_multi(Rv, b->group_lds[g], st));
            }
            if (getenv("MMGPU_TRACE")) {   // debugging aid: run the groups one at a time and say which one is in flight
                fprintf(stderr, "[sw_run] group %d jobs %u lds %zu both %d from_pf %d scratch_cols %u\n", g, L.n_jobs, b->group_lds[g], (int)(b->mode >= MMGPU_SW_START), (int)b->from_pf, b->scratch_cols);
                fflush(stderr);
                const auto t0 = std::chrono::steady_clock::now();
                hipError_t e = hipStreamSynchronize(st);
                fprintf(stderr, "[sw_run] group %d done after %.2f ms: %s\n", g,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), hipGetErrorString(e));
                fflush(stderr);
            }
        }
    }
    for (int k = 0; k < SW_GROUPS; k++) {
        HIP_TRY(hipEventRecord(c->join[k], c->side[k]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->join[k], 0));
    }
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_run(mmgpu_ctx *c, mmgpu_sw_batch_t *b) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_run: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (b->events.size() < 256) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        b->events.push_back(std::make_pair(ev0, ev1));
        HIP_TRY(hipEventRecord(ev0, c->stream));
    }
    if (int e = sw_launch_groups(c, b, false, nullptr)) return e;
    if (ev1) HIP_TRY(hipEventRecord(ev1, c->stream));
    b->ran = true;
    b->h_res_valid = false;
    return MMGPU_OK;
}

// Start positions after the fact for the pairs the caller names (MMGPU_SW_START_NOT_WORD batches: the int16-range hits the block
// aligner declined, StripedSmithWaterman.cpp:873-882 "Block alignment failed" -> alignStartPosBacktrace): the reverse scan of
// exactly those pairs, the forward results staying as they are.
extern "C" int mmgpu_sw_reverse_pairs(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *pair_index, uint32_t n, mmgpu_sw_hit *out) {
    if (!c || !b || (!pair_index && n)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_reverse_pairs: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_reverse_pairs: batch was never run");
    if (b->mode < MMGPU_SW_START) return fail(MMGPU_ERR_STATE, "mmgpu_sw_reverse_pairs: the batch was prepared without a start-position mode");
    if (b->owned) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_reverse_pairs: not for batches of owned pairs (sharded runs)");
    if (n == 0) return MMGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint8_t> flags((size_t)b->pairs, (uint8_t)0);
    for (uint32_t k = 0; k < n; k++) {
        if (pair_index[k] >= b->pairs) return fail(MMGPU_ERR_ARG, "mmgpu_sw_reverse_pairs: pair index out of range");
        flags[pair_index[k]] = 1;
    }
    DevBuf d_flags;
    d_flags.bind(c->cache);
    HIP_TRY(d_flags.alloc(flags.size()));
    HIP_TRY(hipMemcpyAsync(d_flags.p, flags.data(), flags.size(), hipMemcpyHostToDevice, c->stream));
    if (int e = sw_launch_groups(c, b, true, d_flags.as<uint8_t>())) return e;
    if (out)
        for (uint32_t k = 0; k < n; k++)
            HIP_TRY(hipMemcpyAsync(out + k, b->d_out.as<mmgpu_sw_hit>() + pair_index[k], sizeof(mmgpu_sw_hit), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));      // (the flags die with this scope)
    if (b->h_res_valid)      // the host copy mmgpu_sw_traceback / mmgpu_sw_block_backtrace read
        for (uint32_t k = 0; k < n; k++)
            HIP_TRY(hipMemcpy(&b->h_res[pair_index[k]], b->d_out.as<mmgpu_sw_hit>() + pair_index[k], sizeof(mmgpu_sw_hit), hipMemcpyDeviceToHost));
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_fetch(mmgpu_ctx *c, mmgpu_sw_batch_t *b, mmgpu_sw_hit *out) {
    if (!c || !b || (!out && b->pairs)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_fetch: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_fetch: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    if (b->pairs)
        HIP_TRY(hipMemcpyAsync(out, b->d_out.p, (size_t)b->pairs * sizeof(mmgpu_sw_hit), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_fetch_device(mmgpu_ctx *c, mmgpu_sw_batch_t *b, void *d_out) {
    if (!c || !b || (!d_out && b->pairs)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_fetch_device: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_fetch_device: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    if (b->pairs)
        HIP_TRY(hipMemcpyAsync(d_out, b->d_out.p, (size_t)b->pairs * sizeof(mmgpu_sw_hit), hipMemcpyDeviceToDevice, c->stream));
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_batch_stats(mmgpu_sw_batch_t *b, uint64_t *cells, uint64_t *pairs) {
    if (!b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_batch_stats: NULL argument");
    if (cells) *cells = b->cells;
    if (pairs) *pairs = b->from_pf ? b->valid_pairs : b->pairs;
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_last_kernel_ms(mmgpu_ctx *c, mmgpu_sw_batch_t *b, float *ms) {
    if (!c || !b || !ms) return fail(MMGPU_ERR_ARG, "mmgpu_sw_last_kernel_ms: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_last_kernel_ms: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(b->events.back().second));
    HIP_TRY(hipEventElapsedTime(ms, b->events.back().first, b->events.back().second));
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_kernel_ms_mean(mmgpu_ctx *c, mmgpu_sw_batch_t *b, uint32_t last_n, float *ms, uint32_t *n_used) {
    if (!c || !b || !ms) return fail(MMGPU_ERR_ARG, "mmgpu_sw_kernel_ms_mean: NULL argument");
    if (!b->ran || b->events.empty()) return fail(MMGPU_ERR_STATE, "mmgpu_sw_kernel_ms_mean: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = std::min<size_t>(last_n ? last_n : b->events.size(), b->events.size());
    double sum = 0;
    for (size_t k = b->events.size() - n; k < b->events.size(); k++) {
        float t = 0;
        HIP_TRY(hipEventSynchronize(b->events[k].second));
        HIP_TRY(hipEventElapsedTime(&t, b->events[k].first, b->events[k].second));
        sum += t;
    }
    *ms = (float)(sum / (double)n);
    if (n_used) *n_used = (uint32_t)n;
    return MMGPU_OK;
}

extern "C" void mmgpu_sw_free(mmgpu_ctx *c, mmgpu_sw_batch_t *b) {
    if (!b) return;
    if (c) (void)hipSetDevice(c->device);
    for (auto &e : b->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete b;
}

extern "C" int mmgpu_sw_batch(mmgpu_ctx *c, const mmgpu_sw_params *par, const mmgpu_sw_query *qs, uint32_t nq, int mode,
                              mmgpu_sw_hit *out) {
    mmgpu_sw_batch_t *b = nullptr;
    int rc = mmgpu_sw_prepare(c, par, qs, nq, mode, &b);
    if (rc != MMGPU_OK) return rc;
    rc = mmgpu_sw_run(c, b);
    if (rc == MMGPU_OK) rc = mmgpu_sw_fetch(c, b, out);
    mmgpu_sw_free(c, b);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// backtrace
// ---------------------------------------------------------------------------------------------------------
// ---- a15: the block aligner's start position / backtrace for int16-range hits (block_kernel.hip) ----
struct BlockAuto { uint32_t selected = 0, ok = 0, declined = 0, too_large = 0; };      // mmgpu_sw_block_starts: what the device selected / answered

static int block_backtrace(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *pair_index, uint32_t n, mmgpu_sw_block *out,
                           char *bt, size_t bt_cap, size_t *bt_used, uint32_t *growth, uint32_t growth_cap, BlockAuto *au = nullptr) {
    if (!c || !b || (!au && ((!pair_index && n) || (!out && n)))) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_backtrace: NULL argument");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_block_backtrace: batch was never run");
    if (b->alphabet > 26) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_block_backtrace: alphabet above 26 letters");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (b->mode < MMGPU_SW_START && !b->from_pf && b->h_out_target.empty())
        return fail(MMGPU_ERR_STATE, "mmgpu_sw_block_backtrace: the batch keeps no slot -> target map (prepare it with MMGPU_SW_START)");
    static const bool trace_on = getenv("MMGPU_TRACE") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now_s();
    auto lap = [&](const char *what) {
        if (!trace_on) return;
        const double t = now_s();
        fprintf(stderr, "[mmgpu block aligner] %s %.3f s\n", what, t - t_mark);
        t_mark = t;
    };
    std::vector<BlockJob> jobs;
    std::vector<uint64_t> bt_off;
    uint64_t off = 0, longest = 0;
    DevBuf d_out, d_sel_jobs, d_sel_pairs, d_sel_cnt, d_flags;
    for (DevBuf *d : {&d_out, &d_sel_jobs, &d_sel_pairs, &d_sel_cnt, &d_flags}) d->bind(c->cache);
    std::vector<mmgpu_sw_block> out_auto;
    if (au) {
        // ---- the device picks the pairs (block_select.hip): int16-range hits that pass the query's start-score threshold ----
        if (!b->d_qout_off.p) {
            HIP_TRY(b->d_qout_off.alloc(b->h_qout_off.size() * 4));
            HIP_TRY(hipMemcpyAsync(b->d_qout_off.p, b->h_qout_off.data(), b->h_qout_off.size() * 4, hipMemcpyHostToDevice, s));
        }
        if (!b->from_pf && !b->d_out_target.p) {
            HIP_TRY(b->d_out_target.alloc(std::max<size_t>(b->h_out_target.size(), 1) * 4));
            HIP_TRY(hipMemcpyAsync(b->d_out_target.p, b->h_out_target.data(), b->h_out_target.size() * 4, hipMemcpyHostToDevice, s));
        }
        HIP_TRY(d_sel_cnt.alloc(32));
        HIP_TRY(hipMemsetAsync(d_sel_cnt.p, 0, 32, s));
        BlockSelectArgs A;
        A.res = b->d_out.as<mmgpu_sw_hit>();
        A.pairs = (uint32_t)b->pairs;
        A.qout_off = b->d_qout_off.as<uint32_t>();
        A.n_queries = (uint32_t)(b->h_qout_off.size() - 1);
        A.q_minstart = b->d_qminstart.as<int32_t>();
        A.slot_target = b->from_pf ? b->d_slot_target.as<uint32_t>() : b->d_out_target.as<uint32_t>();
        A.jobs = nullptr; A.pair_of_slot = nullptr; A.blk = nullptr;
        A.count = d_sel_cnt.as<uint32_t>();
        A.cap = 0;      // first pass: count only
        HIP_TRY(launch_block_select(A, s));
        uint32_t cnt = 0;
        HIP_TRY(hipMemcpyAsync(&cnt, d_sel_cnt.p, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        n = cnt;
        au->selected = n;
        if (n == 0) return MMGPU_OK;
        HIP_TRY(d_sel_jobs.alloc((size_t)n * sizeof(BlockJob)));
        HIP_TRY(d_sel_pairs.alloc((size_t)n * 4));
        HIP_TRY(d_out.alloc((size_t)n * sizeof(mmgpu_sw_block)));
        HIP_TRY(hipMemsetAsync(d_sel_cnt.p, 0, 4, s));
        A.jobs = d_sel_jobs.as<BlockJob>();
        A.pair_of_slot = d_sel_pairs.as<uint32_t>();
        A.blk = d_out.as<mmgpu_sw_block>();
        A.cap = n;
        HIP_TRY(launch_block_select(A, s));
        jobs.resize(n);
        // the selected jobs come back once and all n answers after every launch (block4_collect): through pinned memory where the
        // context's staging area can be had (mmgpu_ctx::pinned; nothing else uses it between a batch's preparation and its free) -
        // the copies of pageable memory were ~5 ms of a 45 ms call
        const size_t out_bytes = (size_t)n * sizeof(mmgpu_sw_block), jobs_bytes = (size_t)n * sizeof(BlockJob);
        const size_t pin_need = upload_pinned_need(out_bytes) + upload_pinned_need(jobs_bytes);
        if (pin_need > c->pinned_cap) {
            if (c->pinned) (void)hipHostFree(c->pinned);
            c->pinned = nullptr;
            c->pinned_cap = 0;
            if (hipHostMalloc(&c->pinned, pin_need + pin_need / 4, hipHostMallocDefault) == hipSuccess) c->pinned_cap = pin_need + pin_need / 4;
            else (void)hipGetLastError();
        }
        if (c->pinned_cap >= pin_need) {
            BlockJob *pj = reinterpret_cast<BlockJob *>(static_cast<char *>(c->pinned) + upload_pinned_need(out_bytes));
            HIP_TRY(hipMemcpyAsync(pj, d_sel_jobs.p, jobs_bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            memcpy(jobs.data(), pj, jobs_bytes);
            out = static_cast<mmgpu_sw_block *>(c->pinned);
        } else {
            HIP_TRY(hipMemcpyAsync(jobs.data(), d_sel_jobs.p, jobs_bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            out_auto.resize(n);
            out = out_auto.data();
        }
        // (string offsets as in the other form: block_kernel.hip - profile queries, pairs beyond the pool - walks back whatever is asked)
        bt_off.assign(n, 0);
        for (const BlockJob &j : jobs) {
            const uint64_t len = (uint64_t)j.q_end + 1 + (uint64_t)j.t_end + 1;
            bt_off[j.slot] = off;
            off += (len + 1 + 3) & ~3ull;
            longest = std::max(longest, len);
        }
        lap("device selection + job download");
    } else {
        if (!b->h_res_valid) {
            b->h_res.resize((size_t)b->pairs);
            if (b->pairs) HIP_TRY(hipMemcpyAsync(b->h_res.data(), b->d_out.p, (size_t)b->pairs * sizeof(mmgpu_sw_hit), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (b->from_pf) {
                b->h_slot_target.resize((size_t)b->pairs);
                if (b->pairs) HIP_TRY(hipMemcpy(b->h_slot_target.data(), b->d_slot_target.p, (size_t)b->pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
            }
            b->h_res_valid = true;
        }
        bt_off.assign(std::max<uint32_t>(n, 1), 0);
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t p = pair_index[k];
            if (p >= b->pairs) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_backtrace: pair index out of range");
            const mmgpu_sw_hit &h = b->h_res[p];
            out[k].q_start = -1; out[k].t_start = -1; out[k].ident = 0; out[k].bt_len = 0; out[k].bt_off = off; out[k].reserved = 0;
            bt_off[k] = off;
            const uint32_t q = (uint32_t)(std::upper_bound(b->h_qout_off.begin(), b->h_qout_off.end(), p) - b->h_qout_off.begin() - 1);
            // (profile queries run the same kernel with the query's score rows in place of matrix + bias: block_kernel.hip, BkSeq::prof)
            if (h.score <= 0 || h.word != 1 || h.t_end < 0) {
                out[k].status = MMGPU_BLOCK_NOT_WORD;
                continue;
            }
            out[k].status = MMGPU_BLOCK_TOO_LARGE;    // overwritten by the kernel
            BlockJob j;
            j.query = q;
            j.target = b->from_pf ? b->h_slot_target[p] : b->h_out_target[p];
            j.score = h.score; j.q_end = h.q_end; j.t_end = h.t_end;
            j.slot = k;
            jobs.push_back(j);
            const uint64_t len = (uint64_t)h.q_end + 1 + (uint64_t)h.t_end + 1;
            off += (len + 1 + 3) & ~3ull;      // (multiples of four: the walk kernel of block4_kernel.hip stores a string in dwords)
            longest = std::max(longest, len);
        }
        lap("result download + job list");
    }
    if (bt_used) *bt_used = (size_t)off;
    const bool starts_only = au != nullptr || (bt == nullptr && bt_cap == MMGPU_BLOCK_STARTS_ONLY);    // start positions only: no trace, no walk
    const bool no_strings = starts_only || (bt == nullptr && bt_cap == MMGPU_BLOCK_NO_STRINGS);      // start positions / identities / lengths only
    if (!no_strings && (off > bt_cap || (!bt && off))) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_backtrace: bt buffer too small (see *bt_used)");
    if (jobs.empty()) return MMGPU_OK;
    {   // longest pair first, stable: a counting sort over q_end + t_end (both below 65536)
        std::vector<uint32_t> first((size_t)longest + 2, 0u);
        for (const BlockJob &j : jobs) first[(size_t)(j.q_end + j.t_end + 2)]++;      // key = the pair's length
        uint32_t run = 0;
        for (size_t len = (size_t)longest + 1; len-- > 0;) { const uint32_t cnt = first[len]; first[len] = run; run += cnt; }
        std::vector<BlockJob> sorted(jobs.size());
        for (const BlockJob &j : jobs) sorted[first[(size_t)(j.q_end + j.t_end + 2)]++] = j;
        jobs.swap(sorted);
    }
    // the AAMatrix as ssw_init leaves it: new_simple(1, -1) with the substitution matrix written over it (:708,:1469-1474)
    std::vector<int8_t> mat((size_t)b->alphabet * b->alphabet), scores(27 * 32, (int8_t)-128);
    HIP_TRY(hipMemcpy(mat.data(), b->d_mat.p, mat.size(), hipMemcpyDeviceToHost));
    for (int x = 0; x < 26; x++)
        for (int y = 0; y < 26; y++) scores[x * 32 + y] = x == y ? 1 : -1;
    for (int x = 0; x < b->alphabet; x++)
        for (int y = 0; y < b->alphabet; y++) { scores[x * 32 + y] = mat[(size_t)x * b->alphabet + y]; scores[y * 32 + x] = mat[(size_t)x * b->alphabet + y]; }
    // Three tiers (block_kernel.hip), each a launch over what the one before left undecided (TOO_LARGE), all carved out of ONE
    // scratch pool (hipMalloc costs ~40 ms per GB on this platform, so the pool is sized for the usual case, not the worst):
    //   tier 0  blocks <= 512 rows, borders in LDS; slot = block list + TWO trace entries (32 B per 64 rows) per column for the
    //           256th-longest pair - nearly every pair stays at 32 / 64-row blocks; a pair that is longer or grows further
    //           overflows its slot and moves on
    //   tier 1  blocks <= 2048 rows in LDS, slot = the crate's own bound for that size (Trace::new, scan_block.rs:1742-1748)
    //   tier 2  the crate's 4096 rows, borders in the slot
    auto pair_len = [](const BlockJob &j) { return (uint64_t)j.q_end + 1 + (uint64_t)j.t_end + 1; };
    auto slot_size = [](uint64_t len, uint64_t entries_per_col, uint64_t max_rows, bool borders) {
        return (borders ? (uint64_t)8 * BLOCK_REF_MAX_SIZE * 2 : 0ull) + (((len + 64) * 16 + 31) & ~31ull) + entries_per_col * 32 * (len + 2 * max_rows);
    };
    constexpr uint64_t pool_limit = 16384ull << 20;
    // (small calls: slots for the longest pair, everything starts in tier 0)
    uint64_t typical_len = pair_len(jobs[jobs.size() > 1024 ? 255 : 0]);
    DevBuf d_btoff, d_bt, d_scores, d_jobs[3], d_pool[3], d_busy[3];
    for (DevBuf *d : {&d_btoff, &d_bt, &d_scores, &d_jobs[0], &d_jobs[1], &d_jobs[2], &d_pool[0], &d_pool[1], &d_pool[2], &d_busy[0], &d_busy[1], &d_busy[2]})
        d->bind(c->cache);
    if (!au) HIP_TRY(d_out.alloc((size_t)n * sizeof(mmgpu_sw_block)));
    DevBuf d_growth;
    d_growth.bind(c->cache);
    const size_t growth_bytes = growth ? (size_t)n * (1 + 4 * (size_t)growth_cap) * 4 : 0;
    if (growth) {
        HIP_TRY(d_growth.alloc(growth_bytes));
        HIP_TRY(hipMemsetAsync(d_growth.p, 0, growth_bytes, s));
    }
    HIP_TRY(d_btoff.alloc(bt_off.size() * 8));
    HIP_TRY(d_bt.alloc((size_t)off + 16));
    HIP_TRY(d_scores.alloc(scores.size()));
    if (!au) HIP_TRY(hipMemcpyAsync(d_out.p, out, (size_t)n * sizeof(mmgpu_sw_block), hipMemcpyHostToDevice, s));      // (au: block_select_kernel wrote them)
    HIP_TRY(hipMemcpyAsync(d_btoff.p, bt_off.data(), bt_off.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_scores.p, scores.data(), scores.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    lap("buffers + uploads");
    BlockLaunch L;
    L.q_res = b->d_qres.as<uint8_t>();
    L.q_cb = b->d_qcb.as<int8_t>();
    L.q_off = b->d_qoff.as<uint32_t>();
    L.t_res = c->db.res;
    L.t_off4 = c->db.off4;
    L.scores = d_scores.as<int8_t>();
    L.q_prof = b->any_profile ? b->d_qprof.as<int8_t>() : nullptr;
    L.q_prof_off = b->any_profile ? b->d_qprof_off.as<uint32_t>() : nullptr;
    L.alphabet = b->alphabet;
    L.gap_open = -b->gap_open;
    L.gap_extend = -b->gap_extend;
    L.out = d_out.as<mmgpu_sw_block>();
    L.bt_off = d_btoff.as<uint64_t>();
    L.bt = d_bt.as<char>();
    L.growth = growth ? d_growth.as<uint32_t>() : nullptr;
    L.growth_cap = growth_cap;
    const char *first_tier_env = getenv("MMGPU_BLOCK_FIRST_TIER");      // test aid: every pair through block_kernel.hip's tier 0 / 1 / 2
    const int first_tier = first_tier_env ? std::max(0, std::min(2, atoi(first_tier_env))) : 0;
    b->block_pairs_tier[0] = b->block_pairs_tier[1] = b->block_pairs_tier[2] = 0;
    b->block_pairs_fast = 0;
    b->block_pairs_skew = 0;
    // ---- block4_kernel.hip: sequence queries.  Launch 1: four pairs per wavefront, blocks up to 256 rows.  What it answers
    // TOO_LARGE (blocks would grow further, or the trace overflowed the pair's slot) goes through the skewed form - one pair per
    // wavefront, its rows pipelined over the columns - with blocks up to 1024 rows, then the crate's 4096; each launch starts at the
    // minimum block size the one before got to (mmgpu_sw_block::reserved of a TOO_LARGE answer).  With a trace the pairs of a launch
    // run in groups whose block lists + traces fit the pool, each group = fill launch + walk launch on the context's stream ----
    std::vector<BlockJob> slow_jobs;
    if (!first_tier_env) {
        constexpr uint64_t block4_waves = 16;        // wavefronts per CU
        constexpr uint64_t block4_per_res = 48;      // trace bytes per residue of a pair, launch 1
        constexpr uint64_t pool2_limit = 3072ull << 20;
        struct PassBufs {      // what a launch in flight holds: two of them run side by side (the long head, everything else)
            DevBuf j2, cnt, pool, ck;
            hipStream_t st = nullptr;
            std::vector<Block2Job> jobs;
        } PB[2];
        for (PassBufs &x : PB) { x.j2.bind(c->cache); x.cnt.bind(c->cache); x.pool.bind(c->cache); x.ck.bind(c->cache); }
        PB[0].st = s;
        hipStream_t head_stream = nullptr;
        auto drop_head_stream = [&] { if (head_stream) { (void)hipStreamDestroy(head_stream); head_stream = nullptr; } };
        std::vector<uint32_t> resume((size_t)n, 0u);      // per slot: the first minimum block size still to try (bit 16: only the slot was too small)
#define B_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { drop_head_stream(); return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
        // one launch: `todo` (longest first) through form 1 (four pairs per wavefront, 256 rows) or 3 (skewed, one pair per wavefront,
        // 4096 rows), slots of `per_res` trace bytes per residue of the pair; pairs whose slot would not fit the pool go to `left`
        auto block4_launch = [&](PassBufs &B, const std::vector<BlockJob> &todo, int form, uint64_t per_res, uint64_t margin, uint64_t waves_per_cu,
                                 std::vector<BlockJob> &left) -> int {
            std::vector<Block2Job> &j2 = B.jobs;
            j2.clear();
            j2.reserve(todo.size());
            std::vector<uint32_t> group_begin(1, 0u);
            uint64_t pool_used = 0, pool_need = 0;
            const int rows = form == 1 ? BLOCK4_LARGE_SIZE : BLOCK_REF_MAX_SIZE;
            for (const BlockJob &j : todo) {
                Block2Job x;
                x.query = j.query; x.target = j.target; x.score = j.score; x.q_end = j.q_end; x.t_end = j.t_end; x.slot = j.slot;
                x.pool_off = 0; x.pool_bytes = 0; x.pad = resume[j.slot] & 0xFFFFu;
                if (!starts_only) {
                    const uint64_t len = pair_len(j);
                    const uint64_t bytes = (((len + 64) * sizeof(BkBlock) + 31) & ~31ull) + per_res * (len + margin);
                    if (bytes > pool2_limit || bytes > 0xFFFFFFFFull) { left.push_back(j); continue; }
                    if (pool_used + bytes > pool2_limit) { group_begin.push_back((uint32_t)j2.size()); pool_used = 0; }
                    x.pool_off = pool_used; x.pool_bytes = (uint32_t)bytes;
                    pool_used += bytes;
                    pool_need = std::max(pool_need, pool_used);
                }
                j2.push_back(x);
            }
            group_begin.push_back((uint32_t)j2.size());
            if (j2.empty()) return MMGPU_OK;
            const size_t n_groups = group_begin.size() - 1;
            if (j2.size() * sizeof(Block2Job) > B.j2.bytes) B_TRY(B.j2.alloc(j2.size() * sizeof(Block2Job)));
            if (n_groups * 4 > B.cnt.bytes) B_TRY(B.cnt.alloc(n_groups * 4));
            if (pool_need > B.pool.bytes) B_TRY(B.pool.alloc((size_t)pool_need));
            B_TRY(hipMemcpyAsync(B.j2.p, j2.data(), j2.size() * sizeof(Block2Job), hipMemcpyHostToDevice, B.st));
            B_TRY(hipMemsetAsync(B.cnt.p, 0, n_groups * 4, B.st));
            Block2Launch L2;
            L2.q_res = L.q_res; L2.q_cb = L.q_cb; L2.q_off = L.q_off; L2.t_res = L.t_res; L2.t_off4 = L.t_off4;
            L2.scores = L.scores; L2.gap_open = L.gap_open; L2.gap_extend = L.gap_extend;
            L2.out = L.out; L2.bt_off = L.bt_off; L2.bt = no_strings ? nullptr : L.bt;
            L2.pool = B.pool.as<uint8_t>();
            L2.growth = L.growth; L2.growth_cap = L.growth_cap;
            L2.trace_bytes = form == 1 ? 0u : 1u;
            for (size_t g = 0; g < n_groups; g++) {
                L2.jobs = B.j2.as<Block2Job>() + group_begin[g];
                L2.n_jobs = group_begin[g + 1] - group_begin[g];
                L2.counter = B.cnt.as<uint32_t>() + g;
                const uint32_t waves = (uint32_t)std::min<uint64_t>(form == 1 ? (L2.n_jobs + 3) / 4 : L2.n_jobs, (uint64_t)std::max(c->compute_units, 1) * waves_per_cu);
                const size_t ck_bytes = (size_t)waves * (form == 1 ? 4 : 1) * 8 * (size_t)rows;
                if (ck_bytes > B.ck.bytes) B_TRY(B.ck.alloc(ck_bytes));
                L2.ck_pool = B.ck.as<uint8_t>();
                B_TRY(launch_sw_block4(L2, !starts_only, form, waves, B.st));
                if (!starts_only) B_TRY(launch_sw_block4_walk(L2, B.st));
            }
            if (trace_on) fprintf(stderr, "[mmgpu block aligner] %s, blocks <= %d rows, %llu trace bytes per residue: %zu pairs in %zu group(s), pool %.1f MB\n",
                                  form == 1 ? "four pairs per wavefront" : "one pair per wavefront, rows skewed over columns", rows, (unsigned long long)per_res,
                                  j2.size(), n_groups, pool_need / 1048576.0);
            return MMGPU_OK;
        };
        // ... and its end: the answers; what it handed on (MMGPU_BLOCK_TOO_LARGE) is appended to `left`
        auto block4_collect = [&](PassBufs &B, std::vector<BlockJob> &left) -> int {
            if (B.jobs.empty()) return MMGPU_OK;
            B_TRY(hipStreamSynchronize(B.st));
            B_TRY(hipMemcpy(out, d_out.p, (size_t)n * sizeof(mmgpu_sw_block), hipMemcpyDeviceToHost));
            for (const Block2Job &x : B.jobs)
                if (out[x.slot].status == MMGPU_BLOCK_TOO_LARGE) {
                    BlockJob j;
                    j.query = x.query; j.target = x.target; j.score = x.score; j.q_end = x.q_end; j.t_end = x.t_end; j.slot = x.slot;
                    const uint32_t rs = (uint32_t)out[x.slot].reserved;
                    resume[x.slot] = std::max(32u, std::min(rs & 0xFFFFu, (uint32_t)BLOCK_REF_MAX_SIZE)) | (rs & 0x10000u);
                    left.push_back(j);
                }
            if (trace_on && !B.jobs.empty()) {      // where in the (longest-first) launch the handed-on pairs stood
                size_t by_16th[16] = {};
                for (size_t k = 0; k < B.jobs.size(); k++)
                    if (out[B.jobs[k].slot].status == MMGPU_BLOCK_TOO_LARGE) by_16th[k * 16 / B.jobs.size()]++;
                fprintf(stderr, "[mmgpu block aligner] handed on, by sixteenth of the launch's %zu pairs:", B.jobs.size());
                for (size_t z : by_16th) fprintf(stderr, " %zu", z);
                fprintf(stderr, "\n");
            }
            B.jobs.clear();
            return MMGPU_OK;
        };
        auto longest_first = [&](std::vector<BlockJob> &v) {
            std::stable_sort(v.begin(), v.end(), [&](const BlockJob &x, const BlockJob &y) { return pair_len(x) > pair_len(y); });
        };
        std::vector<BlockJob> seq_jobs, head, rest, handed, again, skew, skew_again;
        for (const BlockJob &j : jobs) {      // (`jobs` is sorted longest first)
            if (!b->h_query_is_profile.empty() && b->h_query_is_profile[j.query]) slow_jobs.push_back(j);
            else seq_jobs.push_back(j);
        }
        const size_t before = slow_jobs.size();
        // The longest pairs - one per CU - go straight to the skewed form on a stream of their own, beside everything else: the pairs
        // whose blocks grow to thousands of rows are among them, each a dependent chain of tens of milliseconds that nothing shortens
        // but starting it first.
        const size_t n_head = seq_jobs.size() >= 4096 ? std::min<size_t>(seq_jobs.size() / 16, (size_t)std::max(c->compute_units, 1)) : 0;
        head.assign(seq_jobs.begin(), seq_jobs.begin() + n_head);
        rest.assign(seq_jobs.begin() + n_head, seq_jobs.end());
        int rc2;
        if (n_head) {
            if (hipStreamCreateWithFlags(&head_stream, hipStreamNonBlocking) != hipSuccess) return fail(MMGPU_ERR_HIP, "mmgpu_sw_block_backtrace: hipStreamCreate");
            PB[1].st = head_stream;
            B_TRY(hipStreamSynchronize(s));      // (the uploads above)
            rc2 = block4_launch(PB[1], head, 3, 1024, 2048, 4, skew_again);
            if (rc2 != MMGPU_OK) { drop_head_stream(); return rc2; }
        }
        // launch 1: everything else, slots for the usual 32 / 64-row blocks (half a byte per cell).  Resident wavefronts by LDS: 3.4 KB of
        // score table + 8 KB (four pairs' border arrays); the skewed form's 32 KB allow four
        // (Measured and dropped in round 6: launch 1 cut in two - the longest eighth of the pairs first, so that its hand-ons could start
        // their chains in the skewed form beside the other seven eighths.  500 of the 536 hand-ons of configs[2] do come from that
        // eighth, but it takes 15 ms of its own - the longest pairs - and the 500 chains another 25: 44.7 ms per call against 43.5.
        // The call is as long as ONE pair's chain through blocks of thousands of rows on one wavefront.)
        rc2 = block4_launch(PB[0], rest, 1, block4_per_res, 512, block4_waves, skew);
        if (rc2 == MMGPU_OK) rc2 = block4_collect(PB[0], handed);
        if (rc2 != MMGPU_OK) { drop_head_stream(); return rc2; }
        lap("four pairs per wavefront + status download");
        // ... once more for the pairs whose slot was too small (256-row blocks all the way: 128 bytes per residue)
        for (const BlockJob &j : handed) (resume[j.slot] & 0x10000u ? again : skew).push_back(j);
        if (!again.empty()) {
            longest_first(again);
            rc2 = block4_launch(PB[0], again, 1, 160, 2048, block4_waves, skew);
            if (rc2 == MMGPU_OK) rc2 = block4_collect(PB[0], skew);
            if (rc2 != MMGPU_OK) { drop_head_stream(); return rc2; }
            lap("four pairs per wavefront, larger slots + status download");
        }
        // launch 2: the skewed form for blocks beyond 256 rows (a byte per cell: slots for blocks of 1024 rows, then the crate's bound)
        longest_first(skew);
        rc2 = block4_launch(PB[0], skew, 3, 1024, 2048, 4, skew_again);
        if (rc2 == MMGPU_OK) rc2 = block4_collect(PB[0], skew_again);
        if (rc2 == MMGPU_OK && n_head) rc2 = block4_collect(PB[1], skew_again);
        if (rc2 != MMGPU_OK) { drop_head_stream(); return rc2; }
        if (!skew.empty() || n_head) lap("skewed form (and the head) + status download");
        const size_t n_skew = skew.size() + n_head;
        if (!skew_again.empty()) {
            longest_first(skew_again);
            rc2 = block4_launch(PB[0], skew_again, 3, 4096, 8192, 4, slow_jobs);
            if (rc2 == MMGPU_OK) rc2 = block4_collect(PB[0], slow_jobs);
            if (rc2 != MMGPU_OK) { drop_head_stream(); return rc2; }
            lap("skewed form, the crate's slots + status download");
        }
        drop_head_stream();
#undef B_TRY
        longest_first(slow_jobs);
        b->block_pairs_skew = (uint32_t)(n_skew - (slow_jobs.size() - before));
        b->block_pairs_fast = (uint32_t)(seq_jobs.size() - n_skew);
    } else {
        slow_jobs = jobs;
    }
    // (what is left: profile queries, pairs whose slot would not fit the pool)
    const uint64_t tier0_entries = first_tier_env ? 2 : BLOCK_MAX_SIZE / 64;
    std::vector<BlockJob> wait[3];      // longest first inside each
    if (!slow_jobs.empty()) typical_len = pair_len(slow_jobs[slow_jobs.size() > 1024 ? 255 : 0]);
    for (const BlockJob &j : slow_jobs) wait[first_tier == 0 && pair_len(j) > typical_len ? 1 : first_tier].push_back(j);
    hipStream_t extra[2] = {nullptr, nullptr};
    auto release_streams = [&] { for (hipStream_t &x : extra) if (x) { (void)hipStreamDestroy(x); x = nullptr; } };
    while (!wait[0].empty() || !wait[1].empty() || !wait[2].empty()) {
        int n_launched = 0;
        for (int tier = 2; tier >= 0; tier--) {      // (the long chains first)
            std::vector<BlockJob> &todo = wait[tier];
            if (todo.empty()) continue;
            const uint64_t longest_todo = pair_len(todo.front());
            const uint64_t slot_bytes = tier == 0   ? slot_size(typical_len, tier0_entries, BLOCK_MAX_SIZE, false)
                                        : tier == 1 ? slot_size(longest_todo, BLOCK_MID_SIZE / 64, BLOCK_MID_SIZE, false)
                                                    : slot_size(longest_todo, BLOCK_REF_MAX_SIZE / 64, BLOCK_REF_MAX_SIZE, true);
            // (tier 0: 16 wavefronts per CU is what its 9 KB of LDS allows; 8 per CU, or 32 with a 128-row first tier, move the call by
            // +12 % / -3 %, profiles/r04_exp_block_occupancy.txt - the kernel is bound by its ~200 instructions per column, not by latency)
            const uint64_t want = std::min<uint64_t>(todo.size(), (uint64_t)std::max(c->compute_units, 1) * (tier == 0 ? 16 : 4));
            const uint32_t slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, pool_limit / slot_bytes));
            hipStream_t st = s;
            if (n_launched > 0) {
                if (!extra[n_launched - 1] && hipStreamCreateWithFlags(&extra[n_launched - 1], hipStreamNonBlocking) != hipSuccess) { release_streams(); return fail(MMGPU_ERR_HIP, "mmgpu_sw_block_backtrace: hipStreamCreate"); }
                st = extra[n_launched - 1];
            }
#define R_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { release_streams(); return fail(MMGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
            if ((size_t)slot_bytes * slots > d_pool[tier].bytes) R_TRY(d_pool[tier].alloc((size_t)slot_bytes * slots));
            if ((size_t)slots * 4 > d_busy[tier].bytes) R_TRY(d_busy[tier].alloc((size_t)slots * 4));
            if (todo.size() * sizeof(BlockJob) > d_jobs[tier].bytes) R_TRY(d_jobs[tier].alloc(todo.size() * sizeof(BlockJob)));
            R_TRY(hipMemsetAsync(d_busy[tier].p, 0, (size_t)slots * 4, st));
            R_TRY(hipMemcpyAsync(d_jobs[tier].p, todo.data(), todo.size() * sizeof(BlockJob), hipMemcpyHostToDevice, st));
            L.jobs = d_jobs[tier].as<BlockJob>();
            L.n_jobs = (uint32_t)todo.size();
            L.pool = d_pool[tier].as<uint8_t>();
            L.slot_bytes = slot_bytes;
            L.n_pool_slots = slots;
            L.pool_busy = d_busy[tier].as<uint32_t>();
            R_TRY(launch_sw_block(L, tier, st));
            if (trace_on) fprintf(stderr, "[mmgpu block aligner] tier %d: %zu pairs, %u slots of %.1f KB\n", tier, todo.size(), slots, slot_bytes / 1024.0);
            n_launched++;
        }
        for (hipStream_t x : extra) if (x) R_TRY(hipStreamSynchronize(x));
        R_TRY(hipStreamSynchronize(s));
        R_TRY(hipMemcpy(out, d_out.p, (size_t)n * sizeof(mmgpu_sw_block), hipMemcpyDeviceToHost));
#undef R_TRY
        std::vector<BlockJob> next[3];
        for (int tier = 0; tier < 3; tier++) {
            size_t left = 0;
            for (const BlockJob &j : wait[tier])
                if (out[j.slot].status == MMGPU_BLOCK_TOO_LARGE && tier < 2) { next[tier + 1].push_back(j); left++; }
            b->block_pairs_tier[tier] += (uint32_t)(wait[tier].size() - left);
        }
        for (int tier = 0; tier < 3; tier++) {
            std::stable_sort(next[tier].begin(), next[tier].end(), [&](const BlockJob &x, const BlockJob &y) { return pair_len(x) > pair_len(y); });
            wait[tier].swap(next[tier]);
        }
        lap("round of tier launches + status download");
    }
    release_streams();
    if (au) {
        // ---- the answers into the batch's records; what the block aligner declined gets its reverse scan (:873-882) ----
        HIP_TRY(d_flags.alloc((size_t)b->pairs));
        HIP_TRY(hipMemsetAsync(d_flags.p, 0, (size_t)b->pairs, s));
        HIP_TRY(hipMemsetAsync(d_sel_cnt.p, 0, 32, s));
        BlockScatterArgs S;
        S.blk = d_out.as<mmgpu_sw_block>();
        S.pair_of_slot = d_sel_pairs.as<uint32_t>();
        S.n = n;
        S.res = b->d_out.as<mmgpu_sw_hit>();
        S.rev_force = d_flags.as<uint8_t>();
        S.counts = d_sel_cnt.as<uint32_t>();
        HIP_TRY(launch_block_scatter(S, s));
        uint32_t counts[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyAsync(counts, d_sel_cnt.p, 12, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        au->ok = counts[0]; au->declined = counts[1]; au->too_large = counts[2];
        if (counts[1]) {
            if (int e = sw_launch_groups(c, b, true, d_flags.as<uint8_t>())) return e;
            HIP_TRY(hipStreamSynchronize(s));
        }
        b->h_res_valid = false;
        lap("answers scattered into the records (+ reverse scan of declined pairs)");
        return MMGPU_OK;
    }
    if (off && !no_strings) HIP_TRY(hipMemcpyAsync(bt, d_bt.p, (size_t)off, hipMemcpyDeviceToHost, s));
    if (growth) HIP_TRY(hipMemcpyAsync(growth, d_growth.p, growth_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));      // the host vectors and the buffers above die with this scope
    lap("backtrace strings download");
    return MMGPU_OK;
}

// One call = what `mmseqs search` needs of the block aligner in alignment mode 2 without backtraces: the device picks every int16-range
// hit that passes its query's start-score threshold, runs the block aligner for start positions only, writes them into the batch's
// records and scans backwards for the pairs it declined.  Afterwards mmgpu_sw_fetch returns records whose start positions are the
// reference's for every pair that passes the gate, whichever path they took.
extern "C" int mmgpu_sw_block_starts(mmgpu_ctx *c, mmgpu_sw_batch_t *b, uint32_t *n_selected, uint32_t *n_declined, uint32_t *n_too_large) {
    if (!c || !b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_starts: NULL argument");
    if (b->mode != MMGPU_SW_START_NOT_WORD) return fail(MMGPU_ERR_STATE, "mmgpu_sw_block_starts: the batch was not prepared with MMGPU_SW_START_NOT_WORD");
    if (b->owned) return fail(MMGPU_ERR_UNSUPPORTED, "mmgpu_sw_block_starts: not for batches of owned pairs (sharded runs)");
    BlockAuto au;
    const int rc = block_backtrace(c, b, nullptr, 0, nullptr, nullptr, MMGPU_BLOCK_STARTS_ONLY, nullptr, nullptr, 0, &au);
    if (n_selected) *n_selected = au.selected;
    if (n_declined) *n_declined = au.declined;
    if (n_too_large) *n_too_large = au.too_large;
    return rc;
}

extern "C" int mmgpu_sw_block_backtrace(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *pair_index, uint32_t n, mmgpu_sw_block *out,
                                        char *bt, size_t bt_cap, size_t *bt_used) {
    return block_backtrace(c, b, pair_index, n, out, bt, bt_cap, bt_used, nullptr, 0);
}

extern "C" int mmgpu_sw_block_growth(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *pair_index, uint32_t n, mmgpu_sw_block *out,
                                     uint32_t *growth, uint32_t growth_cap) {
    if (!growth || growth_cap == 0) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_growth: no buffer");
    return block_backtrace(c, b, pair_index, n, out, nullptr, MMGPU_BLOCK_NO_STRINGS, nullptr, growth, growth_cap);
}

extern "C" int mmgpu_sw_block_tiers(const mmgpu_sw_batch_t *b, uint32_t *first_tier, uint32_t *second_tier) {
    if (!b) return fail(MMGPU_ERR_ARG, "mmgpu_sw_block_tiers: NULL batch");
    if (first_tier) *first_tier = b->block_pairs_fast + b->block_pairs_tier[0];
    if (second_tier) *second_tier = b->block_pairs_skew + b->block_pairs_tier[1] + b->block_pairs_tier[2];
    return MMGPU_OK;
}

extern "C" int mmgpu_sw_traceback(mmgpu_ctx *c, mmgpu_sw_batch_t *b, const uint32_t *pair_index, uint32_t n, mmgpu_sw_bt *info,
                                  char *bt, size_t bt_cap, size_t *bt_used) {
    if (!c || !b || (!pair_index && n) || (!info && n)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_traceback: NULL argument");
    if (b->mode < MMGPU_SW_START) return fail(MMGPU_ERR_STATE, "mmgpu_sw_traceback: the batch was prepared without MMGPU_SW_START");
    if (!b->ran) return fail(MMGPU_ERR_STATE, "mmgpu_sw_traceback: batch was never run");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // host copies of the results (and, for device-resident lists, of the slot -> target map): fetched once per run of the
    // batch, callers ask for the size first and for the strings second
    if (!b->h_res_valid) {
        b->h_res.resize((size_t)b->pairs);
        if (b->pairs) HIP_TRY(hipMemcpyAsync(b->h_res.data(), b->d_out.p, (size_t)b->pairs * sizeof(mmgpu_sw_hit), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (b->from_pf) {
            b->h_slot_target.resize((size_t)b->pairs);
            if (b->pairs) HIP_TRY(hipMemcpy(b->h_slot_target.data(), b->d_slot_target.p, (size_t)b->pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        b->h_res_valid = true;
    }
    const std::vector<mmgpu_sw_hit> &res = b->h_res;
    const std::vector<uint32_t> &pf_host = b->h_slot_target;
    std::vector<BtJob> jobs;
    jobs.reserve(n);
    uint64_t off = 0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t p = pair_index[k];
        if (p >= b->pairs) return fail(MMGPU_ERR_ARG, "mmgpu_sw_traceback: pair index out of range");
        const mmgpu_sw_hit &h = res[p];
        info[k].bt_off = off;
        info[k].bt_len = 0;
        info[k].ident = 0;
        info[k].reserved = 0;
        if (h.score <= 0 || h.q_start < 0 || h.t_start < 0 || h.q_end < h.q_start || h.t_end < h.t_start) {
            info[k].status = MMGPU_BT_NO_START;
            continue;
        }
        info[k].status = MMGPU_BT_TOO_LARGE;   // overwritten by the kernel
        BtJob j;
        j.slot = k;
        j.query = (uint32_t)(std::upper_bound(b->h_qout_off.begin(), b->h_qout_off.end(), p) - b->h_qout_off.begin() - 1);
        j.target = b->from_pf ? pf_host[p] : b->h_out_target[p];
        j.q_start = h.q_start; j.q_end = h.q_end; j.t_start = h.t_start; j.t_end = h.t_end; j.score = h.score;
        j.bt_off = off;
        off += (uint64_t)(h.q_end - h.q_start + 1) + (uint64_t)(h.t_end - h.t_start + 1) + 1;
        jobs.push_back(j);
    }
    if (bt_used) *bt_used = (size_t)off;
    if (off > bt_cap || (!bt && off)) return fail(MMGPU_ERR_ARG, "mmgpu_sw_traceback: bt buffer too small (see *bt_used)");
    if (jobs.empty()) return MMGPU_OK;
    auto cost = [](const BtJob &j) {
        const int64_t ql = j.q_end - j.q_start + 1, tl = j.t_end - j.t_start + 1;
        return ql * (2 * (std::llabs(tl - ql) + 1) + 1);
    };
    std::stable_sort(jobs.begin(), jobs.end(), [&](const BtJob &x, const BtJob &y) { return cost(x) > cost(y); });
    HIP_TRY(b->d_bt_jobs.reserve(jobs.size() * sizeof(BtJob)));
    HIP_TRY(b->d_bt_info.reserve((size_t)n * sizeof(mmgpu_sw_bt)));
    HIP_TRY(b->d_bt_str.reserve((size_t)off + 16));
    HIP_TRY(hipMemcpyAsync(b->d_bt_info.p, info, (size_t)n * sizeof(mmgpu_sw_bt), hipMemcpyHostToDevice, s));
    BtLaunch L;
    L.q_res = b->d_qres.as<uint8_t>();
    L.q_cb = b->d_qcb.as<int8_t>();
    L.q_off = b->d_qoff.as<uint32_t>();
    L.q_prof = b->any_profile ? b->d_qprof.as<int8_t>() : nullptr;
    L.q_prof_off = b->any_profile ? b->d_qprof_off.as<uint32_t>() : nullptr;
    L.t_res = c->db.res;
    L.t_off4 = c->db.off4;
    L.mat = b->d_mat.as<int8_t>();
    L.alphabet = b->alphabet;
    L.gap_open = b->gap_open;
    L.gap_extend = b->gap_extend;
    L.info = b->d_bt_info.as<mmgpu_sw_bt>();
    L.bt = b->d_bt_str.as<char>();
    // Scratch tiers (band rows | direction words per lane | blocks of 64 alignments in flight):
    //   0: rows in LDS (band <= 32),     8 K words (64 K cells)  - nearly every pair of a hit list
    //   1: rows in LDS (band <= 32),   128 K words (1 M cells)   - long alignments with a narrow band
    //   2: rows in scratch (band <= 256), 128 K words
    //   3: rows in scratch (band <= 4096),  1 M words (8 M cells), few at a time; beyond: MMGPU_BT_TOO_LARGE, the host runs banded_sw
    // A job starts at the first tier its initial band (|tlen - qlen| + 1) fits and moves up when the doubling outgrows it.
    // band 0 in BtLaunch selects the LDS form of the kernel.
    const int n_tiers = 4;
    const uint32_t tier_band[n_tiers] = {0u, 0u, 515u, 8195u}, tier_dir[n_tiers] = {8192u, 131072u, 131072u, 1048576u};
    const uint32_t tier_blocks[n_tiers] = {16384u, 256u, 256u, 48u};
    auto first_tier = [&](const BtJob &j) {
        const int64_t ql = j.q_end - j.q_start + 1, tl = j.t_end - j.t_start + 1;
        const int64_t bw = std::llabs(tl - ql) + 1, width = 2 * bw + 3, words = (2 * bw + 1 + 7) / 8 * ql;
        for (int t = 0; t < n_tiers; t++) {
            const int64_t cap = tier_band[t] ? (int64_t)tier_band[t] : 67;
            if (width <= cap && words <= (int64_t)tier_dir[t]) return t;
        }
        return n_tiers - 1;
    };
    std::vector<mmgpu_sw_bt> back((size_t)n);
    std::vector<BtJob> pending = jobs, now;
    const bool trace = getenv("MMGPU_TRACE") != nullptr;
    L.dir_pool = nullptr;
    L.dir_pool_bytes = 0;
    L.dir_cursor = nullptr;
    static const bool lane_kernel_only = getenv("MMGPU_BT_LANE_KERNEL") != nullptr;   // cross-check switch: skip the wave kernel
    if (!lane_kernel_only) {
        // ---- the wave kernel takes every job first (bt_wave_kernel.hip); what it declines (band beyond its LDS ring,
        // direction pool exhausted) goes through the tiers of the lane-per-alignment kernel below
        uint64_t want = 0;
        for (const BtJob &j : jobs) {
            const uint64_t ql = (uint64_t)(j.q_end - j.q_start + 1), tl = (uint64_t)(j.t_end - j.t_start + 1);
            const uint64_t bw0 = (ql > tl ? ql - tl : tl - ql) + 1;
            want += ql * (2 * std::min<uint64_t>(4 * bw0, 510) + 1);
        }
        const uint64_t pool = std::min<uint64_t>(std::max<uint64_t>(want, 64ull << 20), 8ull << 30);
        HIP_TRY(b->d_bt_scratch.reserve(pool + 16));
        HIP_TRY(b->d_bt_cursor.reserve(16));
        HIP_TRY(hipMemsetAsync(b->d_bt_cursor.p, 0, 16, s));
        HIP_TRY(hipMemcpyAsync(b->d_bt_jobs.p, jobs.data(), jobs.size() * sizeof(BtJob), hipMemcpyHostToDevice, s));
        L.dir_pool = b->d_bt_scratch.as<uint8_t>();
        L.dir_pool_bytes = pool;
        L.dir_cursor = b->d_bt_cursor.as<unsigned long long>();
        L.jobs = b->d_bt_jobs.as<BtJob>();
        L.n_jobs = (uint32_t)jobs.size();
        HIP_TRY(launch_sw_traceback_wave(L, s));
        HIP_TRY(hipMemcpyAsync(back.data(), b->d_bt_info.p, (size_t)n * sizeof(mmgpu_sw_bt), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        pending.clear();
        for (const BtJob &j : jobs)
            if (back[j.slot].status == MMGPU_BT_TOO_LARGE) pending.push_back(j);
        if (trace) {
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[sw_traceback] wave kernel: %zu jobs, %zu declined, pool %.1f MB\n", jobs.size(), pending.size(), (double)pool / 1048576.0);
            (void)t;
        }
    }
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&]() {
        const auto t = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t - t_prev).count();
        t_prev = t;
        return ms;
    };
    if (trace) fprintf(stderr, "[sw_traceback] %u pairs asked, %zu with a start position\n", n, jobs.size());
    for (int tier = 0; tier < n_tiers && !pending.empty(); tier++) {
        now.clear();
        std::vector<BtJob> later;
        for (const BtJob &j : pending) (first_tier(j) <= tier ? now : later).push_back(j);
        if (now.empty()) continue;
        const uint32_t wpl = 3u * tier_band[tier] + tier_dir[tier];
        const size_t blocks_max = tier_blocks[tier];
        const size_t njobs = now.size();
        const size_t blocks_needed = std::min<size_t>((njobs + 63) / 64, blocks_max);
        HIP_TRY(b->d_bt_scratch.reserve(blocks_needed * (size_t)wpl * 64 * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(b->d_bt_jobs.p, now.data(), now.size() * sizeof(BtJob), hipMemcpyHostToDevice, s));
        L.scratch = b->d_bt_scratch.as<uint32_t>();
        L.words_per_lane = wpl;
        L.band_cap = tier_band[tier];
        for (size_t j0 = 0; j0 < njobs; j0 += blocks_max * 64) {
            L.jobs = b->d_bt_jobs.as<BtJob>() + j0;
            L.n_jobs = (uint32_t)std::min<size_t>(njobs - j0, blocks_max * 64);
            HIP_TRY(launch_sw_traceback(L, s));
        }
        HIP_TRY(hipMemcpyAsync(back.data(), b->d_bt_info.p, (size_t)n * sizeof(mmgpu_sw_bt), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        size_t moved = 0;
        for (const BtJob &j : now)
            if (back[j.slot].status == MMGPU_BT_TOO_LARGE && tier + 1 < n_tiers) { later.push_back(j); moved++; }
        if (trace) fprintf(stderr, "[sw_traceback] tier %d: %zu jobs, %zu moved up, %.2f ms\n", tier, now.size(), moved, lap());
        pending.swap(later);
    }
    for (uint32_t k = 0; k < n; k++)
        if (info[k].status != MMGPU_BT_NO_START) info[k] = back[k];
    if (off) HIP_TRY(hipMemcpyAsync(bt, b->d_bt_str.p, (size_t)off, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (trace) fprintf(stderr, "[sw_traceback] strings downloaded (%.1f MB), %.2f ms\n", (double)off / 1048576.0, lap());
    return MMGPU_OK;
}
