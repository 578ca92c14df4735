// MMGpuAlignBackend over the C-ABI of libmmgpu: keeps the prepared batch of the last align() call resident so that
// traceback() can address its pairs.  Link with -lmmgpu.
#include <cstring>
#include <string>
#include <vector>

#include "MMGpuMatcher.h"

class MMGpuDeviceBackend : public MMGpuAlignBackend {
public:
    explicit MMGpuDeviceBackend(mmgpu_ctx *gpu) : gpu(gpu), batch(NULL) {}
    ~MMGpuDeviceBackend() {
        if (batch) mmgpu_sw_free(gpu, batch);
    }
    int align(const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t nQueries, int mode, mmgpu_sw_hit *out) {
        if (batch) {
            mmgpu_sw_free(gpu, batch);
            batch = NULL;
        }
        int rc = mmgpu_sw_prepare(gpu, params, queries, nQueries, mode, &batch);
        if (rc == 0) rc = mmgpu_sw_run(gpu, batch);
        if (rc == 0) rc = mmgpu_sw_fetch(gpu, batch, out);
        qlens.resize(nQueries);
        for (uint32_t i = 0; i < nQueries; i++) qlens[i] = queries[i].qlen;
        return rc;
    }
    int traceback(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_bt *info, std::string &strings) {
        // every pair reserves (q_end - q_start + 1) + (t_end - t_start + 1) + 1 bytes: ask for the size, then run
        size_t need = 0;
        int rc = mmgpu_sw_traceback(gpu, batch, pairIndex, n, info, NULL, 0, &need);
        if (rc != 0 && need == 0) return rc;
        strings.assign((size_t)need, '\0');
        return mmgpu_sw_traceback(gpu, batch, pairIndex, n, info, &strings[0], need, &need);
    }
    int blockBacktrace(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_block *out, std::string &strings, int want) {
        size_t need = 0;
        if (want != BLOCK_STRINGS) {
            strings.clear();
            return mmgpu_sw_block_backtrace(gpu, batch, pairIndex, n, out, NULL, want == BLOCK_STARTS ? MMGPU_BLOCK_STARTS_ONLY : MMGPU_BLOCK_NO_STRINGS, &need);
        }
        int rc = mmgpu_sw_block_backtrace(gpu, batch, pairIndex, n, out, NULL, 0, &need);
        if (rc != 0 && need == 0) return rc;
        strings.assign((size_t)need, '\0');
        return mmgpu_sw_block_backtrace(gpu, batch, pairIndex, n, out, need ? &strings[0] : NULL, need, &need);
    }
    int reversePairs(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_hit *out) { return mmgpu_sw_reverse_pairs(gpu, batch, pairIndex, n, out); }
    const char *lastError() { return mmgpu_last_error(); }

private:
    mmgpu_ctx *gpu;
    mmgpu_sw_batch_t *batch;
    std::vector<uint32_t> qlens;
};

MMGpuAlignBackend *mmgpuNewDeviceBackend(mmgpu_ctx *gpu) { return new MMGpuDeviceBackend(gpu); }

// Several devices, every one holding the whole target database (MMGPU_DEVICES; the caller loads the targets into each
// context): the queries of a block are dealt to the devices in contiguous slices of about equal cell counts, the slices run
// side by side (prepare + run are enqueued for all of them before the first fetch), results keep the caller's order.
// Pair indices of traceback() / blockBacktrace() are mapped to (device, pair inside its slice).
class MMGpuMultiDeviceBackend : public MMGpuAlignBackend {
public:
    explicit MMGpuMultiDeviceBackend(const std::vector<mmgpu_ctx *> &ctx) : gpus(ctx), batches(ctx.size(), NULL), firstPair(ctx.size() + 1, 0) {}
    ~MMGpuMultiDeviceBackend() {
        for (size_t d = 0; d < gpus.size(); d++)
            if (batches[d]) mmgpu_sw_free(gpus[d], batches[d]);
    }
    int align(const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t nQueries, int mode, mmgpu_sw_hit *out) {
        const size_t nd = gpus.size();
        for (size_t d = 0; d < nd; d++)
            if (batches[d]) { mmgpu_sw_free(gpus[d], batches[d]); batches[d] = NULL; }
        // slices: contiguous, by qlen x number of targets
        std::vector<double> cost(nQueries + 1, 0.0);
        for (uint32_t i = 0; i < nQueries; i++) cost[i + 1] = cost[i] + (double)queries[i].qlen * (double)queries[i].n_targets + 1.0;
        std::vector<uint32_t> firstQuery(nd + 1, nQueries);
        firstQuery[0] = 0;
        for (size_t d = 1; d < nd; d++) {
            const double want = cost[nQueries] * (double)d / (double)nd;
            uint32_t q = firstQuery[d - 1];
            while (q < nQueries && cost[q] < want) q++;
            firstQuery[d] = q;
        }
        firstPair.assign(nd + 1, 0);
        for (size_t d = 0; d < nd; d++) {
            uint64_t pairs = 0;
            for (uint32_t i = firstQuery[d]; i < firstQuery[d + 1]; i++) pairs += queries[i].n_targets;
            firstPair[d + 1] = firstPair[d] + pairs;
        }
        int rc = 0;
        for (size_t d = 0; d < nd && rc == 0; d++) {
            const uint32_t n = firstQuery[d + 1] - firstQuery[d];
            if (n == 0) continue;
            rc = mmgpu_sw_prepare(gpus[d], params, queries + firstQuery[d], n, mode, &batches[d]);
            if (rc == 0) rc = mmgpu_sw_run(gpus[d], batches[d]);
        }
        for (size_t d = 0; d < nd && rc == 0; d++)
            if (batches[d]) rc = mmgpu_sw_fetch(gpus[d], batches[d], out + firstPair[d]);
        return rc;
    }
    int traceback(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_bt *info, std::string &strings) {
        return split<mmgpu_sw_bt>(pairIndex, n, info, strings, true);
    }
    int blockBacktrace(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_block *out, std::string &strings, int want) {
        blockWant = want;
        return split<mmgpu_sw_block>(pairIndex, n, out, strings, false);
    }
    int reversePairs(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_hit *out) {
        for (uint32_t k = 0; k < n; k++) {      // (a handful of pairs per block at most)
            size_t d = 0;
            while (d + 1 < gpus.size() && pairIndex[k] >= firstPair[d + 1]) d++;
            const uint32_t local = (uint32_t)(pairIndex[k] - firstPair[d]);
            const int rc = mmgpu_sw_reverse_pairs(gpus[d], batches[d], &local, 1, out + k);
            if (rc != 0) return rc;
        }
        return 0;
    }
    const char *lastError() { return mmgpu_last_error(); }

private:
    template <typename R>
    int split(const uint32_t *pairIndex, uint32_t n, R *out, std::string &strings, bool banded) {
        const size_t nd = gpus.size();
        strings.clear();
        std::vector<std::vector<uint32_t> > local(nd), where(nd);
        for (uint32_t k = 0; k < n; k++) {
            size_t d = 0;
            while (d + 1 < nd && pairIndex[k] >= firstPair[d + 1]) d++;
            local[d].push_back((uint32_t)(pairIndex[k] - firstPair[d]));
            where[d].push_back(k);
        }
        for (size_t d = 0; d < nd; d++) {
            if (local[d].empty()) continue;
            std::vector<R> part(local[d].size());
            size_t need = 0;
            std::string s;
            int rc;
            if (!banded && blockWant != BLOCK_STRINGS) {      // no strings wanted: one call
                rc = call(d, local[d], part.data(), NULL, blockWant == BLOCK_STARTS ? MMGPU_BLOCK_STARTS_ONLY : MMGPU_BLOCK_NO_STRINGS, &need, banded);
            } else {
                rc = call(d, local[d], part.data(), NULL, 0, &need, banded);
                if (rc != 0 && need == 0) return rc;
                s.assign(need, '\0');
                rc = call(d, local[d], part.data(), need ? &s[0] : NULL, need, &need, banded);
            }
            if (rc != 0) return rc;
            const uint64_t base = strings.size();
            strings += s;
            for (size_t z = 0; z < part.size(); z++) {
                part[z].bt_off += base;
                out[where[d][z]] = part[z];
            }
        }
        return 0;
    }
    int call(size_t d, const std::vector<uint32_t> &idx, mmgpu_sw_bt *o, char *bt, size_t cap, size_t *need, bool) {
        return mmgpu_sw_traceback(gpus[d], batches[d], idx.data(), (uint32_t)idx.size(), o, bt, cap, need);
    }
    int call(size_t d, const std::vector<uint32_t> &idx, mmgpu_sw_block *o, char *bt, size_t cap, size_t *need, bool) {
        return mmgpu_sw_block_backtrace(gpus[d], batches[d], idx.data(), (uint32_t)idx.size(), o, bt, cap, need);
    }
    std::vector<mmgpu_ctx *> gpus;
    std::vector<mmgpu_sw_batch_t *> batches;
    std::vector<uint64_t> firstPair;
    int blockWant = BLOCK_STRINGS;
};

MMGpuAlignBackend *mmgpuNewMultiDeviceBackend(const std::vector<mmgpu_ctx *> &gpus) { return new MMGpuMultiDeviceBackend(gpus); }
