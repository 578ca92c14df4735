// MMGpuAlignBackend over the C-ABI of libmmgpu: keeps the prepared batch of the last align() call resident so that
// traceback() can address its pairs.  Link with -lmmgpu.
#include <cstring>

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
    int blockBacktrace(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_block *out, std::string &strings) {
        size_t need = 0;
        int rc = mmgpu_sw_block_backtrace(gpu, batch, pairIndex, n, out, NULL, 0, &need);
        if (rc != 0 && need == 0) return rc;
        strings.assign((size_t)need, '\0');
        return mmgpu_sw_block_backtrace(gpu, batch, pairIndex, n, out, need ? &strings[0] : NULL, need, &need);
    }
    const char *lastError() { return mmgpu_last_error(); }

private:
    mmgpu_ctx *gpu;
    mmgpu_sw_batch_t *batch;
    std::vector<uint32_t> qlens;
};

MMGpuAlignBackend *mmgpuNewDeviceBackend(mmgpu_ctx *gpu) { return new MMGpuDeviceBackend(gpu); }
