#include "MMGpuRun.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "Debug.h"
#include "Util.h"

bool MMGpuRun::enabled() {
    const char *e = getenv("MMGPU_DISABLE");
    return !(e != NULL && e[0] != '\0' && e[0] != '0');
}

// MMGPU_BLOCK_ALIGNER: "device" (default) | "host" | "sw"
bool MMGpuRun::hostBlockAligner() {      // the host's alignStartPosBacktraceBlock is installed as a hook
    const char *e = getenv("MMGPU_BLOCK_ALIGNER");
    return !(e != NULL && strcmp(e, "sw") == 0);
}
bool MMGpuRun::deviceBlockAligner() {
    const char *e = getenv("MMGPU_BLOCK_ALIGNER");
    return e == NULL || e[0] == '\0' || strcmp(e, "device") == 0;
}

size_t MMGpuRun::envSize(const char *name, size_t fallback) {
    const char *e = getenv(name);
    if (e == NULL || e[0] == '\0') return fallback;
    const long long v = atoll(e);
    return v > 0 ? (size_t)v : fallback;
}

mmgpu_ctx *MMGpuRun::context() {
    static mmgpu_ctx *ctx = NULL;
    static std::mutex lock;      // a fused search opens the device on a helper thread while the prefilter module sets itself up
    std::lock_guard<std::mutex> guard(lock);
    if (ctx == NULL) {
        const int device = (int)envSize("MMGPU_DEVICE", 0);
        if (mmgpu_init(&ctx, device) != 0) {
            // no silent CPU fallback: a build with the device path enabled either runs on the device or stops
            Debug(Debug::ERROR) << "MMGPU: cannot open HIP device " << device << ": " << mmgpu_last_error()
                                << "\n(set MMGPU_DISABLE=1 to run this binary on the CPU path)\n";
            EXIT(EXIT_FAILURE);
        }
        char name[128];
        int cus = 0;
        if (mmgpu_device_info(ctx, &cus, name, sizeof(name)) == 0)
            Debug(Debug::INFO) << "MMGPU: device " << device << " " << name << " (" << cus << " compute units)\n";
    }
    return ctx;
}

// MMGPU_DEVICES=0,1,2,...: all listed devices of the node work on one module call (a repeated id puts several contexts on one
// device: the shard logic can then be exercised on a 1-GPU box).  NULL: one device (MMGPU_DEVICE).
mmgpu_multi *MMGpuRun::multi() {
    static mmgpu_multi *m = NULL;
    static bool tried = false;
    if (tried) return m;
    tried = true;
    const char *e = getenv("MMGPU_DEVICES");
    if (e == NULL || e[0] == '\0') return NULL;
    std::vector<int> ids;
    for (const char *p = e; *p != '\0';) {
        char *end = NULL;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        ids.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
    }
    if (ids.size() < 2) return NULL;
    if (mmgpu_init_multi(&m, ids.data(), (int)ids.size()) != 0) {
        Debug(Debug::ERROR) << "MMGPU: cannot open the devices of MMGPU_DEVICES=" << e << ": " << mmgpu_last_error() << "\n";
        EXIT(EXIT_FAILURE);
    }
    char transport[32] = "";
    mmgpu_comm_info(mmgpu_multi_ctx(m, 0), NULL, NULL, transport, sizeof(transport));
    Debug(Debug::INFO) << "MMGPU: " << ids.size() << " device contexts (MMGPU_DEVICES=" << e << "), exchange transport: " << transport << "\n";
    return m;
}
