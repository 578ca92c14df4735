#include "MMGpuRun.h"

#include <cstdlib>
#include <cstring>

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
