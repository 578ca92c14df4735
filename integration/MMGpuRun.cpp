#include "MMGpuRun.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
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
// device: the shard logic can then be exercised on a 1-GPU box).  Fewer than two ids: one device (MMGPU_DEVICE).
const std::vector<int> &MMGpuRun::deviceIds() {
    static std::vector<int> ids;
    static bool parsed = false;
    if (parsed) return ids;
    parsed = true;
    const char *e = getenv("MMGPU_DEVICES");
    if (e == NULL || e[0] == '\0') return ids;
    for (const char *p = e; *p != '\0';) {
        char *end = NULL;
        const long v = strtol(p, &end, 10);
        if (end == p) break;
        ids.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
    }
    if (ids.size() < 2) ids.clear();
    return ids;
}

// The layout of a prefilter run over N device contexts: G query groups x S target shards (N = G * S).  A group holds the whole
// target split, dealt to its S contexts; the query blocks of a run are dealt to the groups.  The model behind the default is
// bench.py's choose_query_groups() (stage times of the 10 000 x 1 M search on one device, ms, round 5: the similar-k-mer stage k = 10.5
// follows the queries only, e = 142.4 follows the index entries a context holds, x = 3 per exchange step): t = k / G + e / (G S)
// + x (S > 1), with two shards or more per group, so that a device holds half of the index at most.
int MMGpuRun::queryGroups(int nDevices, bool shardsPossible) {
    if (nDevices < 2) return 1;
    if (!shardsPossible) return nDevices;      // no exchange for this configuration: every context holds the split, queries dealt
    const char *e = getenv("MMGPU_QUERY_GROUPS");
    if (e != NULL && e[0] != '\0') {
        const int g = atoi(e);
        if (g < 1 || nDevices % g != 0) {
            Debug(Debug::ERROR) << "MMGPU: MMGPU_QUERY_GROUPS=" << e << " does not divide the " << nDevices << " contexts of MMGPU_DEVICES\n";
            EXIT(EXIT_FAILURE);
        }
        return g;
    }
    const double k = 10.5, ent = 142.4, x = 3.0;      // (ms per 10 000 queries x 1 M targets, profiles/r05_bench_n1.json)
    int best = 1;
    double bestT = -1;
    for (int g = 1; g <= nDevices; g++) {
        if (nDevices % g != 0 || nDevices / g < 2) continue;
        const int s = nDevices / g;
        const double t = k / g + ent / ((double)g * s) + x;
        if (bestT < 0 || t < bestT - 1e-9) {
            best = g;
            bestT = t;
        }
    }
    return best;
}

namespace {
std::vector<mmgpu_multi *> layout;      // the groups opened last (a process runs one module at a time)
size_t layoutContexts = 0;
std::mutex layoutLock;
}

const std::vector<mmgpu_multi *> &MMGpuRun::groups(int g, int shardsOfOneDevice) {
    std::lock_guard<std::mutex> guard(layoutLock);
    std::vector<int> ids = deviceIds();
    // no MMGPU_DEVICES, but a target split beyond what one context indexes (MMGPU_PF_MAX_TARGETS): that many contexts on the one device
    if (ids.empty() && shardsOfOneDevice > 1) ids.assign((size_t)shardsOfOneDevice, (int)envSize("MMGPU_DEVICE", 0));
    if (ids.empty()) return layout;      // (empty)
    if (g < 1) g = layout.empty() ? 1 : (int)layout.size();
    if ((int)layout.size() == g && layoutContexts == ids.size()) return layout;
    for (size_t i = 0; i < layout.size(); i++) mmgpu_destroy_multi(layout[i]);
    layout.clear();
    layoutContexts = ids.size();
    const int s = (int)ids.size() / g;
    for (int i = 0; i < g; i++) {
        mmgpu_multi *m = NULL;
        if (mmgpu_init_multi(&m, ids.data() + (size_t)i * s, s) != 0) {
            Debug(Debug::ERROR) << "MMGPU: cannot open " << ids.size() << " device contexts: " << mmgpu_last_error() << "\n";
            EXIT(EXIT_FAILURE);
        }
        layout.push_back(m);
    }
    char transport[32] = "none";
    if (s > 1) mmgpu_comm_info(mmgpu_multi_ctx(layout[0], 0), NULL, NULL, transport, sizeof(transport));
    Debug(Debug::INFO) << "MMGPU: " << ids.size() << " device contexts (" << (deviceIds().empty() ? "one device, target split above the limit of a context" : "MMGPU_DEVICES=" + std::string(getenv("MMGPU_DEVICES"))) << "): " << g
                       << " query group" << (g > 1 ? "s" : "") << " x " << s << " target shard" << (s > 1 ? "s" : "")
                       << ", exchange transport: " << transport << "\n";
    return layout;
}

std::vector<mmgpu_ctx *> MMGpuRun::allContexts() {
    std::vector<mmgpu_ctx *> all;
    if (deviceIds().empty()) return all;
    const std::vector<mmgpu_multi *> &gs = groups(0);
    for (size_t i = 0; i < gs.size(); i++)
        for (int d = 0; d < mmgpu_multi_size(gs[i]); d++) all.push_back(mmgpu_multi_ctx(gs[i], d));
    return all;
}
