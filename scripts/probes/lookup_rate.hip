// Micro-probe: rate of independent random look-ups into a table of S bytes on gfx950, the access pattern of the prefilter's
// similar-k-mer stage (one look-up per lane per round, every wave slot of the chip busy).  What it answers: does a table that fits
// the 256 MiB Infinity Cache serve random look-ups faster than one that does not, by how much, at which request width, and does a
// stream of record writes beside the look-ups (the 16 B list record per similar k-mer) change the answer.
//   hipcc --offload-arch=gfx950 -O3 -o lookup_rate scripts/probes/lookup_rate.hip && ./lookup_rate
// Output: one line per (table MB, request bytes, records written per look-up) with G look-ups/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((packed, aligned(4))) U32Pair { uint32_t a, b; };

__device__ __forceinline__ uint32_t mix(uint32_t x) {      // cheap integer hash
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// REQ = 8: two adjacent uint32 at a random 4-byte aligned place (today's offsets[k], offsets[k + 1]);
// REQ = 32: one aligned 32-byte block (base + 28 one-byte counts); REQ = 4: one dword; REQ = 16: one aligned 16-byte block.
// WR = bytes of "record" each lane appends per look-up to a private stream (0, 8 or 16), coalesced across the wave.
template <int REQ, int WR>
__global__ __launch_bounds__(256) void probe(const uint32_t *table, uint64_t n_dwords, int rounds, uint32_t seed, uint32_t *sink, uint4 *rec) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    uint32_t x = mix(tid * 2654435761u + seed);
    const uint64_t n_units = REQ == 8 ? n_dwords - 1 : n_dwords / (REQ / 4);
    size_t rbase = ((size_t)(tid >> 6) * (size_t)rounds) * 64u + (tid & 63u);
    for (int r = 0; r < rounds; r++) {
        x = mix(x + 0x9E3779B9u);
        const uint64_t u = ((uint64_t)x * n_units) >> 32;
        uint32_t v0 = 0, v1 = 0;
        if (REQ == 4) {
            v0 = table[u];
        } else if (REQ == 8) {
            const U32Pair p = *reinterpret_cast<const U32Pair *>(table + u);
            v0 = p.a; v1 = p.b;
        } else if (REQ == 16) {
            const uint4 p = reinterpret_cast<const uint4 *>(table)[u];
            v0 = p.x ^ p.z; v1 = p.y ^ p.w;
        } else {
            const uint4 p = reinterpret_cast<const uint4 *>(table)[2 * u], q = reinterpret_cast<const uint4 *>(table)[2 * u + 1];
            v0 = p.x ^ p.z ^ q.x ^ q.z; v1 = p.y ^ p.w ^ q.y ^ q.w;
        }
        acc += v0 + (v1 - v0);
        if (WR == 16) rec[rbase + (size_t)r * 64u] = make_uint4(v0, v1, acc, tid);
        if (WR == 8) reinterpret_cast<uint2 *>(rec)[rbase + (size_t)r * 64u] = make_uint2(v0, v1);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int REQ, int WR>
static void run(const uint32_t *table, size_t table_bytes, uint32_t *sink, uint4 *rec, int blocks, int rounds) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((probe<REQ, WR>), dim3(blocks), dim3(256), 0, 0, table, (uint64_t)(table_bytes / 4), rounds, 17u + rep, sink, rec);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) {
            const double n = (double)blocks * 256.0 * rounds;
            printf("table_MB %6.0f req_B %2d rec_B %2d lookups %.3e ms %8.3f Glookups_per_s %7.2f req_GBps %8.1f\n", table_bytes / 1048576.0, REQ, WR, n,
                   ms, n / ms * 1e-6, n * REQ / ms * 1e-6);
            fflush(stdout);
        }
    }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 64;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    const int blocks = cus * 8 * 6;      // 8 workgroups of 4 wavefronts per CU resident, six rounds of them
    const size_t max_table = 1024ull << 20;
    uint32_t *table = nullptr, *sink = nullptr;
    uint4 *rec = nullptr;
    CHECK(hipMalloc(&table, max_table));
    CHECK(hipMemset(table, 1, max_table));
    CHECK(hipMalloc(&sink, 64));
    const size_t rec_bytes = (size_t)blocks * 256 * rounds * 16;
    CHECK(hipMalloc(&rec, rec_bytes));
    printf("# CUs %d blocks %d rounds %d record stream %.1f GB\n", cus, blocks, rounds, rec_bytes / 1e9);
    const size_t sizes_mb[] = {2, 8, 32, 64, 80, 96, 128, 192, 256, 512, 1024};
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20;
        run<4, 0>(table, bytes, sink, rec, blocks, rounds);
        run<8, 0>(table, bytes, sink, rec, blocks, rounds);
        run<16, 0>(table, bytes, sink, rec, blocks, rounds);
        run<32, 0>(table, bytes, sink, rec, blocks, rounds);
        run<8, 16>(table, bytes, sink, rec, blocks, rounds);
        run<32, 16>(table, bytes, sink, rec, blocks, rounds);
        run<32, 8>(table, bytes, sink, rec, blocks, rounds);
    }
    return 0;
}
