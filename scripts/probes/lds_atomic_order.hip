// Probe: are the lanes of ONE LDS atomic instruction that hit the same address applied in ascending lane order?
// (ds_mskor_rtn_b32 as a byte-granular atomic exchange, ds_or_rtn_b32 as test-and-set)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
__device__ __forceinline__ uint32_t mskor_rtn(uint32_t *addr, uint32_t mask, uint32_t val) {
    uint32_t old;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"((uint32_t)(uintptr_t)addr), "v"(mask), "v"(val) : "memory");
    return old;
}
__global__ __launch_bounds__(256) void probe(const uint32_t *keys, const uint8_t *vals, uint32_t rounds, uint32_t nkeys, uint8_t *prev_out, uint8_t *bit_out) {
    __shared__ uint32_t S[4][1024];
    __shared__ uint32_t E[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = lane; k < 1024; k += 64) S[wave][k] = 0;
    for (int k = lane; k < 128; k += 64) E[wave][k] = 0;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * rounds * 64;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t key = keys[base + r * 64 + lane];
        const uint32_t d8 = vals[base + r * 64 + lane];
        const bool act = key != 0xFFFFFFFFu;
        uint32_t prev = 0, ob = 0;
        if (act) {
            const uint32_t sh = (key & 3u) * 8u;
            const uint32_t old = mskor_rtn(&S[wave][key >> 2], 0xFFu << sh, d8 << sh);
            prev = (old >> sh) & 0xFFu;
            ob = (atomicOr(&E[wave][key >> 5], 1u << (key & 31u)) >> (key & 31u)) & 1u;
        }
        prev_out[base + r * 64 + lane] = (uint8_t)prev;
        bit_out[base + r * 64 + lane] = (uint8_t)ob;
    }
}
int main() {
    const uint32_t waves = 4096, rounds = 64, nkeys = 4096;
    const size_t n = (size_t)waves * rounds * 64;
    std::vector<uint32_t> keys(n);
    std::vector<uint8_t> vals(n), exp_prev(n), exp_bit(n);
    srand(7);
    for (uint32_t w = 0; w < waves; w++) {
        std::vector<uint8_t> st(nkeys, 0), eb(nkeys, 0);
        const uint32_t range = 1u << (1 + rand() % 12);      // 2 .. 4096 distinct keys: from total conflict to almost none
        for (uint32_t r = 0; r < rounds; r++)
            for (int l = 0; l < 64; l++) {
                const size_t i = ((size_t)w * rounds + r) * 64 + l;
                const bool idle = rand() % 17 == 0;
                keys[i] = idle ? 0xFFFFFFFFu : (uint32_t)rand() % range;
                vals[i] = (uint8_t)(rand() % 5);
                if (!idle) {
                    exp_prev[i] = st[keys[i]]; st[keys[i]] = vals[i];
                    exp_bit[i] = eb[keys[i]]; eb[keys[i]] = 1;
                }
            }
    }
    uint32_t *dk; uint8_t *dv, *dp, *db;
    hipMalloc(&dk, n * 4); hipMalloc(&dv, n); hipMalloc(&dp, n); hipMalloc(&db, n);
    hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, vals.data(), n, hipMemcpyHostToDevice);
    size_t bad_p = 0, bad_b = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(probe, dim3(waves / 4), dim3(256), 0, 0, dk, dv, rounds, nkeys, dp, db);
        std::vector<uint8_t> gp(n), gb(n);
        hipMemcpy(gp.data(), dp, n, hipMemcpyDeviceToHost); hipMemcpy(gb.data(), db, n, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; i++) if (keys[i] != 0xFFFFFFFFu) { bad_p += gp[i] != exp_prev[i]; bad_b += gb[i] != exp_bit[i]; }
    }
    printf("lanes checked %zu x 3, previous-byte mismatches %zu, test-and-set mismatches %zu -> %s\n", n, bad_p, bad_b,
           bad_p == 0 && bad_b == 0 ? "conflicting lanes of one LDS atomic are applied in ascending lane order" : "ORDER NOT LANE-ASCENDING");
    return bad_p || bad_b ? 1 : 0;
}
