// Micro-probe: issue rate of the VALU instructions the Smith-Waterman inner loop is made of, on gfx950.
// Each kernel runs a long unrolled stream of one instruction over 8 independent register chains, 256 threads x
// (CUs * 8) blocks, and reports lane-ops per second.  Used to set the VALU roofline in DESIGN.md / bench.py.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate scripts/probes/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define PROBE(NAME, ASM)                                                                      \
    __global__ void NAME(unsigned *out, int iters, unsigned s) {                              \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, b = s;                                                          \
        for (int i = 0; i < iters; ++i) {                                                     \
            REP16(asm volatile(ASM "\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
        }                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;  \
    }

#define OP8(op, suffix)                                     \
    op " %0, %0, %8" suffix "\n" op " %1, %1, %8" suffix "\n" \
    op " %2, %2, %8" suffix "\n" op " %3, %3, %8" suffix "\n" \
    op " %4, %4, %8" suffix "\n" op " %5, %5, %8" suffix "\n" \
    op " %6, %6, %8" suffix "\n" op " %7, %7, %8" suffix

PROBE(k_pk_add_i16, OP8("v_pk_add_i16", " clamp"))
PROBE(k_pk_max_i16, OP8("v_pk_max_i16", ""))
PROBE(k_pk_max_u16, OP8("v_pk_max_u16", ""))
PROBE(k_pk_sub_u16, OP8("v_pk_sub_u16", " clamp"))
PROBE(k_max_i32, OP8("v_max_i32", ""))
PROBE(k_add_u32, OP8("v_add_u32", ""))
PROBE(k_sub_u32_clamp, OP8("v_sub_u32_e64", " clamp"))
PROBE(k_and_b32, OP8("v_and_b32", ""))
PROBE(k_fma_f32, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                 "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8")
PROBE(k_max3_i32, "v_max3_i32 %0, %0, %8, %1\nv_max3_i32 %1, %1, %8, %2\nv_max3_i32 %2, %2, %8, %3\nv_max3_i32 %3, %3, %8, %4\n"
                  "v_max3_i32 %4, %4, %8, %5\nv_max3_i32 %5, %5, %8, %6\nv_max3_i32 %6, %6, %8, %7\nv_max3_i32 %7, %7, %8, %0")
PROBE(k_perm_b32, "v_perm_b32 %0, %0, %8, %1\nv_perm_b32 %1, %1, %8, %2\nv_perm_b32 %2, %2, %8, %3\nv_perm_b32 %3, %3, %8, %4\n"
                  "v_perm_b32 %4, %4, %8, %5\nv_perm_b32 %5, %5, %8, %6\nv_perm_b32 %6, %6, %8, %7\nv_perm_b32 %7, %7, %8, %0")
PROBE(k_bfi_b32, "v_bfi_b32 %0, %8, %0, %1\nv_bfi_b32 %1, %8, %1, %2\nv_bfi_b32 %2, %8, %2, %3\nv_bfi_b32 %3, %8, %3, %4\n"
                 "v_bfi_b32 %4, %8, %4, %5\nv_bfi_b32 %5, %8, %5, %6\nv_bfi_b32 %6, %8, %6, %7\nv_bfi_b32 %7, %8, %7, %0")
PROBE(k_add3_u32, "v_add3_u32 %0, %0, %8, %1\nv_add3_u32 %1, %1, %8, %2\nv_add3_u32 %2, %2, %8, %3\nv_add3_u32 %3, %3, %8, %4\n"
                  "v_add3_u32 %4, %4, %8, %5\nv_add3_u32 %5, %5, %8, %6\nv_add3_u32 %6, %6, %8, %7\nv_add3_u32 %7, %7, %8, %0")
PROBE(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %8, %8\nv_pk_fma_f16 %1, %1, %8, %8\nv_pk_fma_f16 %2, %2, %8, %8\nv_pk_fma_f16 %3, %3, %8, %8\n"
                    "v_pk_fma_f16 %4, %4, %8, %8\nv_pk_fma_f16 %5, %5, %8, %8\nv_pk_fma_f16 %6, %6, %8, %8\nv_pk_fma_f16 %7, %7, %8, %8")
PROBE(k_pk_add_u16, OP8("v_pk_add_u16", ""))
PROBE(k_pk_min_i16, OP8("v_pk_min_i16", ""))
PROBE(k_max_i16, OP8("v_max_i16", ""))
PROBE(k_sad_u16, "v_sad_u16 %0, %0, %8, %1\nv_sad_u16 %1, %1, %8, %2\nv_sad_u16 %2, %2, %8, %3\nv_sad_u16 %3, %3, %8, %4\n"
                 "v_sad_u16 %4, %4, %8, %5\nv_sad_u16 %5, %5, %8, %6\nv_sad_u16 %6, %6, %8, %7\nv_sad_u16 %7, %7, %8, %0")

typedef void (*kern_t)(unsigned *, int, unsigned);

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8, threads = 256, iters = 4000;
    unsigned *d;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    struct { const char *name; kern_t k; } ks[] = {
        {"v_pk_add_i16 clamp", k_pk_add_i16}, {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_max_u16", k_pk_max_u16},
        {"v_pk_sub_u16 clamp", k_pk_sub_u16}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_min_i16", k_pk_min_i16},
        {"v_max_i16", k_max_i16}, {"v_max_i32", k_max_i32}, {"v_add_u32", k_add_u32}, {"v_sub_u32 clamp", k_sub_u32_clamp},
        {"v_and_b32", k_and_b32}, {"v_max3_i32", k_max3_i32}, {"v_add3_u32", k_add3_u32}, {"v_perm_b32", k_perm_b32},
        {"v_bfi_b32", k_bfi_b32}, {"v_mov_b32_dpp row_shr:1", k_mov_dpp}, {"v_sad_u16", k_sad_u16},
        {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f16", k_pk_fma_f16},
    };
    printf("device: %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    printf("%-26s %12s %14s %16s\n", "instruction", "ms", "Tlane-op/s", "cyc/wave-instr@2.4GHz/SIMD");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(threads), 0, 0, d, 10, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(threads), 0, 0, d, iters, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_thread = (double)iters * 16 * 8;
        const double lane_ops = instr_per_thread * blocks * threads;
        const double rate = lane_ops / (ms * 1e-3);
        // wave-instructions per SIMD per second -> cycles per wave-instruction at 2.4 GHz
        const double wave_instr_per_simd = lane_ops / 64.0 / (cus * 4.0);
        const double cyc = (ms * 1e-3) * 2.4e9 / wave_instr_per_simd;
        printf("%-26s %12.3f %14.2f %16.2f\n", k.name, ms, rate / 1e12, cyc);
    }
    return 0;
}
