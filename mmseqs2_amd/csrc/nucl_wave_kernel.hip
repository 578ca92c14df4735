// Nucleotide alignment step, one wavefront per alignment with the DP state in registers: see nucl_wave.h for the mapping.
// Cross-lane primitives on gfx950:
//   NUCL_ROR1_U32  "value of lane - 1, lane 0 takes lane 63": DPP wave_ror:1 (one VALU move, no LDS crossbar)
//   NUCL_READLANE  v_readlane_b32 with a wave-uniform lane (the band positions are functions of the loop counter)
//   NUCL_SHFL*     ds_bpermute / swizzle through __shfl (maximum reduction, prefix sums: off the per-cell chain)
#define NUCL_NG 64
#define NUCL_NS nucl64
#define NUCL_HD __device__ __forceinline__
#define NUCL_LANE() ((int)(threadIdx.x & 63u))
#define NUCL_SHFL(v, src) __shfl((v), (src), 64)
#define NUCL_SHFL_XOR(v, mask) __shfl_xor((v), (mask), 64)
#define NUCL_SHFL_U64(v, src) ((unsigned long long)__shfl((long long)(v), (src), 64))
#define NUCL_SYNC()                                                     \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local"); \
        __builtin_amdgcn_wave_barrier();                                \
    } while (0)
#define NUCL_SYNC_MEM()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
#define NUCL_ATOMIC_ADD_U32(p, v) atomicAdd((p), (v))
#define NUCL_ATOMIC_ADD_U64(p, v) atomicAdd((p), (v))
#ifdef MMGPU_NUCL_NO_DPP
#define NUCL_ROR1_U32(v) ((unsigned)__shfl((int)(v), (NUCL_LANE() + 63) & 63, 64))
#else
// DPP_WF_RR1 = 0x13C: wave_ror:1, every lane reads its left neighbour, lane 0 reads lane 63
#define NUCL_ROR1_U32(v) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), 0x13C, 0xF, 0xF, false))
#endif
#define NUCL_READLANE(v, l) __builtin_amdgcn_readlane((int)(v), (l))
#include "mmgpu_internal.h"   // pulls in nucl_core.h with the macros above (SeqView, seeds, band_of ...)
// Wave reduction on DPP: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes (lanes without a source keep their own value:
// old = v, op(v, v) = v), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lane 63 holds the result.
#define NUCL_DPP_STEP(v, OP, ctrl, rmask)                                                  \
    do {                                                                                   \
        const int o__ = __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rmask), 0xF, false); \
        (v) = OP((v), o__);                                                                \
    } while (0)
static __device__ __forceinline__ int nucl_max_i32(int a, int b) { return a > b ? a : b; }
static __device__ __forceinline__ int nucl_min_u32(int a, int b) { return (unsigned)a < (unsigned)b ? a : b; }
static __device__ __forceinline__ int nucl_wave_reduce_max(int v) {
    NUCL_DPP_STEP(v, nucl_max_i32, 0x111, 0xF);
    NUCL_DPP_STEP(v, nucl_max_i32, 0x112, 0xF);
    NUCL_DPP_STEP(v, nucl_max_i32, 0x114, 0xF);
    NUCL_DPP_STEP(v, nucl_max_i32, 0x118, 0xF);
    NUCL_DPP_STEP(v, nucl_max_i32, 0x142, 0xA);
    NUCL_DPP_STEP(v, nucl_max_i32, 0x143, 0xC);
    return __builtin_amdgcn_readlane(v, 63);
}
static __device__ __forceinline__ unsigned nucl_wave_reduce_minu(unsigned u) {
    int v = (int)u;
    NUCL_DPP_STEP(v, nucl_min_u32, 0x111, 0xF);
    NUCL_DPP_STEP(v, nucl_min_u32, 0x112, 0xF);
    NUCL_DPP_STEP(v, nucl_min_u32, 0x114, 0xF);
    NUCL_DPP_STEP(v, nucl_min_u32, 0x118, 0xF);
    NUCL_DPP_STEP(v, nucl_min_u32, 0x142, 0xA);
    NUCL_DPP_STEP(v, nucl_min_u32, 0x143, 0xC);
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
#ifdef MMGPU_NUCL_NO_DPP
static __device__ __forceinline__ int nucl_wave_reduce_max_shfl(int v) {
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
static __device__ __forceinline__ unsigned nucl_wave_reduce_minu_shfl(unsigned v) {
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, d, 64); v = o < v ? o : v; }
    return v;
}
#define NUCL_WAVE_MAX_I32(v) nucl_wave_reduce_max_shfl(v)
#define NUCL_WAVE_MIN_U32(v) nucl_wave_reduce_minu_shfl(v)
#else
#define NUCL_WAVE_MAX_I32(v) nucl_wave_reduce_max(v)
#define NUCL_WAVE_MIN_U32(v) nucl_wave_reduce_minu(v)
#endif
#include "nucl_wave.h"

namespace mmgpu {

namespace {

__global__ __launch_bounds__(256) void nucl_align_wave_kernel(NuclLaunch L) {
    __shared__ nuclw::WaveLds lds[4];
    const int wslot = (int)(threadIdx.x >> 6);
    const size_t slot = (size_t)blockIdx.x * 4 + (size_t)wslot;
    nuclw::align_wave(L, lds[wslot], L.pscratch + slot * L.pscratch_stride, L.wscratch + slot * L.wscratch_stride);
}

}  // namespace

hipError_t launch_nucl_align_wave(const NuclLaunch &L, unsigned blocks, hipStream_t stream) {
    if (L.n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(nucl_align_wave_kernel, dim3(blocks), dim3(256), 0, stream, L);
    return hipGetLastError();
}

}  // namespace mmgpu
