// Nucleotide alignment step for gfx950: BandedNucleotideAligner::align (src/alignment/BandedNucleotideAligner.cpp:76-263)
// = ungapped seed on the prefilter diagonal (DistanceCalculator.h:93-200), left extension on the reversed sequences
// (score only), right extension with CIGAR, both by ksw_extz2_sse (lib/ksw2/ksw2_extz2_sse.cpp:44-285, band 64, z-drop),
// backtrack (lib/ksw2/ksw2.h:134-173) - bit for bit (score, start / end positions, backtrace string, identities).
//
// ksw_extz2_sse is an anti-diagonal DP on 8-bit differences (u, v, x, y) processed in blocks of 16 target positions;
// a block is computed whole even where it sticks out of the band, the cells outside use what the byte arrays hold
// there (stale values, the allocation's zeros), and the band later moves over some of them - so the result depends
// on the block structure.  The mapping keeps it: ONE 16-LANE GROUP (a DPP row) PER ALIGNMENT, lane = byte of the
// reference's 128-bit vector.  The five byte arrays and the 32-bit score row live in LDS as a window of 256 target
// positions (the band plus its block padding spans < 9 blocks; a block is zeroed when the window first reaches it,
// which is what the reference's calloc gives it), blocks of an anti-diagonal are walked from the highest to the lowest
// so that "cell t reads x[t-1], v[t-1] of the previous anti-diagonal" needs no carry, the exact maximum of the band
// (the reference's four-lane scan order decides ties) is a 16-lane shuffle reduction, the direction bytes go to a
// per-group scratch in HBM and are walked back by lane 0.  Work per alignment ~ (qlen + tlen) x 7 blocks, three
// passes at most; bounded by the dependent chain over the anti-diagonals (LDS latency), not by HBM or VALU peak -
// parallelism comes from the alignments (4 per wavefront, pulled from a length-sorted queue).
#ifndef NUCL_NG
#define NUCL_NG 16
#define NUCL_NS nucl16
#define NUCL_LAUNCH launch_nucl_align
#endif
#define NUCL_HD __device__ __forceinline__
#define NUCL_LANE() ((int)(threadIdx.x & (unsigned)(NUCL_NG - 1)))
#define NUCL_SHFL(v, src) __shfl((v), (src), NUCL_NG)
#define NUCL_SHFL_XOR(v, mask) __shfl_xor((v), (mask), NUCL_NG)
#define NUCL_SHFL_U64(v, src) ((unsigned long long)__shfl((long long)(v), (src), NUCL_NG))
// the lanes of a group run in lock step: a phase boundary only has to keep the memory operations in order.  The LDS
// form must not wait for the direction bytes still on their way to HBM (a full fence per phase made the kernel 20x
// slower than its LDS chain).
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
#include "mmgpu_internal.h"   // pulls in nucl_core.h with the macros above

namespace mmgpu {

namespace {

__global__ __launch_bounds__(256) void nucl_align_kernel(NuclLaunch L) {
    __shared__ NUCL_NS::GroupLds lds[256 / NUCL_NS::NG];
    const int gslot = (int)(threadIdx.x / NUCL_NS::NG);
    const size_t slot = (size_t)blockIdx.x * (256 / NUCL_NS::NG) + (size_t)gslot;
    NUCL_NS::align_group(L, lds[gslot], L.pscratch + slot * L.pscratch_stride, L.wscratch + slot * L.wscratch_stride);
}

}  // namespace

hipError_t NUCL_LAUNCH(const NuclLaunch &L, unsigned blocks, hipStream_t stream) {
    if (L.n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(nucl_align_kernel, dim3(blocks), dim3(256), 0, stream, L);
    return hipGetLastError();
}

}  // namespace mmgpu
