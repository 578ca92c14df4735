// tantan repeat masking of the resident targets (the masking step of IndexBuilder::fillDatabase,
// src/prefiltering/IndexBuilder.cpp:148 -> Masker::maskSequence, src/commons/Masker.cpp:14-57 -> tantan::maskSequences,
// lib/tantan/tantan.cpp:469-487), on the device.
//
// What it computes: the forward - backward recursion of tantan's hidden Markov model (one background state, 50 repeat
// states, no gap states: firstGapProb = 0, the code paths of tantan.cpp:300-391) and, per letter, the repeat probability
// that decides the mask - as the reference's AVX2 + FMA build computes them, bit for bit: the sums over the repeat states
// in four interleaved lanes folded (l0 + l2) + (l1 + l3) with the last states added one by one, and fused multiply-adds
// exactly where gcc contracts them (oracle/tantan_oracle.c documents how that was established; tests/test_tantan.py compares
// with the compiled reference).  `#pragma clang fp contract(off)` keeps the compiler from adding fusions of its own.
//
// Mapping: the model is a serial recursion over the letters of ONE sequence with 50 independent states - and a database
// is a million independent sequences.  One LANE per sequence: the 50 state probabilities of its sequence live in the
// lane's registers (100 VGPRs), the 64 lanes of a wavefront advance through their sequences in step (sequences are
// dealt to wavefronts in order of length, so the lanes of a wavefront end together); no cross-lane operation anywhere.
// The last 52 letters of the sequence are a byte shift register in 13 VGPRs (the letter at distance i is one v_bfe),
// the 21 x 21 likelihood ratios sit in LDS (one ds_read_b64 per state and letter).  Per letter and pass about 350
// instructions for 64 sequences; HBM traffic is the sequence bytes (read twice) plus 4 B per letter of forward
// probabilities written and read back in a lane-interleaved layout (coalesced) - about 10 B per residue.
#include <type_traits>
#include <utility>

#include "mmgpu_internal.h"

#pragma clang fp contract(off)

namespace mmgpu {

namespace {

constexpr int TT_STATES = 50;      // maxRepeatOffset (Masker.cpp:24)
constexpr int TT_HIST = 13;        // 52 letters
constexpr int TT_SCALE_STEP = 16;  // tantan.cpp:58

struct TtConst { double b2b, f2b, f2f0; };

// fn(integral_constant<int, I>) for I = 0 .. N - 1, in order, fully unrolled (the state array must stay in registers)
template <int I, int N, typename F>
__device__ __forceinline__ void tt_each(F &&fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        tt_each<I + 1, N>(fn);
    }
}

// letter at distance i (the one i + 1 positions back) from the shift register
template <int I>
__device__ __forceinline__ unsigned tt_letter(const unsigned (&h)[TT_HIST]) {
    return __builtin_amdgcn_ubfe(h[I / 4], (unsigned)(I % 4) * 8u, 8u);
}

// one forward step over the first MAXO states (calcForwardTransitionAndEmissionProbs, tantan.cpp:300-348); FULL: MAXO = 50
template <bool FULL>
__device__ __forceinline__ double tt_forward(double (&fg)[TT_STATES], const unsigned (&h)[TT_HIST], const double *lr_row, const double *b2f,
                                             double b, int max_offset, const TtConst &K) {
    double lane0 = 0, lane1 = 0, lane2 = 0, lane3 = 0;
    const int vec_end = FULL ? (TT_STATES & ~3) : (max_offset & ~3);
    auto upd = [&](auto ic, double f) {
        constexpr int I = decltype(ic)::value;
        return __builtin_fma(b, b2f[I], f * K.f2f0) * lr_row[tt_letter<I>(h)];
    };
    auto vec = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if (FULL ? I < (TT_STATES & ~3) : I < vec_end) {
            const double f = fg[I];
            if ((I & 3) == 0) lane0 += f;
            else if ((I & 3) == 1) lane1 += f;
            else if ((I & 3) == 2) lane2 += f;
            else lane3 += f;
            fg[I] = upd(ic, f);
        }
    };
    tt_each<0, TT_STATES>(vec);
    double sum = (lane0 + lane2) + (lane1 + lane3);      // simdHorizontalAddDbl (mcf_simd.h:175-179)
    auto tail = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if (FULL ? I >= (TT_STATES & ~3) : (I >= vec_end && I < max_offset)) {
            const double f = fg[I];
            sum += f;
            fg[I] = upd(ic, f);
        }
    };
    tt_each<0, TT_STATES>(tail);
    return __builtin_fma(b, K.b2b, sum * K.f2b);
}

// one backward step (calcEmissionAndBackwardTransitionProbs, tantan.cpp:350-391)
template <bool FULL>
__device__ __forceinline__ double tt_backward(double (&fg)[TT_STATES], const unsigned (&h)[TT_HIST], const double *lr_row, const double *b2f,
                                              double bg, int max_offset, const TtConst &K) {
    const double to_bg = K.f2b * bg;
    double lane0 = 0, lane1 = 0, lane2 = 0, lane3 = 0;
    const int vec_end = FULL ? (TT_STATES & ~3) : (max_offset & ~3);
    auto vec = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if (FULL ? I < (TT_STATES & ~3) : I < vec_end) {
            const double f = fg[I] * lr_row[tt_letter<I>(h)];
            if ((I & 3) == 0) lane0 = __builtin_fma(b2f[I], f, lane0);
            else if ((I & 3) == 1) lane1 = __builtin_fma(b2f[I], f, lane1);
            else if ((I & 3) == 2) lane2 = __builtin_fma(b2f[I], f, lane2);
            else lane3 = __builtin_fma(b2f[I], f, lane3);
            fg[I] = __builtin_fma(f, K.f2f0, to_bg);
        }
    };
    tt_each<0, TT_STATES>(vec);
    double to_fg = (lane0 + lane2) + (lane1 + lane3);
    auto tail = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if (FULL ? I >= (TT_STATES & ~3) : (I >= vec_end && I < max_offset)) {
            const double f = fg[I] * lr_row[tt_letter<I>(h)];
            to_fg = __builtin_fma(b2f[I], f, to_fg);
            fg[I] = __builtin_fma(f, K.f2f0, to_bg);
        }
    };
    tt_each<0, TT_STATES>(tail);
    return __builtin_fma(bg, K.b2b, to_fg);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void tantan_mask_kernel(TantanArgs A) {
    __shared__ double s_lr[32 * 32];
    __shared__ double s_b2f[TT_STATES];
    const int lane = (int)threadIdx.x;
    for (int k = lane; k < A.alphabet * A.alphabet; k += 64) s_lr[(k / A.alphabet) * 32 + (k % A.alphabet)] = A.lr[k];
    for (int k = lane; k < TT_STATES; k += 64) s_b2f[k] = A.b2f[k];
    __syncthreads();
    const uint32_t wave = blockIdx.x;
    const uint32_t slot = wave * 64u + (uint32_t)lane;
    const bool have = slot < A.n;
    const uint32_t t = have ? A.order[slot] : 0u;
    const int len = have ? (int)A.t_len[t] : 0;
    const uint8_t *seq = A.t_res + (size_t)A.t_off4[t] * 4;
    uint8_t *out = A.out_res + (size_t)A.t_off4[t] * 4;
    int max_len = len;
    for (int d = 1; d < 64; d <<= 1) max_len = max(max_len, __shfl_xor(max_len, d));
    max_len = __builtin_amdgcn_readfirstlane(max_len);
    float *probs = A.probs + A.wave_prob_base[wave] + lane;          // [pos][64]
    double *scale = A.scales + A.wave_scale_base[wave] + lane;       // [pos / 16][64]
    TtConst K;
    K.b2b = 1 - A.repeat_prob;
    K.f2b = A.repeat_end_prob;
    K.f2f0 = 1 - A.repeat_end_prob;

    double fg[TT_STATES];
#pragma unroll
    for (int i = 0; i < TT_STATES; i++) fg[i] = 0.0;
    unsigned h[TT_HIST];
#pragma unroll
    for (int i = 0; i < TT_HIST; i++) h[i] = 0u;
    double bg = 1.0, z = 1.0;
    // ---- forward (calcRepeatProbs, tantan.cpp:419-430) ----
    for (int pos = 0; pos < max_len; pos++) {
        if (pos < len) {
            const unsigned c = seq[pos];
            const double *row = s_lr + c * 32;
            if (pos >= TT_STATES) bg = tt_forward<true>(fg, h, row, s_b2f, bg, TT_STATES, K);
            else bg = tt_forward<false>(fg, h, row, s_b2f, bg, pos, K);
            if (pos % TT_SCALE_STEP == TT_SCALE_STEP - 1) {      // rescaleForward
                const double s = 1 / bg;
                scale[(size_t)(pos / TT_SCALE_STEP) * 64] = s;
                bg *= s;
#pragma unroll
                for (int i = 0; i < TT_STATES; i++) fg[i] *= s;
            }
            probs[(size_t)pos * 64] = (float)bg;
            // the letter joins the history: h[i] = seq[pos - i] for the next position
#pragma unroll
            for (int d = TT_HIST - 1; d > 0; d--) h[d] = __builtin_amdgcn_alignbit(h[d], h[d - 1], 24);
            h[0] = (h[0] << 8) | c;
            if (pos == len - 1) {      // forwardTotal (tantan.cpp:140-146)
                double total = 0.0;
#pragma unroll
                for (int i = 0; i < TT_STATES; i++) total += fg[i];
                z = __builtin_fma(K.f2b, total, bg * K.b2b);
            }
        }
    }
    // ---- backward (tantan.cpp:434-448) ----
    uint32_t masked = 0;
    for (int pos = max_len - 1; pos >= 0; pos--) {
        if (pos < len) {
            if (pos == len - 1) {      // initializeBackwardAlgorithm
                bg = K.b2b;
#pragma unroll
                for (int i = 0; i < TT_STATES; i++) fg[i] = K.f2b;
            }
            // the history moves back by one letter: h[0] (= seq[pos]) leaves, seq[pos - 52] enters at the far end
            const unsigned c = h[0] & 0xFFu;
            const int j = pos - 4 * TT_HIST;
            const unsigned incoming = j >= 0 ? (unsigned)seq[j] : 0u;
#pragma unroll
            for (int d = 0; d + 1 < TT_HIST; d++) h[d] = __builtin_amdgcn_alignbit(h[d + 1], h[d], 8);
            h[TT_HIST - 1] = (h[TT_HIST - 1] >> 8) | (incoming << 24);
            const double non_repeat = (double)probs[(size_t)pos * 64] * bg / z;
            const float p = 1 - (float)non_repeat;
            if ((double)p >= A.min_mask_prob) {      // maskProbableLetters (tantan.cpp:498-512), then Masker::finalizeMasking
                out[pos] = A.mask_letter;
                masked++;
            }
            if (pos % TT_SCALE_STEP == TT_SCALE_STEP - 1) {      // rescaleBackward
                const double s = scale[(size_t)(pos / TT_SCALE_STEP) * 64];
                bg *= s;
#pragma unroll
                for (int i = 0; i < TT_STATES; i++) fg[i] *= s;
            }
            const double *row = s_lr + c * 32;
            if (pos >= TT_STATES) bg = tt_backward<true>(fg, h, row, s_b2f, bg, TT_STATES, K);
            else bg = tt_backward<false>(fg, h, row, s_b2f, bg, pos, K);
        }
    }
    for (int d = 1; d < 64; d <<= 1) masked += __shfl_xor(masked, d);
    if (lane == 0 && masked) atomicAdd(A.n_masked, (unsigned long long)masked);
}

}  // namespace

hipError_t launch_tantan_mask(const TantanArgs &A, hipStream_t s) {
    if (A.n == 0) return hipSuccess;
    hipLaunchKernelGGL(tantan_mask_kernel, dim3((A.n + 63) / 64), dim3(64), 0, s, A);
    return hipGetLastError();
}

// first use of any kernel of this file loads its code object (tens of milliseconds): mmgpu_warmup does it ahead of time
void warm_tantan() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&tantan_mask_kernel));
}

}  // namespace mmgpu
