/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the tantan repeat masker as IndexBuilder::fillDatabase applies it to amino-acid targets (SURVEY.md
 * section 8 row a3; src/prefiltering/IndexBuilder.cpp:148 -> Masker::maskSequence, src/commons/Masker.cpp:14-57 ->
 * tantan::maskSequences, lib/tantan/tantan.cpp:469-487): forward - backward over a hidden Markov model with one background
 * state and maxRepeatOffset = 50 repeat states, no gap states (firstGapProb = 0: the paths of :300-370), a letter is masked
 * when its repeat probability reaches minMaskProb.
 *
 * The probabilities are doubles and the decision compares a float with a threshold, so the result depends on the ORDER and
 * the ROUNDING of every operation.  What is restated is the reference's AVX2 build (-mavx2 -mfma, CMakeLists.txt:76) as gcc
 * compiles it:
 *   * the sums over the repeat states run in four interleaved lanes (SimdDbl = 4 doubles, lib/tantan/mcf_simd.h) over the
 *     multiples of four below maxOffset, are folded (l0 + l2) + (l1 + l3) (simdHorizontalAddDbl, mcf_simd.h:175-179), and the
 *     remaining one to three states are added one by one (:338-345, :377-383);
 *   * gcc contracts a * b + c into one fused multiply-add where the source has that shape.  Which products are fused was read off
 *     the compiled code (g++ 11 -O3 -mavx2 -mfma, the flags oracle/Makefile and the reference's own build use) and is written
 *     out below with fma(); the tests compare this file with that build on thousands of sequences, probabilities bit for bit
 *     (tests/test_tantan.py).  A build without FMA, or a compiler that contracts differently, rounds a few products differently:
 *     its masks differ from this one's only where a repeat probability lies within an ulp of the threshold.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

#define TT_SCALE_STEP 16      /* tantan.cpp:58 scaleStepSize */

static double first_repeat_offset_prob(double probMult, int maxRepeatOffset) {      /* tantan.cpp:38-44 */
    if (probMult < 1 || probMult > 1) return (1 - probMult) / (1 - pow(probMult, maxRepeatOffset));
    return 1.0 / maxRepeatOffset;
}

/* b2fProbs of the Tantan constructor (tantan.cpp:118-131): what the device takes as a table */
void mmo_tantan_b2f(double repeatProb, double decay, int maxRepeatOffset, double *b2f) {
    double p = repeatProb * first_repeat_offset_prob(decay, maxRepeatOffset);
    for (int i = 0; i < maxRepeatOffset; i++) {
        b2f[i] = p;
        p *= decay;
    }
}

/* Tantan::calcRepeatProbs (tantan.cpp:419-452): probs[len] */
void mmo_tantan_probs(const uint8_t *seq, int len, const double *lr, int alph, double repeatProb, double repeatEndProb, double decay,
                      int maxRepeatOffset, float *probs) {
    const double b2b = 1 - repeatProb, f2b = repeatEndProb, f2f0 = 1 - repeatEndProb;
    double *b2f = (double *)malloc(sizeof(double) * (size_t)maxRepeatOffset);
    double *fg = (double *)calloc((size_t)maxRepeatOffset, sizeof(double));
    double *scale = (double *)malloc(sizeof(double) * (size_t)(len / TT_SCALE_STEP + 1));
    mmo_tantan_b2f(repeatProb, decay, maxRepeatOffset, b2f);
    double bg = 1.0;                                            /* initializeForwardAlgorithm */
    for (int pos = 0; pos < len; pos++) {
        /* calcForwardTransitionAndEmissionProbs (:300-348) */
        const double *row = lr + (size_t)seq[pos] * alph;
        const int maxOffset = pos < maxRepeatOffset ? pos : maxRepeatOffset;
        const double b = bg;
        double lane[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i <= maxOffset - 4; i += 4)
            for (int k = 0; k < 4; k++) {
                const double f = fg[i + k];
                lane[k] += f;
                fg[i + k] = fma(b, b2f[i + k], f * f2f0) * row[seq[pos - (i + k) - 1]];
            }
        double from_fg = (lane[0] + lane[2]) + (lane[1] + lane[3]);
        for (; i < maxOffset; i++) {
            const double f = fg[i];
            from_fg += f;
            fg[i] = fma(b, b2f[i], f * f2f0) * row[seq[pos - i - 1]];
        }
        bg = fma(b, b2b, from_fg * f2b);
        if (pos % TT_SCALE_STEP == TT_SCALE_STEP - 1) {          /* rescaleForward (:398-405) */
            const double s = 1 / bg;
            scale[pos / TT_SCALE_STEP] = s;
            bg *= s;
            for (int k = 0; k < maxRepeatOffset; k++) fg[k] *= s;
        }
        probs[pos] = (float)bg;
    }
    double total = 0.0;                                         /* forwardTotal (:140-146) */
    for (int k = 0; k < maxRepeatOffset; k++) total += fg[k];
    const double z = fma(f2b, total, bg * b2b);
    bg = b2b;                                                   /* initializeBackwardAlgorithm */
    for (int k = 0; k < maxRepeatOffset; k++) fg[k] = f2b;
    for (int pos = len - 1; pos >= 0; pos--) {
        const double non_repeat = (double)probs[pos] * bg / z;
        probs[pos] = 1 - (float)non_repeat;
        if (pos % TT_SCALE_STEP == TT_SCALE_STEP - 1) {          /* rescaleBackward */
            const double s = scale[pos / TT_SCALE_STEP];
            bg *= s;
            for (int k = 0; k < maxRepeatOffset; k++) fg[k] *= s;
        }
        /* calcEmissionAndBackwardTransitionProbs (:350-391) */
        const double *row = lr + (size_t)seq[pos] * alph;
        const int maxOffset = pos < maxRepeatOffset ? pos : maxRepeatOffset;
        const double to_bg = f2b * bg;
        double lane[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i <= maxOffset - 4; i += 4)
            for (int k = 0; k < 4; k++) {
                const double f = fg[i + k] * row[seq[pos - (i + k) - 1]];
                lane[k] = fma(b2f[i + k], f, lane[k]);
                fg[i + k] = fma(f, f2f0, to_bg);
            }
        double to_fg = (lane[0] + lane[2]) + (lane[1] + lane[3]);
        for (; i < maxOffset; i++) {
            const double f = fg[i] * row[seq[pos - i - 1]];
            to_fg = fma(b2f[i], f, to_fg);
            fg[i] = fma(f, f2f0, to_bg);
        }
        bg = fma(bg, b2b, to_fg);
    }
    free(b2f); free(fg); free(scale);
}

/* tantan::maskSequences + Masker::finalizeMasking with the Masker's constants (Masker.cpp:22-31): masks seq in place,
 * returns the number of letters masked by tantan */
int mmo_tantan_mask(uint8_t *seq, int len, const double *lr, int alph, double min_mask_prob, uint8_t mask_letter, float *probs_out) {
    float *probs = (float *)malloc(sizeof(float) * (size_t)(len > 0 ? len : 1));
    mmo_tantan_probs(seq, len, lr, alph, 0.005, 0.05, 0.9, 50, probs);
    int masked = 0;
    for (int k = 0; k < len; k++)
        if (probs[k] >= min_mask_prob) {      /* float against double (maskProbableLetters, :498-512) */
            seq[k] = mask_letter;
            masked++;
        }
    if (probs_out) memcpy(probs_out, probs, sizeof(float) * (size_t)len);
    free(probs);
    return masked;
}
