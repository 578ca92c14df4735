/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 * Restatement of SubstitutionMatrix::calcLocalAaBiasCorrection
 * (src/commons/SubstitutionMatrix.cpp:79-112): per-position composition bias, window +-20.
 * The float/double mixing below follows the reference expression by expression so the result is
 * bit-identical: `deltaS_i /= -1.0 * (float)windowLength` divides in double and rounds to float;
 * `deltaS_i += pBack[a] * (float)subMat[a]` accumulates in double per term and rounds to float. */
#include <stdint.h>
#include "mm_oracle.h"

void mmo_comp_bias(const int16_t *submat, const double *pback, int alphabet, const uint8_t *seq, int n,
                   float scale, float *out) {
    const int windowSize = 40;
    for (int i = 0; i < n; i++) {
        const int minPos = (i - windowSize / 2) > 0 ? (i - windowSize / 2) : 0;
        const int maxPos = (i + windowSize / 2) < n ? (i + windowSize / 2) : n;
        const int windowLength = maxPos - minPos;
        int sum = 0;
        const int16_t *row = submat + (int)seq[i] * alphabet;
        for (int j = minPos; j < maxPos; j++) sum += row[seq[j]];
        sum -= row[seq[i]];
        float d = (float)sum;
        d = (float)((double)d / (-1.0 * (double)(float)windowLength));
        for (int a = 0; a < alphabet; a++) d = (float)((double)d + pback[a] * (double)(float)row[a]);
        out[i] = scale * d;
    }
}
