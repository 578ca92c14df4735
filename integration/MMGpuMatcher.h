// Host side of the alignment seam, in the reference's own types: what a maintainer compiles into MMseqs2 next to
// src/alignment/Matcher.cpp.  MMGpuMatcher is the batch form of Matcher::initQuery + Matcher::getSWResult
// (src/alignment/Matcher.cpp:49-144) for amino-acid sequence queries: the Smith-Waterman scans, the start positions
// and the backtraces come from libmmgpu (include/mmgpu.h) for a whole block of queries at once, everything
// getSWResult and ssw_align_private do around them on the host - composition bias, E-value and coverage gates,
// sequence identity, bit score, the result_t record - is done here with the reference's own functions, so that
// Alignment::run's checkCriteria / sort / resultToBuffer (Alignment.cpp:398-514) continue unchanged.
//
// This file and MMGpuMatcher.cpp include reference headers (Matcher.h, Sequence.h, ...): they are built only where the
// reference tree is present (oracle/Makefile compiles them into the test library and checks them against the real
// Matcher on the CPU, tests/test_integration_host.py); nothing in the GPU library or the GPU tests depends on them.
#ifndef MMGPU_MATCHER_H
#define MMGPU_MATCHER_H

#include <string>
#include <vector>

#include "BaseMatrix.h"
#include "EvalueComputation.h"
#include "Matcher.h"
#include "Sequence.h"

#include "mmgpu.h"

// The device calls the matcher needs, as an interface: the production implementation (MMGpuDeviceBackend.cpp) is a
// thin layer over the C-ABI; the CPU test of the host logic plugs in the reference's own scalar path.
class MMGpuAlignBackend {
public:
    virtual ~MMGpuAlignBackend() {}
    // forward scan of every (query, target) pair, reverse scan (start positions) for the pairs reaching the query's
    // min_start_score when mode == MMGPU_SW_START; out = sum of n_targets records, query-major, list order
    virtual int align(const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t nQueries, int mode,
                      mmgpu_sw_hit *out) = 0;
    // backtraces (banded_sw + walk) of pairs of the last align() call, by index into its result array
    virtual int traceback(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_bt *info, std::string &strings) = 0;
    virtual const char *lastError() = 0;
};

class MMGpuMatcher {
public:
    struct Target {
        unsigned int id;        // id in the resident target database (SequenceLookup order)
        DBKeyType dbKey;        // Sequence::getDbKey() of the target
        int length;             // Sequence::L
        const unsigned char *numSequence;   // needed for identity hits only (may be NULL otherwise)
        bool isIdentity;        // Alignment.cpp:360-365: same key as the query and self-hit handling on
    };
    struct Query {
        Sequence *seq;
        std::vector<Target> targets;   // the prefilter list, in list order
    };

    MMGpuMatcher(MMGpuAlignBackend *backend, BaseMatrix *m, EvalueComputation *evaluer, bool aaBiasCorrection,
                 float aaBiasCorrectionScale, int gapOpen, int gapExtend);

    // One call per block of queries; results[q][k] is what Matcher::getSWResult returns for queries[q].targets[k]
    // (diagonal unused, isReverse false, wrappedScoring false, correlationScoreWeight 0).
    // Returns false (message in error()) if the device call failed.
    bool alignBlock(const std::vector<Query> &queries, int covMode, float covThr, double evalThr, unsigned int alignmentMode,
                    unsigned int seqIdMode, std::vector<std::vector<Matcher::result_t> > &results);

    const std::string &error() const { return err; }

    // smallest raw score whose E-value passes evalThr for a query of this length (ssw_align_private's gate,
    // StripedSmithWaterman.cpp:857-863); 32768 if none does
    int minScoreForEvalue(double evalThr, int queryLength) const;

private:
    MMGpuAlignBackend *backend;
    BaseMatrix *m;
    EvalueComputation *evaluer;
    bool aaBiasCorrection;
    float aaBiasCorrectionScale;
    int gapOpen, gapExtend;
    std::vector<int8_t> tinySubMat;      // Matcher::setSubstitutionMatrix, Matcher.cpp:29-36
    std::vector<int16_t> subMat16;
    std::string err;
};

#endif
