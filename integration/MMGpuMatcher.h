// Host side of the alignment seam, in the reference's own types: what a maintainer compiles into MMseqs2 next to
// src/alignment/Matcher.cpp.  MMGpuMatcher is the batch form of Matcher::initQuery + Matcher::getSWResult
// (src/alignment/Matcher.cpp:49-144) for amino-acid sequence and profile queries against amino-acid targets: the Smith-Waterman scans, the start positions
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

#include <cstring>
#include <string>
#include <vector>

#include "BaseMatrix.h"
#include "EvalueComputation.h"
#include "Matcher.h"
#include "Sequence.h"
#include "StripedSmithWaterman.h"

#include "mmgpu.h"

// The device calls the matcher needs, as an interface: the production implementation (MMGpuDeviceBackend.cpp) is a
// thin layer over the C-ABI; the CPU test of the host logic plugs in the reference's own scalar path.
class MMGpuAlignBackend {
public:
    virtual ~MMGpuAlignBackend() {}
    // forward scan of every (query, target) pair, reverse scan (start positions) for the pairs reaching the query's
    // min_start_score when mode >= MMGPU_SW_START (MMGPU_SW_START_NOT_WORD: of those, the hits of the uint8 pass only); out = sum of n_targets records, query-major, list order
    virtual int align(const mmgpu_sw_params *params, const mmgpu_sw_query *queries, uint32_t nQueries, int mode,
                      mmgpu_sw_hit *out) = 0;
    // backtraces (banded_sw + walk) of pairs of the last align() call, by index into its result array
    virtual int traceback(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_bt *info, std::string &strings) = 0;
    // the block aligner's start positions / identities / backtraces of int16-range pairs of the last align() call
    // (mmgpu_sw_block_backtrace); the default says "not here" for every pair, which sends them to the host's block aligner.
    // want: BLOCK_STRINGS = everything; BLOCK_IDENT = start positions, identities and lengths (a run that writes no backtraces);
    // BLOCK_STARTS = start positions only (alignment mode 2 without backtraces: Matcher.cpp:107-127 reads nothing else)
    enum { BLOCK_STARTS = 0, BLOCK_IDENT = 1, BLOCK_STRINGS = 2 };
    virtual int blockBacktrace(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_block *out, std::string &strings, int want = BLOCK_STRINGS) {
        (void)pairIndex;
        (void)want;
        strings.clear();
        for (uint32_t k = 0; k < n; k++) { memset(&out[k], 0, sizeof(out[k])); out[k].status = MMGPU_BLOCK_TOO_LARGE; }
        return 0;
    }
    // align() was called with MMGPU_SW_START_NOT_WORD (only a backend with a block aligner is): the reverse scan of these pairs of
    // the last align() call after the fact - the ones the block aligner declined (StripedSmithWaterman.cpp:873-882); out[k] = the
    // pair's record with its start position
    virtual int reversePairs(const uint32_t *pairIndex, uint32_t n, mmgpu_sw_hit *out) {
        (void)pairIndex; (void)n; (void)out;
        return -1;
    }
    virtual const char *lastError() = 0;
};

// Optional host hook for the pairs whose score left the uint8 range (s_align::word == 1).  The stock reference takes
// start positions, backtrace and identities of those pairs from the Rust block-aligner
// (alignStartPosBacktraceBlock, StripedSmithWaterman.cpp:865-882,943-1127) and only falls back to its own reverse scan +
// banded_sw when that fails.  A build that links the real crate installs a hook that calls the host's own
// alignStartPosBacktraceBlock with the score / end positions the device computed, so that those fields are the stock
// binary's by construction; run() returns false when the block aligner declines (r.score1 == UINT32_MAX), in which
// case the device's fallback results are used exactly like the reference's fallback.  Called from OpenMP threads.
class MMGpuBlockBacktracer {
public:
    virtual ~MMGpuBlockBacktracer() {}
    virtual bool run(unsigned int thread, size_t queryIndex, const unsigned char *query, int queryLength,
                     const unsigned char *target, int targetLength, s_align &a, std::string &backtrace) = 0;
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
        const unsigned char *numSequence;   // Sequence::numSequence of the query (the caller keeps it alive)
        int L;                              // Sequence::L
        std::vector<Target> targets;        // the prefilter list, in list order
        // profile query (DBTYPE_HMM_PROFILE): Sequence::getAlignmentProfile(), [Sequence::PROFILE_AA_SIZE][L], kept alive by
        // the caller; numSequence is then the consensus sequence.  NULL: sequence query.  (Matcher::initQuery, Matcher.cpp:49-60)
        const int8_t *profile;
        // E-value threshold of this query when it differs from the block's (the LCA pass aligns every query of a block under
        // its own top hit's E-value, Alignment.cpp:456,483); negative: the block's
        double evalThr;
        Query() : numSequence(NULL), L(0), profile(NULL), evalThr(-1.0) {}
    };

    MMGpuMatcher(MMGpuAlignBackend *backend, BaseMatrix *m, EvalueComputation *evaluer, bool aaBiasCorrection,
                 float aaBiasCorrectionScale, int gapOpen, int gapExtend);

    // One call per block of queries; results[q][k] is what Matcher::getSWResult returns for queries[q].targets[k]
    // (diagonal unused, isReverse false, wrappedScoring false, correlationScoreWeight 0).
    // Returns false (message in error()) if the device call failed.  Pairs whose backtrace the device declined
    // (MMGPU_BT_TOO_LARGE / MMGPU_BT_FAILED) are listed in refusedPairs as (query, index into targets): the caller
    // recomputes those results with the host's Matcher::getSWResult (NULL: such a pair is an error).
    bool alignBlock(const std::vector<Query> &queries, int covMode, float covThr, double evalThr, unsigned int alignmentMode,
                    unsigned int seqIdMode, std::vector<std::vector<Matcher::result_t> > &results,
                    std::vector<std::pair<size_t, size_t> > *refusedPairs = NULL);

    const std::string &error() const { return err; }

    // threads of the parallel loops inside alignBlock (Alignment lowers its thread count to the number of queries,
    // Alignment.cpp:128, while the OpenMP default stays at --threads: per-thread state is sized by the former)
    void setThreads(unsigned int n) { numThreads = n; }
    // --corr-score-weight (alignStartPosBacktrace, StripedSmithWaterman.cpp:1249-1253): after the backtrace the score gains
    // weight x the lag-1..4 autocorrelation of the per-column scores of the aligned pairs, the E-value follows
    void setCorrelationScoreWeight(float w) { correlationScoreWeight = w; }

    // targetSequence(id) must return the numeric residues of a resident target (needed by the hook only)
    typedef const unsigned char *(*TargetLookup)(void *ctx, unsigned int id);
    // int16-range pairs go to the device's block aligner first (MMGPU_BLOCK_ALIGNER=device, the default of the patched binary);
    // the host hook, if installed, then only serves what the device declines as too large
    void setDeviceBlockAligner(bool on) { deviceBlockAligner = on; }
    // false: the caller never reads result_t::backtrace (no -a / --realign / --alt-ali): the strings of block-aligned pairs stay on
    // the device; the alignment length of those pairs (Matcher.cpp:111-114) comes from the string's length alone
    void setNeedBacktraceStrings(bool on) { needBacktraceStrings = on; }
    void setBlockBacktracer(MMGpuBlockBacktracer *hook, TargetLookup lookup, void *lookupCtx) {
        blockHook = hook;
        targetLookup = lookup;
        targetLookupCtx = lookupCtx;
    }

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
    unsigned int numThreads;
    float correlationScoreWeight;
    bool deviceBlockAligner;
    bool needBacktraceStrings;
    MMGpuBlockBacktracer *blockHook;
    TargetLookup targetLookup;
    void *targetLookupCtx;
};

#endif
