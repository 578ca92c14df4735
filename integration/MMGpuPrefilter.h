// Host side of the prefilter seam, in the reference's own types: what a maintainer compiles into MMseqs2 next to
// src/prefiltering/Prefiltering.cpp.  MMGpuPrefilter takes the place of the per-thread QueryMatcher objects of
// Prefiltering::runSplit (Prefiltering.cpp:826-842): loadIndex() hands the reference's IndexTable / SequenceLookup /
// ScoreMatrix objects to the device once per split, matchBlock() is the batch form of QueryMatcher::matchQuery
// (QueryMatcher.cpp:103-241) and returns hit_t lists in the reference's final order; the loop body after matchQuery
// (:876-917: id -> key, canBeCovered, prefilterHitToBuffer) runs on them unchanged.  Queries the device declines
// (MMGPU_PF_OVERFLOW) are reported so that the caller runs its CPU QueryMatcher for them.
//
// Includes reference headers: built (compile check against both header sets) by oracle/Makefile only where the reference
// tree is present.
#ifndef MMGPU_PREFILTER_H
#define MMGPU_PREFILTER_H

#include <string>
#include <vector>

#include "BaseMatrix.h"
#include "IndexTable.h"
#include "QueryMatcher.h"
#include "ScoreMatrix.h"
#include "Sequence.h"
#include "SequenceLookup.h"

#include "mmgpu.h"

class MMGpuPrefilter {
public:
    MMGpuPrefilter(mmgpu_ctx *gpu, BaseMatrix *kmerSubMat, BaseMatrix *ungappedSubMat, bool aaBiasCorrection,
                   float aaBiasCorrectionScale);
    // Several devices (one group of MMGpuRun::groups()): buildIndex() deals the targets to them by length bucket and builds one index per
    // device, matchBlock() runs every shard, exchanges the lists over the library's communicator and returns the merged lists,
    // which equal the single-device ones.  Sequence queries with diagonal scoring only (the caller checks multiCapable()).
    void useDevices(mmgpu_multi *m) { multi = m; }
    bool usesSeveralDevices() const { return multi != NULL; }
    static bool multiCapable(bool profileQuery, bool nucleotide, bool kmerScoring, size_t maxResListLen, int nDevices) {
        return !profileQuery && !nucleotide && !kmerScoring && maxResListLen * (size_t)nDevices <= 4096;
    }

    // once per target split: SequenceLookup -> resident targets, IndexTable + similar-k-mer tables -> resident index
    bool loadIndex(IndexTable *indexTable, SequenceLookup *sequenceLookup, ScoreMatrix &threeMer, ScoreMatrix &twoMer, bool spacedKmer);

    struct Query {
        const unsigned char *numSequence;   // Sequence::numSequence (the caller keeps it alive)
        int L;
        unsigned int identityId;      // targetSeqId of Prefiltering.cpp:855-868, UINT_MAX = none
        // profile query (Prefiltering.cpp:832-834): copies of Sequence::profile_score / profile_index ([L][profileRow], rows
        // sorted by Sequence::mapProfile) and Sequence::getAlignmentProfile() ([20][L]); all NULL for a sequence query
        const short *profileScore;
        const unsigned int *profileIndex;
        unsigned int profileRow;
        const int8_t *profile;
        // sequence query: compositionBias() of it if the caller has it already (L floats, kept alive by the caller), else NULL
        const float *compBias;
        Query() : numSequence(NULL), L(0), identityId(0xFFFFFFFFu), profileScore(NULL), profileIndex(NULL), profileRow(0), profile(NULL),
                  compBias(NULL) {}
    };
    // results[q] = the hit_t list of QueryMatcher::matchQuery; needsCpu[q] = the device declined the query;
    // stats[q] (optional) = what QueryMatcher::getStatistics() would report for the query
    bool matchBlock(const std::vector<Query> &queries, int kmerThr, size_t maxResListLen, unsigned int minDiagScoreThr,
                    std::vector<std::vector<hit_t> > &results, std::vector<bool> &needsCpu,
                    std::vector<mmgpu_pf_qstat> *stats = NULL);

    // The same in two halves, so that the device has the next block queued while the host collects this one: submitBlock()
    // enqueues (descriptors, uploads, kernels), finishBlock() downloads and converts (and deletes the Pending).  `queries` must
    // stay alive in between.  A block the device cannot take whole is run in pieces by finishBlock().
    struct Pending;
    Pending *submitBlock(const std::vector<Query> &queries, int kmerThr, size_t maxResListLen, unsigned int minDiagScoreThr);
    bool finishBlock(Pending *pending, std::vector<std::vector<hit_t> > &results, std::vector<bool> &needsCpu,
                     std::vector<mmgpu_pf_qstat> *stats = NULL);
    // QueryMatcher::matchQuery's composition bias of a sequence query (QueryMatcher.cpp:109-117): zeros where the reference applies none
    void compositionBias(const Query &query, std::vector<float> &bias) const;

    // the same hand-over without a host index: the index is built on the device from the (masked) SequenceLookup with the
    // k-mer threshold IndexBuilder::fillDatabase would have used (IndexTable.h:146-154); tables may be invalid (exact k-mers)
    // Persisted device layout (include/mmgpu.h, mmgpu_db_save / mmgpu_db_load; the reference's makepaddedseqdb + createindex): one
    // file with the targets, their masked view and the k-mer index as they lie on the device.  The fingerprints say what it was made
    // from: the target database (keys, lengths, a sample of its bytes - read off the DBReader, no sequence is mapped for it) and the
    // index parameters.
    struct Persisted {
        std::string path;
        uint64_t sourceFp, indexFp;
        Persisted() : sourceFp(0), indexFp(0) {}
    };
    static uint64_t fingerprint(const void *p, size_t n, uint64_t h = 1469598103934665603ull);      // FNV-1a
    uint64_t indexFingerprint(int kmerSize, bool spacedKmer, int indexKmerThr, bool maskOnDevice, double maskProb, bool similarKmerTables,
                              const int32_t *more, size_t nMore) const;
    // brings the context to the state buildIndex() leaves, from the file: no lookup, no masking, no index build.  false = no such
    // file / made from something else (error() says which): the caller builds as ever
    bool loadPersisted(const Persisted &file, size_t nTargets, int kmerSize, ScoreMatrix &threeMer, ScoreMatrix &twoMer, bool spacedKmer);
    // ... and the context already holds it (loaded by an earlier MMGpuPrefilter over the same context)
    void adoptResident(size_t nTargets) { dbSize = nTargets; }

    bool buildIndex(SequenceLookup *sequenceLookup, int kmerSize, int indexKmerThr, ScoreMatrix &threeMer, ScoreMatrix &twoMer,
                    bool spacedKmer, bool maskOnDevice = false, double maskProb = 0.9, bool logMasked = true, const Persisted *saveAs = NULL);

    // takeOnlyBestKmer (--exact-kmer-matching; every nucleotide search) / nucleotide target database (matchQuery's isNucleotide)
    void setMode(bool exactKmer, bool nucleotide, bool kmerScoring = false) {
        exactKmerMatching = exactKmer;
        nucleotideSearch = nucleotide;
        kmerScore = kmerScoring;      // --diag-score 0
    }

    // the CacheFriendlyOperations<N> QueryMatcher::initDiagonalMatcher picks on this host (QueryMatcher.cpp:460-488)
    static unsigned int referenceBins(size_t dbSize);

    const std::string &error() const { return err; }

    // queries handed back to the host's matcher so far, by status (MMGPU_PF_OVERFLOW ... MMGPU_PF_SHARD_INEXACT)
    size_t handedBack[8];
    // sharded runs: queries whose merged list came back flagged and were run once more against the whole database ON A DEVICE
    // (mmgpu_multi_pf_redone) - none of them reaches the host's matcher for the way the database was dealt
    size_t rerunUnsplit;

private:
    mmgpu_ctx *gpu;
    mmgpu_multi *multi;
    BaseMatrix *kmerSubMat;
    BaseMatrix *ungappedSubMat;
    bool aaBiasCorrection;
    float aaBiasCorrectionScale;
    size_t dbSize;
    std::string err;
    bool exactKmerMatching, nucleotideSearch, kmerScore;
};

#endif
