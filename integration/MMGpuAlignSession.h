// State of one Alignment::run on the device (MMGpuAlignRun.cpp, MMGpuNuclAlignRun.cpp): the resident targets and, bucket by
// bucket of queries, the results the reference's loop takes at getSWResult's call site.
#ifndef MMGPU_ALIGN_SESSION_H
#define MMGPU_ALIGN_SESSION_H

#include <mutex>
#include <unordered_map>
#include <vector>

#include "Matcher.h"

#include "MMGpuMatcher.h"
#include "MMGpuRun.h"

class Alignment;
class EvalueComputation;
struct MMGpuNuclState;      // MMGpuNuclAlignRun.cpp

struct MMGpuAlignSession {
    MMGpuAlignSession(Alignment &al, EvalueComputation &evaluer);
    ~MMGpuAlignSession();

    Alignment &al;
    EvalueComputation &evaluer;
    mmgpu_ctx *gpu;
    bool nucleotide;
    MMGpuStopwatch watch;
    // resident targets: Sequence::numSequence of the entries the lists name, ids = DBReader ids
    std::vector<unsigned char> targetResidues;
    std::vector<uint64_t> targetOffsets;
    // ... or, in a fused search, the lookup the prefilter module left resident (MMGpuFusedSearch::residentTargets): what the
    // amino-acid path reads
    const unsigned char *tData;
    const uint64_t *tOff;
    // ... or nothing on the host at all (tData == NULL: the prefilter module loaded a persisted device layout,
    // MMGpuFusedSearch::keepResidentTargetsOnDevice): the few sequences the host still reads - identity hits, --corr-score-weight,
    // the test binary's host-side block aligner - are mapped when asked for and kept for the bucket
    // (MMGpuAlignRun::targetResidues)
    std::mutex onDemandLock;
    std::unordered_map<size_t, std::vector<unsigned char> > onDemand;
    Sequence *onDemandSeq;
    // the target database numbers its sequences by their keys (key i = id i, what createdb writes): DBReader::getId's binary
    // search per list entry - half of the list parsing at millions of targets - is then the identity
    bool denseTargetKeys;
    size_t targetId(DBReader<unsigned int> *tdbr, unsigned int key) const {
        return denseTargetKeys ? (key < tdbr->getSize() ? (size_t)key : SIZE_MAX) : tdbr->getId(key);
    }
    // amino-acid / profile queries
    MMGpuAlignBackend *backend;
    MMGpuMatcher *matcher;
    MMGpuBlockBacktracer *blockHook;
    // the bucket in flight: queries [start, start + size) of the prefilter database
    size_t start, size;
    std::vector<MMGpuMatcher::Query> block;
    std::vector<std::vector<unsigned char> > queryNum;
    std::vector<std::vector<int8_t> > queryProfile;       // profile queries: Sequence::getAlignmentProfile()
    std::vector<std::vector<Matcher::result_t> > results;     // [query][k-th entry that reaches getSWResult]
    std::vector<std::vector<unsigned char> > hostPair;        // same shape: 1 = the loop's own Matcher computes it (take)
    // nucleotide databases
    MMGpuNuclState *nucl;
};

#endif
