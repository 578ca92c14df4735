// The hook behind Alignment::run (src/alignment/Alignment.cpp:248-542): Matcher::getSWResult of every list entry on libmmgpu.
//
// The reference's loop stays the reference's: it parses the lists, stops at --max-accept / --max-rejected, applies
// checkCriteria, --alt-ali, --realign, lcaalign, sorts and writes.  The patch (integration/mmseqs_mmgpu.patch) adds four
// things to it: MMGpuAlignRun::begin before the bucket loop (targets resident; NULL = configuration not covered), a cap on
// the bucket size, MMGpuAlignRun::plan before a bucket's OpenMP region - ONE device call aligns all pairs of the bucket's
// lists (MMGpuMatcher::alignBlock = the batch form of initQuery + getSWResult, MMGpuMatcher.cpp) - and MMGpuAlignRun::take
// at the call site of getSWResult, which hands the loop the result the device computed for that entry.  Pairs the device
// declines (traceback storage above its budget, queries outside the regime the kernels restate) are computed by the loop's
// own Matcher inside take().  Second alignments of accepted hits (--realign, lcaalign, --alt-ali) are the loop's own as well.
// The output DB is the same DB, entry for entry (tests/test_mmseqs_dropin.py diffs it against the stock binary).
//
// Compiled into MMseqs2 by integration/build_mmseqs.sh (HAVE_MMGPU); Alignment.h declares the class a friend.
#include <cstring>
#include <string>
#include <vector>

#include "Alignment.h"
#include "Debug.h"
#include "EvalueComputation.h"
#include "StripedSmithWaterman.h"
#include "Util.h"

#include "MMGpuAlignSession.h"
#include "MMGpuFusedSearch.h"
#include "MMGpuHostBlock.h"

#ifdef OPENMP
#include <omp.h>
#endif

MMGpuAlignBackend *mmgpuNewDeviceBackend(mmgpu_ctx *gpu);
MMGpuAlignBackend *mmgpuNewMultiDeviceBackend(const std::vector<mmgpu_ctx *> &gpus);

namespace {

const unsigned char *lookupTarget(void *ctx, unsigned int id) {
    MMGpuAlignSession *s = static_cast<MMGpuAlignSession *>(ctx);
    return MMGpuAlignRun::targetResidues(s, id);
}

}  // namespace

const unsigned char *MMGpuAlignRun::targetResidues(MMGpuAlignSession *s, size_t id) {
    if (s->tData != NULL) return s->tData + s->tOff[id];
    Alignment &al = s->al;
    std::lock_guard<std::mutex> guard(s->onDemandLock);
    std::unordered_map<size_t, std::vector<unsigned char> >::iterator it = s->onDemand.find(id);
    if (it != s->onDemand.end()) return it->second.data();
    if (s->onDemandSeq == NULL) s->onDemandSeq = new Sequence(al.maxSeqLen, al.targetSeqType, al.m, 0, false, al.compBiasCorrection);
    char *data = al.tdbr->getDataUncompressed(id);      // (MMGpuFusedSearch::residentTargets: never a compressed database)
    std::vector<unsigned char> &v = s->onDemand[id];
    if (data != NULL) {
        s->onDemandSeq->mapSequence(id, al.tdbr->getDbKey(id), data, al.tdbr->getSeqLen(id));
        v.assign(s->onDemandSeq->numSequence, s->onDemandSeq->numSequence + s->onDemandSeq->L);
    }
    v.push_back(0);
    return v.data();
}

MMGpuAlignSession::MMGpuAlignSession(Alignment &al, EvalueComputation &evaluer)
    : al(al), evaluer(evaluer), gpu(NULL), nucleotide(false), watch("align"), backend(NULL), matcher(NULL), blockHook(NULL), start(0), size(0),
      tData(NULL), tOff(NULL), onDemandSeq(NULL), nucl(NULL) {}

MMGpuAlignSession::~MMGpuAlignSession() {
    delete matcher;
    delete backend;
    delete static_cast<HostBlockBacktracer *>(blockHook);
    delete onDemandSeq;
}

bool MMGpuAlignRun::usable(const Alignment &a) {
    if (!MMGpuRun::enabled()) return false;
    if (usableNucleotide(a)) return true;
    const bool profileQuery = Parameters::isEqualDbtype(a.querySeqType, Parameters::DBTYPE_HMM_PROFILE);
    const bool aa = (Parameters::isEqualDbtype(a.querySeqType, Parameters::DBTYPE_AMINO_ACIDS) ||
                     (profileQuery && !a.includeIdentity && !a.sameQTDB)) &&
                    Parameters::isEqualDbtype(a.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS);
    // what the device path does not cover keeps the reference's CPU loop: profile targets, nucleotide databases outside
    // MMGpuNuclAlignRun.cpp's conditions, wrapped scoring, the correlation score with profile queries
    if (!aa || a.wrappedScoring || (a.correlationScoreWeight != 0.0f && profileQuery)) {
        Debug(Debug::INFO) << "MMGPU: alignment configuration not covered by the device path, using the CPU path\n";
        return false;
    }
    return true;
}

size_t MMGpuAlignRun::bucketQueries(MMGpuAlignSession *) {
    // overlapped fused search: a bucket starts as soon as the prefilter module has written its entries - smaller buckets start earlier
    if (MMGpuFusedSearch::overlappedRun()) return MMGpuRun::envSize("MMGPU_FUSED_BUCKET_QUERIES", 2048);
    return MMGpuRun::envSize("MMGPU_ALIGN_BLOCK_QUERIES", 16384);
}

void MMGpuAlignRun::end(MMGpuAlignSession *s) {
    if (s == NULL) return;
    endNucleotide(s);
    delete s;
}

// Targets resident: Sequence::numSequence of the entries the prefilter lists of this run name (ids = DBReader ids, :361,367;
// the others keep their id with length 0 - a few queries against a large database do not pay for the whole database).
// A target set the device cannot hold leaves the run to the CPU loop.
MMGpuAlignSession *MMGpuAlignRun::begin(Alignment &al, EvalueComputation &evaluer, size_t dbFrom, size_t dbSize) {
    if (!usable(al)) return NULL;
    MMGpuAlignSession *s = new MMGpuAlignSession(al, evaluer);
    s->gpu = MMGpuRun::context();     // EXITs with the library's message if no device can be opened
    s->watch.lap("open device");
    s->nucleotide = usableNucleotide(al);
    const unsigned int threads = al.threads;
    const size_t nTargets = al.tdbr->getSize();
    {
        bool dense = true;
#pragma omp parallel for schedule(static) reduction(&& : dense) num_threads(threads)
        for (size_t i = 0; i < nTargets; i++) dense = dense && al.tdbr->getDbKey(i) == i;
        s->denseTargetKeys = dense;
    }
    // fused search with the masking on the device: the prefilter module of this process left the unmasked targets resident, host
    // copy included (the same Sequence::numSequence, the same ids: checked key by key)
    bool residentAlready = false;
    std::vector<mmgpu_ctx *> devices;
    if (!s->nucleotide && MMGpuRun::deviceIds().empty() && MMGpuFusedSearch::residentTargets(al.tdbr, s->gpu, &s->tData, &s->tOff)) {
        residentAlready = true;
        devices.push_back(s->gpu);
        s->watch.lap("targets already resident (fused search)");
    }
    if (!residentAlready) {
        std::vector<unsigned char> named(nTargets, 1);
        MMGpuFusedSearch::waitForEntries(dbFrom, dbSize);      // (overlapped fused search without resident targets: all lists first)
        if (Util::getTotalSystemMemory() > al.prefdbr->getTotalDataSize()) {
            std::fill(named.begin(), named.end(), 0);
#pragma omp parallel num_threads(threads)
            {
                unsigned int thread_idx = 0;
#ifdef OPENMP
                thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
                char key[255 + 1];
#pragma omp for schedule(dynamic, 64)
                for (size_t id = dbFrom; id < dbFrom + dbSize; id++) {
                    char *data = al.prefdbr->getData(id, thread_idx);
                    while (*data != '\0') {
                        Util::parseKey(data, key);
                        const size_t dbId = s->targetId(al.tdbr, Util::fast_atoi<DBKeyType>(key));
                        if (dbId < nTargets) named[dbId] = 1;
                        data = Util::skipLine(data);
                    }
                }
            }
        }
        s->targetOffsets.assign(nTargets + 1, 0);
        for (size_t id = 0; id < nTargets; id++) s->targetOffsets[id + 1] = s->targetOffsets[id] + (named[id] ? al.tdbr->getSeqLen(id) : 0);
        s->targetResidues.resize(s->targetOffsets[nTargets] + 1);
#pragma omp parallel num_threads(threads)
        {
            unsigned int thread_idx = 0;
#ifdef OPENMP
            thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
            Sequence dbSeq(al.maxSeqLen, al.targetSeqType, al.m, 0, false, al.compBiasCorrection);
#pragma omp for schedule(dynamic, 256)
            for (size_t id = 0; id < nTargets; id++) {
                if (!named[id]) continue;
                char *data = al.tdbr->getData(id, thread_idx);
                if (data == NULL) continue;
                dbSeq.mapSequence(id, al.tdbr->getDbKey(id), data, al.tdbr->getSeqLen(id));
                memcpy(s->targetResidues.data() + s->targetOffsets[id], dbSeq.numSequence, dbSeq.L);
            }
        }
        s->watch.lap("map targets");
        // MMGPU_DEVICES: every device holds the targets, the queries of a bucket are dealt to them (MMGpuMultiDeviceBackend)
        if (!s->nucleotide) devices = MMGpuRun::allContexts();
        if (devices.empty()) devices.push_back(s->gpu);
        for (size_t d = 0; d < devices.size(); d++)
            if (mmgpu_load_targets(devices[d], s->targetResidues.data(), s->targetOffsets.data(), (uint32_t)nTargets, al.m->alphabetSize) != 0) {
                Debug(Debug::WARNING) << "MMGPU: the targets of this run cannot be made resident (" << mmgpu_last_error() << "), using the CPU path\n";
                delete s;
                return NULL;
            }
        s->tData = s->targetResidues.data();
        s->tOff = s->targetOffsets.data();
    }
    s->watch.lap("mmgpu_load_targets");
    if (s->nucleotide) {
        beginNucleotide(s);
        return s;
    }
    s->backend = devices.size() > 1 ? mmgpuNewMultiDeviceBackend(devices) : mmgpuNewDeviceBackend(s->gpu);
    s->matcher = new MMGpuMatcher(s->backend, al.m, &evaluer, al.compBiasCorrection, al.compBiasCorrectionScale, al.gapOpen, al.gapExtend);
    const size_t maxMatcherSeqLen = std::max(al.tdbr->getMaxSeqLen(), al.qdbr->getMaxSeqLen());
    HostBlockBacktracer *hook = new HostBlockBacktracer(threads, maxMatcherSeqLen, al.m, al.compBiasCorrection, al.compBiasCorrectionScale,
                                                        al.gapOpen, al.gapExtend, al.querySeqType);
    s->blockHook = hook;
    s->matcher->setThreads(threads);
    s->matcher->setCorrelationScoreWeight(al.correlationScoreWeight);
    if (MMGpuRun::hostBlockAligner()) s->matcher->setBlockBacktracer(hook, lookupTarget, s);
    s->matcher->setDeviceBlockAligner(MMGpuRun::deviceBlockAligner());
    // result_t::backtrace is read by resultToBuffer with -a (Matcher.cpp:317-323), by --realign (:397) and --alt-ali only
    s->matcher->setNeedBacktraceStrings(al.addBacktrace || al.realign || al.altAlignment > 0 || al.lcaAlign);
    return s;
}

// The list walk of :316-375 without the alignment, for the queries [start, start + bucketSize) of the prefilter database, and
// one device call for all of their pairs.
void MMGpuAlignRun::plan(MMGpuAlignSession *s, size_t start, size_t bucketSize) {
    s->start = start;
    s->size = bucketSize;
    MMGpuFusedSearch::waitForEntries(start, bucketSize);      // (overlapped fused search: the prefilter module may still be at work)
    if (s->nucleotide) {
        planNucleotide(s);
        return;
    }
    Alignment &al = s->al;
    const size_t nq = bucketSize;
    const bool profileQuery = Parameters::isEqualDbtype(al.querySeqType, Parameters::DBTYPE_HMM_PROFILE);
    s->onDemand.clear();
    s->block.assign(nq, MMGpuMatcher::Query());
    s->queryNum.assign(nq, std::vector<unsigned char>());
    s->queryProfile.assign(nq, std::vector<int8_t>());
    s->hostPair.assign(nq, std::vector<unsigned char>());
#pragma omp parallel num_threads(al.threads)
    {
        unsigned int thread_idx = 0;
#ifdef OPENMP
        thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
        Sequence qSeq(al.maxSeqLen, al.querySeqType, al.m, 0, false, al.compBiasCorrection);
        char buffer[1024];
#pragma omp for schedule(dynamic, 5)
        for (size_t b = 0; b < nq; b++) {
            const size_t id = start + b;
            char *data = al.prefdbr->getData(id, thread_idx);
            const DBKeyType queryDbKey = al.prefdbr->getDbKey(id);
            MMGpuMatcher::Query &q = s->block[b];
            size_t origQueryLen = 0;
            if (*data != '\0') {
                const size_t qId = al.qdbr->getId(queryDbKey);
                char *querySeqData = al.qdbr->getData(qId, thread_idx);
                if (querySeqData == NULL) continue;      // (the loop reports it)
                origQueryLen = al.qdbr->getSeqLen(qId);
                qSeq.mapSequence(qId, queryDbKey, querySeqData, origQueryLen);
                s->queryNum[b].assign(qSeq.numSequence, qSeq.numSequence + qSeq.L);
                q.numSequence = s->queryNum[b].data();
                q.L = qSeq.L;
                if (profileQuery) {     // Matcher::initQuery: the aligner gets the profile's own score rows (Matcher.cpp:49-60)
                    const int8_t *ap = qSeq.getAlignmentProfile();
                    s->queryProfile[b].assign(ap, ap + Sequence::PROFILE_AA_SIZE * (size_t)qSeq.L);
                    q.profile = s->queryProfile[b].data();
                }
            }
            // the entry's target keys: parsed out of its lines (:345-347), or - fused search - as the prefilter hook of this process
            // left them beside the text (MMGpuFusedSearch::capturedKeys; the reference's loop still walks the text for its take() calls)
            const unsigned int *keptKeys = NULL;
            size_t nKept = 0, nextKept = 0;
            const bool kept = MMGpuFusedSearch::capturedKeys(al.prefdbr, id, &keptKeys, &nKept);
            if (kept) q.targets.reserve(nKept);
            while (kept ? nextKept < nKept : *data != '\0') {
                DBKeyType dbKey;
                if (kept) {
                    dbKey = keptKeys[nextKept++];
                } else {
                    Util::parseKey(data, buffer);
                    dbKey = Util::fast_atoi<DBKeyType>(buffer);
                    data = Util::skipLine(data);
                }
                const size_t dbId = s->targetId(al.tdbr, dbKey);
                if (dbId >= al.tdbr->getSize() || al.tdbr->getData(dbId, thread_idx) == NULL) break;      // (the loop reports it and ends the run)
                const int dbLen = (int)(s->tOff[dbId + 1] - s->tOff[dbId]);
                if (!Util::canBeCovered(al.canCovThr, al.covMode, static_cast<float>(origQueryLen), static_cast<float>(dbLen))) continue;
                MMGpuMatcher::Target t;
                t.id = (unsigned int)dbId;
                t.dbKey = dbKey;
                t.length = dbLen;
                t.isIdentity = (queryDbKey == dbKey && (al.includeIdentity || al.sameQTDB)) ? true : false;
                t.numSequence = s->tData != NULL ? s->tData + s->tOff[dbId] : ((t.isIdentity || al.correlationScoreWeight > 0.0f) ? targetResidues(s, dbId) : NULL);
                q.targets.push_back(t);
            }
            s->hostPair[b].assign(q.targets.size(), 0);
        }
    }
    s->watch.lap("parse bucket");
    static_cast<HostBlockBacktracer *>(s->blockHook)->newBlock();
    std::vector<std::pair<size_t, size_t> > refused;
    if (!s->matcher->alignBlock(s->block, al.covMode, al.covThr, al.evalThr, al.swMode, al.seqIdMode, s->results, &refused)) {
        Debug(Debug::ERROR) << "MMGPU: " << s->matcher->error() << "\n";
        EXIT(EXIT_FAILURE);
    }
    for (size_t r = 0; r < refused.size(); r++) s->hostPair[refused[r].first][refused[r].second] = 1;
    s->watch.lap("alignBlock");
}

// entry k (counting the entries that reach getSWResult, :379) of the list of query `id`
Matcher::result_t MMGpuAlignRun::take(MMGpuAlignSession *s, size_t id, size_t k, Matcher &matcher, Sequence *dbSeq, int diagonal, bool isReverse,
                                      bool isIdentity) {
    const size_t b = id - s->start;
    if (b >= s->results.size() || k >= s->results[b].size()) {
        Debug(Debug::ERROR) << "MMGPU: no planned result for entry " << k << " of query " << id << "\n";
        EXIT(EXIT_FAILURE);
    }
    if (!s->nucleotide && s->hostPair[b][k]) {
        const Alignment &al = s->al;
        return matcher.getSWResult(dbSeq, diagonal, isReverse, al.covMode, al.covThr, al.evalThr, al.swMode, al.seqIdMode, isIdentity, al.wrappedScoring);
    }
    return s->results[b][k];
}
