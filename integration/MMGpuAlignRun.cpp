// Alignment::run (src/alignment/Alignment.cpp:248-542) with the Smith-Waterman work on libmmgpu.
//
// The reference's loop handles one query per OpenMP iteration: parse the prefilter list, matcher.initQuery (:340),
// matcher.getSWResult per list entry (:379), checkCriteria, sort, resultToBuffer, DBWriter.  Here the same function runs
// over blocks of queries: every thread parses lists and maps sequences, ONE device call aligns all pairs of the block
// (MMGpuMatcher::alignBlock = the batch form of initQuery + getSWResult, MMGpuMatcher.cpp), then every thread replays
// the reference's accept / reject bookkeeping (:344-397) on the results in list order, sorts with
// Matcher::compareHits and serialises with Matcher::resultToBuffer - the reference's own functions, unchanged.
// The output DB is the same DB, entry for entry (tests/test_mmseqs_dropin.py diffs it against the stock binary).
//
// Compiled into MMseqs2 by integration/build_mmseqs.sh (HAVE_MMGPU); Alignment.h declares the class a friend.
#include <cfloat>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Alignment.h"
#include "DBWriter.h"
#include "Debug.h"
#include "EvalueComputation.h"
#include "QueryMatcher.h"
#include "StripedSmithWaterman.h"
#include "Util.h"

#include "MMGpuMatcher.h"
#include "MMGpuRun.h"

#ifdef OPENMP
#include <omp.h>
#endif

MMGpuAlignBackend *mmgpuNewDeviceBackend(mmgpu_ctx *gpu);
MMGpuAlignBackend *mmgpuNewMultiDeviceBackend(const std::vector<mmgpu_ctx *> &gpus);

namespace {

// MMGpuBlockBacktracer over the host's own SmithWaterman objects: ssw_init once per (thread, query), then the
// reference's alignStartPosBacktraceBlock through the public wrapper the patch adds to SmithWaterman.
class HostBlockBacktracer : public MMGpuBlockBacktracer {
public:
    HostBlockBacktracer(unsigned int threads, size_t maxSeqLen, BaseMatrix *m, bool compBias, float compBiasScale, int gapOpen,
                        int gapExtend, int seqType)
        : m(m), gapOpen(gapOpen), gapExtend(gapExtend), sw(threads, NULL), seq(threads, NULL), lastQuery(threads, (size_t)-1),
          maxSeqLen(maxSeqLen), compBias(compBias), compBiasScale(compBiasScale), seqType(seqType) {
        const int a = m->alphabetSize;
        tiny.resize(a * a);
        for (int i = 0; i < a; i++)
            for (int j = 0; j < a; j++) tiny[i * a + j] = (int8_t)m->subMatrix[i][j];
    }
    ~HostBlockBacktracer() {
        for (size_t i = 0; i < sw.size(); i++) {
            delete sw[i];
            delete seq[i];
        }
    }
    void newBlock() { std::fill(lastQuery.begin(), lastQuery.end(), (size_t)-1); }
    bool run(unsigned int thread, size_t queryIndex, const unsigned char *query, int queryLength, const unsigned char *target,
             int targetLength, s_align &a, std::string &backtrace) {
        if (thread >= sw.size()) {
            Debug(Debug::ERROR) << "MMGPU: OpenMP thread " << thread << " outside the " << sw.size() << " threads of this run\n";
            EXIT(EXIT_FAILURE);
        }
        if (sw[thread] == NULL) {
            sw[thread] = new SmithWaterman(maxSeqLen, m->alphabetSize, compBias, compBiasScale, (SubstitutionMatrix *)m);
            seq[thread] = new Sequence(maxSeqLen, seqType, m, 0, false, compBias);
        }
        if (lastQuery[thread] != queryIndex) {
            seq[thread]->mapSequence(0, 0, std::make_pair(query, (const unsigned int)queryLength));
            sw[thread]->ssw_init(seq[thread], tiny.data(), m);
            lastQuery[thread] = queryIndex;
        }
        s_align r = sw[thread]->mmgpuBlockBacktrace(target, targetLength, (uint8_t)gapOpen, (uint8_t)gapExtend, backtrace, a);
        if (r.score1 == UINT32_MAX) {
            backtrace.clear();
            return false;
        }
        a = r;
        return true;
    }

private:
    BaseMatrix *m;
    int gapOpen, gapExtend;
    std::vector<SmithWaterman *> sw;
    std::vector<Sequence *> seq;
    std::vector<size_t> lastQuery;
    std::vector<int8_t> tiny;
    size_t maxSeqLen;
    bool compBias;
    float compBiasScale;
    int seqType;
};

struct ListEntry {
    DBKeyType dbKey;
    int target;     // index into MMGpuMatcher::Query::targets, -1: Util::canBeCovered said no (:370-373)
};

struct TargetStore {
    std::vector<unsigned char> residues;
    std::vector<uint64_t> offsets;
};
const unsigned char *lookupTarget(void *ctx, unsigned int id) {
    TargetStore *s = static_cast<TargetStore *>(ctx);
    return s->residues.data() + s->offsets[id];
}

}  // namespace

bool MMGpuAlignRun::usable(const Alignment &a) {
    if (!MMGpuRun::enabled()) return false;
    if (usableNucleotide(a)) return true;
    const bool profileQuery = Parameters::isEqualDbtype(a.querySeqType, Parameters::DBTYPE_HMM_PROFILE);
    const bool aa = (Parameters::isEqualDbtype(a.querySeqType, Parameters::DBTYPE_AMINO_ACIDS) ||
                     (profileQuery && !a.includeIdentity && !a.sameQTDB)) &&
                    Parameters::isEqualDbtype(a.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS);
    // what the device path does not cover keeps the reference's CPU loop: profile targets, nucleotide databases outside
    // MMGpuNuclAlignRun.cpp's conditions,
    // wrapped scoring, realignment (incl. the LCA form) of profile queries, the correlation score with profile queries / realignment
    // (--realign with sequence queries - the first iteration of an iterative search - is served: run() below)
    // (--alt-ali is served: the list on the device, the few re-alignments of masked targets on the host)
    if (!aa || (a.realign && profileQuery) || a.wrappedScoring || (a.correlationScoreWeight != 0.0f && (profileQuery || a.realign))) {
        Debug(Debug::INFO) << "MMGPU: alignment configuration not covered by the device path, using the CPU path\n";
        return false;
    }
    return true;
}

bool MMGpuAlignRun::run(Alignment &al, const std::string &outDB, const std::string &outDBIndex, const size_t dbFrom,
                        const size_t dbSize, bool merge) {
    if (!usable(al)) return false;
    if (usableNucleotide(al)) return runNucleotide(al, outDB, outDBIndex, dbFrom, dbSize, merge);
    MMGpuStopwatch watch("align");
    mmgpu_ctx *gpu = MMGpuRun::context();     // EXITs with the library's message if no device can be opened
    watch.lap("open device");

    int dbtype = Parameters::DBTYPE_ALIGNMENT_RES;
    if (al.alignmentOutputMode == Parameters::ALIGNMENT_OUTPUT_CLUSTER) {
        dbtype = Parameters::DBTYPE_CLUSTER_RES;
    }
    dbtype = DBReader<DBKeyType>::setExtendedDbtype(dbtype, DBReader<DBKeyType>::getExtendedDbtype(al.prefdbr->getDbtype()));
    DBWriter dbw(outDB.c_str(), outDBIndex.c_str(), al.threads, al.compressed, dbtype);
    if (dbSize == 0) {
        dbw.open();
        dbw.close(merge);
        return true;
    }
    // (the writer's files are created once the targets are resident: a database the device cannot hold leaves the run to the CPU loop)
    EvalueComputation evaluer(al.tdbr->getAminoAcidDBSize(), al.m, al.gapOpen, al.gapExtend);
    const unsigned int threads = al.threads;

    // ---- resident targets: Sequence::numSequence of every entry of the target DB, ids = DBReader ids (:361,367)
    TargetStore store;
    const size_t nTargets = al.tdbr->getSize();
    // (only the targets some prefilter list of this run names are mapped and uploaded - a few queries against a large database
    // do not pay for the whole database; the others keep their id with length 0.  One extra pass over the list text.)
    std::vector<unsigned char> named(nTargets, 1);
    if (Util::getTotalSystemMemory() > al.prefdbr->getTotalDataSize()) {
        std::fill(named.begin(), named.end(), 0);
#pragma omp parallel num_threads(al.threads)
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
                    const size_t dbId = al.tdbr->getId(Util::fast_atoi<DBKeyType>(key));
                    if (dbId < nTargets) named[dbId] = 1;
                    data = Util::skipLine(data);
                }
            }
        }
    }
    store.offsets.assign(nTargets + 1, 0);
    for (size_t id = 0; id < nTargets; id++) store.offsets[id + 1] = store.offsets[id] + (named[id] ? al.tdbr->getSeqLen(id) : 0);
    store.residues.resize(store.offsets[nTargets] + 1);
    watch.lap("target offsets + host buffer");
    std::vector<Sequence *> qSeqs(threads, NULL), dbSeqs(threads, NULL);
#pragma omp parallel num_threads(threads)
    {
        unsigned int thread_idx = 0;
#ifdef OPENMP
        thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
        qSeqs[thread_idx] = new Sequence(al.maxSeqLen, al.querySeqType, al.m, 0, false, al.compBiasCorrection);
        dbSeqs[thread_idx] = new Sequence(al.maxSeqLen, al.targetSeqType, al.m, 0, false, al.compBiasCorrection);
        Sequence &dbSeq = *dbSeqs[thread_idx];
#pragma omp for schedule(dynamic, 256)
        for (size_t id = 0; id < nTargets; id++) {
            if (!named[id]) continue;
            char *data = al.tdbr->getData(id, thread_idx);
            if (data == NULL) continue;
            dbSeq.mapSequence(id, al.tdbr->getDbKey(id), data, al.tdbr->getSeqLen(id));
            memcpy(store.residues.data() + store.offsets[id], dbSeq.numSequence, dbSeq.L);
        }
    }
    watch.lap("map targets");
    // MMGPU_DEVICES: every device holds the targets, the queries of a block are dealt to them (MMGpuMultiDeviceBackend)
    std::vector<mmgpu_ctx *> devices;
    if (mmgpu_multi *multi = MMGpuRun::multi())
        for (int d = 0; d < mmgpu_multi_size(multi); d++) devices.push_back(mmgpu_multi_ctx(multi, d));
    if (devices.empty()) devices.push_back(gpu);
    for (size_t d = 0; d < devices.size(); d++)
        if (mmgpu_load_targets(devices[d], store.residues.data(), store.offsets.data(), (uint32_t)nTargets, al.m->alphabetSize) != 0) {
            Debug(Debug::WARNING) << "MMGPU: the targets of this run cannot be made resident (" << mmgpu_last_error() << "), using the CPU path\n";
            for (size_t i = 0; i < threads; i++) {
                delete qSeqs[i];
                delete dbSeqs[i];
            }
            return false;
        }
    dbw.open();

    watch.lap("mmgpu_load_targets");
    MMGpuAlignBackend *backend = devices.size() > 1 ? mmgpuNewMultiDeviceBackend(devices) : mmgpuNewDeviceBackend(gpu);
    MMGpuMatcher gpuMatcher(backend, al.m, &evaluer, al.compBiasCorrection, al.compBiasCorrectionScale, al.gapOpen, al.gapExtend);
    const size_t maxMatcherSeqLen = std::max(al.tdbr->getMaxSeqLen(), al.qdbr->getMaxSeqLen());
    HostBlockBacktracer blockHook(threads, maxMatcherSeqLen, al.m, al.compBiasCorrection, al.compBiasCorrectionScale, al.gapOpen,
                                  al.gapExtend, al.querySeqType);
    gpuMatcher.setThreads(threads);
    gpuMatcher.setCorrelationScoreWeight(al.correlationScoreWeight);
    if (MMGpuRun::hostBlockAligner()) gpuMatcher.setBlockBacktracer(&blockHook, lookupTarget, &store);
    gpuMatcher.setDeviceBlockAligner(MMGpuRun::deviceBlockAligner());
    std::vector<Matcher *> cpuMatchers(threads, NULL);      // only for pairs whose backtrace the device declines
    // --realign (:298-305,408-437): the accepted hits of a query are aligned a second time with the (biased) realign matrix
    // for their boundaries and backtraces; scores and E-values stay the first pass's
    BaseMatrix *realignMat = al.realign_m != NULL ? al.realign_m : al.m;
    MMGpuMatcher gpuRealigner(backend, realignMat, &evaluer, al.compBiasCorrection, al.compBiasCorrectionScale, al.gapOpen, al.gapExtend);
    HostBlockBacktracer realignBlockHook(threads, maxMatcherSeqLen, realignMat, al.compBiasCorrection, al.compBiasCorrectionScale, al.gapOpen,
                                         al.gapExtend, al.querySeqType);
    gpuRealigner.setThreads(threads);
    if (MMGpuRun::hostBlockAligner()) gpuRealigner.setBlockBacktracer(&realignBlockHook, lookupTarget, &store);
    gpuRealigner.setDeviceBlockAligner(MMGpuRun::deviceBlockAligner());
    std::vector<Matcher *> cpuRealigners(threads, NULL);      // refused pairs of the realignment, --alt-ali after --realign
    std::vector<std::vector<Matcher::result_t> > accepted, realigned;
    std::vector<MMGpuMatcher::Query> block2, block3;
    std::vector<std::vector<Matcher::result_t> > lcaResults;
    std::vector<std::pair<size_t, size_t> > refused2;

    // block = as many queries as keep the pair count of one device call bounded (a prefilter line has >= 6 bytes)
    const size_t maxBlockQueries = MMGpuRun::envSize("MMGPU_ALIGN_BLOCK_QUERIES", 16384);
    const size_t maxBlockBytes = MMGpuRun::envSize("MMGPU_ALIGN_BLOCK_BYTES", 192u << 20);
    const bool remap = Util::getTotalSystemMemory() <= al.prefdbr->getTotalDataSize();

    size_t alignmentsNum = 0;
    size_t totalPassedNum = 0;
    Debug::Progress progress(dbSize);
    std::vector<MMGpuMatcher::Query> block;
    std::vector<std::vector<unsigned char> > queryNum;
    std::vector<std::vector<int8_t> > queryProfile;       // profile queries: Sequence::getAlignmentProfile()
    std::vector<size_t> queryIds;
    const bool profileQuery = Parameters::isEqualDbtype(al.querySeqType, Parameters::DBTYPE_HMM_PROFILE);
    std::vector<std::vector<ListEntry> > lists;
    std::vector<DBKeyType> queryKeys;
    std::vector<std::vector<Matcher::result_t> > results;
    std::vector<std::pair<size_t, size_t> > refused;
    size_t next = dbFrom;
    const size_t end = dbFrom + dbSize;
    while (next < end) {
        size_t blockEnd = next, bytes = 0;
        while (blockEnd < end && blockEnd - next < maxBlockQueries && (bytes < maxBlockBytes || blockEnd == next)) {
            bytes += al.prefdbr->getEntryLen(blockEnd);
            blockEnd++;
        }
        const size_t nq = blockEnd - next;
        block.assign(nq, MMGpuMatcher::Query());
        queryNum.assign(nq, std::vector<unsigned char>());
        queryProfile.assign(nq, std::vector<int8_t>());
        queryIds.assign(nq, 0);
        lists.assign(nq, std::vector<ListEntry>());
        queryKeys.assign(nq, 0);

        // ---- parse: the list walk of :316-375 without the alignment
#pragma omp parallel num_threads(threads)
        {
            unsigned int thread_idx = 0;
#ifdef OPENMP
            thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
            Sequence &qSeq = *qSeqs[thread_idx];
            char buffer[1024 + 32768 * 4];
#pragma omp for schedule(dynamic, 5)
            for (size_t b = 0; b < nq; b++) {
                const size_t id = next + b;
                char *data = al.prefdbr->getData(id, thread_idx);
                const DBKeyType queryDbKey = al.prefdbr->getDbKey(id);
                queryKeys[b] = queryDbKey;
                MMGpuMatcher::Query &q = block[b];
                q.numSequence = NULL;
                q.L = 0;
                size_t origQueryLen = 0;
                if (*data != '\0') {
                    size_t qId = al.qdbr->getId(queryDbKey);
                    char *querySeqData = al.qdbr->getData(qId, thread_idx);
                    if (querySeqData == NULL) {
                        Debug(Debug::ERROR) << "Query sequence " << queryDbKey
                                            << " is required in the prefiltering, but is not contained in the query sequence database.\nPlease check your database.\n";
                        EXIT(EXIT_FAILURE);
                    }
                    origQueryLen = al.qdbr->getSeqLen(qId);
                    qSeq.mapSequence(qId, queryDbKey, querySeqData, origQueryLen);
                    queryNum[b].assign(qSeq.numSequence, qSeq.numSequence + qSeq.L);
                    q.numSequence = queryNum[b].data();
                    q.L = qSeq.L;
                    queryIds[b] = qId;
                    if (profileQuery) {     // Matcher::initQuery: the aligner gets the profile's own score rows (Matcher.cpp:49-60)
                        const int8_t *ap = qSeq.getAlignmentProfile();
                        queryProfile[b].assign(ap, ap + Sequence::PROFILE_AA_SIZE * (size_t)qSeq.L);
                        q.profile = queryProfile[b].data();
                    }
                }
                while (*data != '\0') {
                    Util::parseKey(data, buffer);
                    const DBKeyType dbKey = Util::fast_atoi<DBKeyType>(buffer);
                    data = Util::skipLine(data);
                    const size_t dbId = al.tdbr->getId(dbKey);
                    if (al.tdbr->getData(dbId, thread_idx) == NULL) {
                        Debug(Debug::ERROR) << "Sequence " << dbKey << " is required in the prefiltering, but is not contained in the target sequence database!\nPlease check your database.\n";
                        EXIT(EXIT_FAILURE);
                    }
                    const int dbLen = (int)(store.offsets[dbId + 1] - store.offsets[dbId]);
                    ListEntry e;
                    e.dbKey = dbKey;
                    e.target = -1;
                    if (Util::canBeCovered(al.canCovThr, al.covMode, static_cast<float>(origQueryLen), static_cast<float>(dbLen))) {
                        MMGpuMatcher::Target t;
                        t.id = (unsigned int)dbId;
                        t.dbKey = dbKey;
                        t.length = dbLen;
                        t.numSequence = store.residues.data() + store.offsets[dbId];
                        t.isIdentity = (queryDbKey == dbKey && (al.includeIdentity || al.sameQTDB)) ? true : false;
                        e.target = (int)q.targets.size();
                        q.targets.push_back(t);
                    }
                    lists[b].push_back(e);
                }
            }
        }

        // ---- one device call for the block
        watch.lap("parse block");
        blockHook.newBlock();
        if (!gpuMatcher.alignBlock(block, al.covMode, al.covThr, al.evalThr, al.swMode, al.seqIdMode, results, &refused)) {
            Debug(Debug::ERROR) << "MMGPU: " << gpuMatcher.error() << "\n";
            EXIT(EXIT_FAILURE);
        }
        watch.lap("alignBlock");
        // pairs whose backtrace the device declined (band storage above its budget): the reference's own call
        for (size_t r = 0; r < refused.size(); r++) {
            const size_t b = refused[r].first;
            const MMGpuMatcher::Target &t = block[b].targets[refused[r].second];
            if (cpuMatchers[0] == NULL)
                cpuMatchers[0] = new Matcher(al.querySeqType, maxMatcherSeqLen, al.m, &evaluer, al.compBiasCorrection,
                                             al.compBiasCorrectionScale, al.gapOpen, al.gapExtend, al.correlationScoreWeight, al.zdrop);
            if (profileQuery) qSeqs[0]->mapSequence(queryIds[b], queryKeys[b], al.qdbr->getData(queryIds[b], 0), al.qdbr->getSeqLen(queryIds[b]));
            else qSeqs[0]->mapSequence(0, queryKeys[b], std::make_pair(block[b].numSequence, (const unsigned int)block[b].L));
            dbSeqs[0]->mapSequence(t.id, t.dbKey, std::make_pair(t.numSequence, (const unsigned int)t.length));
            cpuMatchers[0]->initQuery(qSeqs[0]);
            results[b][refused[r].second] = cpuMatchers[0]->getSWResult(dbSeqs[0], 0, false, al.covMode, al.covThr, al.evalThr,
                                                                        al.swMode, al.seqIdMode, false, false);
        }

        // ---- replay of :344-397 on the results, sort
        accepted.assign(nq, std::vector<Matcher::result_t>());
#pragma omp parallel num_threads(threads)
        {
#pragma omp for schedule(dynamic, 5) reduction(+ : alignmentsNum, totalPassedNum)
            for (size_t b = 0; b < nq; b++) {
                std::vector<Matcher::result_t> &swResults = accepted[b];
                size_t passedNum = 0;
                unsigned int rejected = 0;
                for (size_t k = 0; k < lists[b].size() && passedNum < al.maxAccept && rejected < al.maxReject; k++) {
                    const ListEntry &e = lists[b][k];
                    if (e.target < 0) {
                        rejected++;
                        continue;
                    }
                    Matcher::result_t &res = results[b][e.target];
                    const bool isIdentity = block[b].targets[e.target].isIdentity;
                    alignmentsNum++;
                    if (isIdentity) {
                        res.qcov = 1.0f;
                        res.dbcov = 1.0f;
                        res.seqId = 1.0f;
                    }
                    if (Alignment::checkCriteria(res, isIdentity, al.evalThr, al.seqIdThr, al.alnLenThr, al.covMode, al.covThr)) {
                        swResults.emplace_back(res);
                        passedNum++;
                        totalPassedNum++;
                        rejected = 0;
                    } else {
                        rejected++;
                    }
                }
                // --alt-ali (:399-401, computeAlternativeAlignment :569-601): the aligned range of an accepted target is masked
                // with X and the pair aligned again, up to altAlignment times.  The masked targets are not the resident ones
                // and there are few of them: the reference's own function with a host Matcher per thread.
                if (al.altAlignment > 0 && al.realign == false && al.wrappedScoring == false && !swResults.empty()) {
                    unsigned int thread_idx = 0;
#ifdef OPENMP
                    thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
                    if (cpuMatchers[thread_idx] == NULL)
                        cpuMatchers[thread_idx] = new Matcher(al.querySeqType, maxMatcherSeqLen, al.m, &evaluer, al.compBiasCorrection,
                                                              al.compBiasCorrectionScale, al.gapOpen, al.gapExtend, al.correlationScoreWeight, al.zdrop);
                    Sequence &qSeq = *qSeqs[thread_idx];
                    if (profileQuery) qSeq.mapSequence(queryIds[b], queryKeys[b], al.qdbr->getData(queryIds[b], thread_idx), al.qdbr->getSeqLen(queryIds[b]));
                    else qSeq.mapSequence(0, queryKeys[b], std::make_pair(block[b].numSequence, (const unsigned int)block[b].L));
                    cpuMatchers[thread_idx]->initQuery(&qSeq);
                    al.computeAlternativeAlignment(queryKeys[b], *dbSeqs[thread_idx], swResults, *cpuMatchers[thread_idx], al.covThr, al.evalThr,
                                                   al.swMode, thread_idx);
                }
                if (swResults.size() > 1) {
                    SORT_SERIAL(swResults.begin(), swResults.end(), Matcher::compareHits);
                }
            }
        }
        // ---- --realign: second device call over the accepted hits (:408-437)
        if (al.realign) {
            watch.lap("accept / sort");
            block2.assign(nq, MMGpuMatcher::Query());
            for (size_t b = 0; b < nq; b++) {
                MMGpuMatcher::Query &q = block2[b];
                q.numSequence = block[b].numSequence;
                q.L = block[b].L;
                q.profile = NULL;
                if (lists[b].empty()) continue;       // *origData == '\0': the first pass's (empty) result is written
                // (with a small --realign-max-seqs - lcaalign sets 1 - only the first hits go to the device; the loop below
                // takes later ones, if the first do not pass, from the host's Matcher one by one like the reference)
                const size_t onDevice = std::min(accepted[b].size(), (size_t)std::min<long long>((long long)al.realignMaxSeqs * 4ll, 1ll << 30));
                for (size_t r = 0; r < onDevice; r++) {
                    const DBKeyType dbKey = accepted[b][r].dbKey;
                    const size_t dbId = al.tdbr->getId(dbKey);
                    MMGpuMatcher::Target t;
                    t.id = (unsigned int)dbId;
                    t.dbKey = dbKey;
                    t.length = (int)(store.offsets[dbId + 1] - store.offsets[dbId]);
                    t.numSequence = store.residues.data() + store.offsets[dbId];
                    t.isIdentity = (queryKeys[b] == dbKey && (al.includeIdentity || al.sameQTDB)) ? true : false;
                    q.targets.push_back(t);
                }
            }
            realignBlockHook.newBlock();
            if (!gpuRealigner.alignBlock(block2, al.covMode, al.realignCov, FLT_MAX, al.realignSwMode, al.seqIdMode, realigned, &refused2)) {
                Debug(Debug::ERROR) << "MMGPU: " << gpuRealigner.error() << "\n";
                EXIT(EXIT_FAILURE);
            }
            for (size_t r = 0; r < refused2.size(); r++) {
                const size_t b = refused2[r].first;
                const MMGpuMatcher::Target &t = block2[b].targets[refused2[r].second];
                if (cpuRealigners[0] == NULL)
                    cpuRealigners[0] = new Matcher(al.querySeqType, maxMatcherSeqLen, realignMat, &evaluer, al.compBiasCorrection,
                                                   al.compBiasCorrectionScale, al.gapOpen, al.gapExtend, 0.0f, al.zdrop);
                qSeqs[0]->mapSequence(0, queryKeys[b], std::make_pair(block2[b].numSequence, (const unsigned int)block2[b].L));
                dbSeqs[0]->mapSequence(t.id, t.dbKey, std::make_pair(t.numSequence, (const unsigned int)t.length));
                cpuRealigners[0]->initQuery(qSeqs[0]);
                realigned[b][refused2[r].second] = cpuRealigners[0]->getSWResult(dbSeqs[0], INT_MAX, false, al.covMode, al.realignCov, FLT_MAX,
                                                                                 al.realignSwMode, al.seqIdMode, t.isIdentity);
            }
            watch.lap("realign block");
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
            for (size_t b = 0; b < nq; b++) {
                if (lists[b].empty()) continue;
                std::vector<Matcher::result_t> out;
                int realignAccepted = 0;
                for (size_t r = 0; r < accepted[b].size() && realignAccepted < al.realignMaxSeqs; r++) {
                    Matcher::result_t res;
                    bool isIdentity;
                    if (r < block2[b].targets.size()) {
                        res = realigned[b][r];
                        isIdentity = block2[b].targets[r].isIdentity;
                    } else {      // beyond the hits that went to the device: the reference's own call
                        unsigned int thread_idx = 0;
#ifdef OPENMP
                        thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
                        if (cpuRealigners[thread_idx] == NULL)
                            cpuRealigners[thread_idx] = new Matcher(al.querySeqType, maxMatcherSeqLen, realignMat, &evaluer, al.compBiasCorrection,
                                                                    al.compBiasCorrectionScale, al.gapOpen, al.gapExtend, 0.0f, al.zdrop);
                        const DBKeyType dbKey = accepted[b][r].dbKey;
                        const size_t dbId = al.tdbr->getId(dbKey);
                        isIdentity = (queryKeys[b] == dbKey && (al.includeIdentity || al.sameQTDB)) ? true : false;
                        qSeqs[thread_idx]->mapSequence(0, queryKeys[b], std::make_pair(block2[b].numSequence, (const unsigned int)block2[b].L));
                        dbSeqs[thread_idx]->mapSequence(dbId, dbKey, std::make_pair(store.residues.data() + store.offsets[dbId],
                                                                                   (const unsigned int)(store.offsets[dbId + 1] - store.offsets[dbId])));
                        cpuRealigners[thread_idx]->initQuery(qSeqs[thread_idx]);
                        res = cpuRealigners[thread_idx]->getSWResult(dbSeqs[thread_idx], INT_MAX, false, al.covMode, al.realignCov, FLT_MAX,
                                                                     al.realignSwMode, al.seqIdMode, isIdentity);
                    }
                    const bool covOK = Util::hasCoverage(al.realignCov, al.covMode, res.qcov, res.dbcov);
                    if (covOK == true || isIdentity) {
                        res.score = accepted[b][r].score;
                        res.eval = accepted[b][r].eval;
                        out.emplace_back(res);
                        realignAccepted++;
                    }
                }
                if (al.altAlignment > 0 && !out.empty()) {      // :433-435, with the realigner and the realignment's thresholds
                    unsigned int thread_idx = 0;
#ifdef OPENMP
                    thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
                    if (cpuRealigners[thread_idx] == NULL)
                        cpuRealigners[thread_idx] = new Matcher(al.querySeqType, maxMatcherSeqLen, realignMat, &evaluer, al.compBiasCorrection,
                                                                al.compBiasCorrectionScale, al.gapOpen, al.gapExtend, 0.0f, al.zdrop);
                    Sequence &qSeq = *qSeqs[thread_idx];
                    qSeq.mapSequence(0, queryKeys[b], std::make_pair(block2[b].numSequence, (const unsigned int)block2[b].L));
                    cpuRealigners[thread_idx]->initQuery(&qSeq);
                    al.computeAlternativeAlignment(queryKeys[b], *dbSeqs[thread_idx], out, *cpuRealigners[thread_idx], al.realignCov, FLT_MAX,
                                                   al.realignSwMode, thread_idx);
                }
                if (out.size() > 1) {
                    SORT_SERIAL(out.begin(), out.end(), Matcher::compareHits);
                }
                accepted[b].swap(out);
            }
        }
        // ---- lcaalign (:444-498): the aligned stretch of the top hit's TARGET becomes the query, every entry of the prefilter
        // list is aligned against it under the top hit's E-value; what passes is the result
        if (al.lcaAlign) {
            block3.assign(nq, MMGpuMatcher::Query());
            for (size_t b = 0; b < nq; b++) {
                MMGpuMatcher::Query &q = block3[b];
                if (accepted[b].empty()) continue;
                const Matcher::result_t &top = accepted[b][0];
                const size_t topId = al.tdbr->getId(top.dbKey);
                q.numSequence = store.residues.data() + store.offsets[topId] + top.dbStartPos;
                q.L = top.dbEndPos - top.dbStartPos + 1;
                q.profile = NULL;
                q.evalThr = top.eval;
                for (size_t k = 0; k < lists[b].size(); k++) {
                    const size_t dbId = al.tdbr->getId(lists[b][k].dbKey);
                    MMGpuMatcher::Target t;
                    t.id = (unsigned int)dbId;
                    t.dbKey = lists[b][k].dbKey;
                    t.length = (int)(store.offsets[dbId + 1] - store.offsets[dbId]);
                    t.numSequence = store.residues.data() + store.offsets[dbId];
                    t.isIdentity = false;
                    q.targets.push_back(t);
                }
            }
            realignBlockHook.newBlock();
            if (!gpuRealigner.alignBlock(block3, al.covMode, al.realignCov, 0.0, al.lcaSwMode, al.seqIdMode, lcaResults, &refused2)) {
                Debug(Debug::ERROR) << "MMGPU: " << gpuRealigner.error() << "\n";
                EXIT(EXIT_FAILURE);
            }
            for (size_t r = 0; r < refused2.size(); r++) {
                const size_t b = refused2[r].first;
                const MMGpuMatcher::Target &t = block3[b].targets[refused2[r].second];
                if (cpuRealigners[0] == NULL)
                    cpuRealigners[0] = new Matcher(al.querySeqType, maxMatcherSeqLen, realignMat, &evaluer, al.compBiasCorrection,
                                                   al.compBiasCorrectionScale, al.gapOpen, al.gapExtend, 0.0f, al.zdrop);
                qSeqs[0]->mapSequence(0, queryKeys[b], std::make_pair(block3[b].numSequence, (const unsigned int)block3[b].L));
                dbSeqs[0]->mapSequence(t.id, t.dbKey, std::make_pair(t.numSequence, (const unsigned int)t.length));
                cpuRealigners[0]->initQuery(qSeqs[0]);
                lcaResults[b][refused2[r].second] = cpuRealigners[0]->getSWResult(dbSeqs[0], INT_MAX, false, al.covMode, al.realignCov,
                                                                                  block3[b].evalThr, al.lcaSwMode, al.seqIdMode, false);
            }
            watch.lap("lca block");
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
            for (size_t b = 0; b < nq; b++) {
                if (accepted[b].empty()) continue;
                const double topHitEval = block3[b].evalThr;
                std::vector<Matcher::result_t> out;
                unsigned int rejected = 0;
                for (size_t k = 0; k < lists[b].size() && rejected < al.maxReject; k++) {
                    Matcher::result_t &res = lcaResults[b][k];
                    if (Alignment::checkCriteria(res, false, topHitEval, al.seqIdThr, al.alnLenThr, al.covMode, al.realignCov)) {
                        out.emplace_back(res);
                        rejected = 0;
                    } else {
                        rejected++;
                    }
                }
                if (out.size() > 1) {
                    SORT_SERIAL(out.begin(), out.end(), Matcher::compareHits);
                }
                accepted[b].swap(out);
            }
        }
        // ---- serialise, write
#pragma omp parallel num_threads(threads)
        {
            unsigned int thread_idx = 0;
#ifdef OPENMP
            thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
            std::string alnResultsOutString;
            alnResultsOutString.reserve(1024 * 1024);
            char buffer[1024 + 32768 * 4];
#pragma omp for schedule(dynamic, 5)
            for (size_t b = 0; b < nq; b++) {
                progress.updateProgress();
                const std::vector<Matcher::result_t> &swResults = accepted[b];
                if (al.alignmentOutputMode == Parameters::ALIGNMENT_OUTPUT_CLUSTER) {
                    for (size_t result = 0; result < swResults.size(); result++) {
                        alnResultsOutString.append(SSTR(swResults[result].dbKey));
                        alnResultsOutString.push_back('\n');
                    }
                } else {
                    for (size_t result = 0; result < swResults.size(); result++) {
                        size_t len = Matcher::resultToBuffer(buffer, swResults[result], al.addBacktrace);
                        alnResultsOutString.append(buffer, len);
                    }
                }
                dbw.writeData(alnResultsOutString.c_str(), alnResultsOutString.length(), queryKeys[b], thread_idx);
                alnResultsOutString.clear();
            }
        }
        watch.lap("accept / sort / write");
        next = blockEnd;
        if (remap && next < end) al.prefdbr->remapData();
    }
    delete backend;
    for (size_t i = 0; i < threads; i++) {
        delete cpuRealigners[i];
        delete qSeqs[i];
        delete dbSeqs[i];
        delete cpuMatchers[i];
    }
    dbw.close(merge);

    Debug(Debug::INFO) << alignmentsNum << " alignments calculated\n";
    Debug(Debug::INFO) << totalPassedNum << " sequence pairs passed the thresholds";
    if (alignmentsNum > 0) {
        Debug(Debug::INFO) << " (" << ((float)totalPassedNum / (float)alignmentsNum) << " of overall calculated)";
    }
    Debug(Debug::INFO) << "\n";
    if (dbSize > 0) {
        size_t hits = totalPassedNum / dbSize;
        size_t hits_rest = totalPassedNum % dbSize;
        float hits_f = ((float)hits) + ((float)hits_rest) / (float)dbSize;
        Debug(Debug::INFO) << hits_f << " hits per query sequence\n";
    }
    return true;
}
