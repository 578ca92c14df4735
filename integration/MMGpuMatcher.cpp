#include "MMGpuMatcher.h"

#include <memory>
#include <mutex>
#include "MMGpuRun.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "StripedSmithWaterman.h"
#include "SubstitutionMatrix.h"
#include "Util.h"

#ifdef OPENMP
#include <omp.h>
#endif

MMGpuMatcher::MMGpuMatcher(MMGpuAlignBackend *backend, BaseMatrix *m, EvalueComputation *evaluer, bool aaBiasCorrection,
                           float aaBiasCorrectionScale, int gapOpen, int gapExtend)
    : backend(backend), m(m), evaluer(evaluer), aaBiasCorrection(aaBiasCorrection),
      aaBiasCorrectionScale(aaBiasCorrectionScale), gapOpen(gapOpen), gapExtend(gapExtend), numThreads(0), correlationScoreWeight(0.0f), deviceBlockAligner(false), needBacktraceStrings(true), blockHook(NULL),
      targetLookup(NULL), targetLookupCtx(NULL) {
    const int a = m->alphabetSize;
    tinySubMat.resize(a * a);
    subMat16.resize(a * a);
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) {
            tinySubMat[i * a + j] = (int8_t)m->subMatrix[i][j];
            subMat16[i * a + j] = (int16_t)m->subMatrix[i][j];
        }
}

int MMGpuMatcher::minScoreForEvalue(double evalThr, int queryLength) const {
    // computeEvalue falls monotonically with the score: bisect for the first score that passes
    if (evaluer->computeEvalue(32767, queryLength) > evalThr) return 32768;
    int lo = 1, hi = 32767;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (evaluer->computeEvalue(mid, queryLength) > evalThr) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

namespace {
struct Pending {
    s_align a;
    bool wantsBacktrace;   // reaches banded_sw in alignStartPosBacktrace (mode 2 and coverage ok)
    bool blockDone;        // start / backtrace / identities came from the host's block aligner
    bool refuse;           // the pair is recomputed by the host's Matcher (profile query in block-aligner range)
    bool needsBlock;       // int16-range pair that passed the gates: waits for the device's block aligner
    bool needsReverse;     // ... which declined it: the pair's reverse scan is run after the fact
    int32_t blockBt;       // index of the pair's block-aligner backtrace in the call's string store, -1 = none (a block of 10 000
                           // queries holds 3 M of these records, one in twelve with a string: the record itself stays plain data)
    uint32_t blockBtLen;   // its length when the string itself was not fetched
};
}  // namespace

bool MMGpuMatcher::alignBlock(const std::vector<Query> &queries, int covMode, float covThr, double evalThr,
                              unsigned int alignmentMode, unsigned int seqIdMode,
                              std::vector<std::vector<Matcher::result_t> > &results,
                              std::vector<std::pair<size_t, size_t> > *refusedPairs) {
    const size_t nq = queries.size();
    int nthreads = 1;
#ifdef OPENMP
    nthreads = numThreads > 0 ? (int)numThreads : omp_get_max_threads();
#endif
    if (refusedPairs) refusedPairs->clear();
    results.assign(nq, std::vector<Matcher::result_t>());
    // ---- per query: rounded composition bias as ssw_init builds it (StripedSmithWaterman.cpp:1364-1383) and the list
    // of the non-identity targets (identity hits never reach the aligner, Matcher.cpp:88-90)
    std::vector<std::vector<int8_t> > bias(nq);
    std::vector<std::vector<uint32_t> > ids(nq);
    std::vector<mmgpu_sw_query> dq(nq);
    std::vector<size_t> firstPair(nq + 1, 0);
    MMGpuStopwatch watch("matcher");
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for (size_t q = 0; q < nq; q++) {
        const Query &qu = queries[q];
        bias[q].assign(qu.L, 0);
        if (aaBiasCorrection && qu.profile == NULL) {      // ssw_init: no correction for profile queries (:1375-1384)
            std::vector<float> tmp(qu.L);
            SubstitutionMatrix::calcLocalAaBiasCorrection(m, qu.numSequence, qu.L, tmp.data(), aaBiasCorrectionScale);
            for (int i = 0; i < qu.L; i++)      // the statement of ssw_init, :1379 (the cast binds to the comparison)
                bias[q][i] = (int8_t)(tmp[i] < 0.0) ? tmp[i] - 0.5 : tmp[i] + 0.5;
        }
        for (size_t k = 0; k < qu.targets.size(); k++)
            if (!qu.targets[k].isIdentity) ids[q].push_back(qu.targets[k].id);
        dq[q].q = qu.numSequence;
        dq[q].qlen = (uint32_t)qu.L;
        dq[q].comp_bias = qu.profile ? NULL : bias[q].data();
        dq[q].profile = qu.profile;
        dq[q].profile_letters = qu.profile ? (uint32_t)Sequence::PROFILE_AA_SIZE : 0u;
        dq[q].target_ids = ids[q].data();
        dq[q].n_targets = (uint32_t)ids[q].size();
        // (a query with an empty prefilter list is never mapped, Alignment.cpp:322: nothing to align, no threshold)
        dq[q].min_start_score = (alignmentMode == Matcher::SCORE_ONLY || qu.L <= 0 || ids[q].empty()) ? 0
                                    : minScoreForEvalue(qu.evalThr >= 0.0 ? qu.evalThr : evalThr, qu.L);
    }
    watch.lap("composition bias + thresholds");
    for (size_t q = 0; q < nq; q++) firstPair[q + 1] = firstPair[q] + ids[q].size();
    const size_t total = firstPair[nq];
    mmgpu_sw_params par;
    par.mat = tinySubMat.data();
    par.alphabet = m->alphabetSize;
    par.gap_open = gapOpen;
    par.gap_extend = gapExtend;
    // The device computes the textbook Gotoh recurrence; the reference's striped lazy-F loop equals it only while
    // min(P) + gapExtend > -gapOpen (include/mmgpu.h, mmgpu_sw_prepare refuses a batch otherwise).  With small gap
    // penalties (--gap-open 9 --gap-extend 2) a query with a strongly negative composition bias leaves that regime: its
    // pairs go to the host's own Matcher::getSWResult as a whole (refusedPairs), the other queries of the block to the device.
    std::vector<unsigned char> hostQuery(nq, 0);
    size_t nHost = 0, nHostPairs = 0;
    {
        int minMat = 0;
        for (size_t i = 0; i < tinySubMat.size(); i++) minMat = std::min(minMat, (int)tinySubMat[i]);
        for (size_t q = 0; q < nq; q++) {
            const Query &qu = queries[q];
            int minP = minMat, minCb = 0;
            if (qu.profile != NULL) {
                minP = 0;
                for (size_t i = 0; i < (size_t)Sequence::PROFILE_AA_SIZE * (size_t)qu.L; i++) minP = std::min(minP, (int)qu.profile[i]);
            } else {
                for (int i = 0; i < qu.L; i++) minCb = std::min(minCb, (int)bias[q][i]);
            }
            if (!(minP + minCb + gapExtend > -gapOpen)) {     // (also without pairs: the device checks every query it is given)
                hostQuery[q] = 1;
                nHost++;
                if (!ids[q].empty()) nHostPairs += ids[q].size();
            }
        }
    }
    if (nHostPairs != 0 && refusedPairs == NULL) {
        err = "gap penalties too small for this matrix and query (the caller must run Matcher::getSWResult for such queries)";
        return false;
    }
    // pair p of the block is pair devPair[p] of the device batch (the same without host queries)
    std::vector<uint32_t> devPair;
    std::vector<mmgpu_sw_hit> hits(total);
    // With the block aligner on the device the reverse scan runs for the hits of the uint8 pass only: an int16-range hit takes its start
    // position from the block aligner (StripedSmithWaterman.cpp:865-882), and the few it declines get their scan afterwards (below)
    const int mode = alignmentMode == Matcher::SCORE_ONLY ? MMGPU_SW_SCORE_END : (deviceBlockAligner ? MMGPU_SW_START_NOT_WORD : MMGPU_SW_START);
    if (nHost == 0) {
        if (total && backend->align(&par, dq.data(), (uint32_t)nq, mode, hits.data()) != 0) {
            err = backend->lastError();
            return false;
        }
    } else {
        std::vector<mmgpu_sw_query> dqDev;
        devPair.assign(total, 0xFFFFFFFFu);
        size_t totalDev = 0;
        for (size_t q = 0; q < nq; q++) {
            if (hostQuery[q]) continue;
            dqDev.push_back(dq[q]);
            for (size_t k = 0; k < ids[q].size(); k++) devPair[firstPair[q] + k] = (uint32_t)(totalDev + k);
            totalDev += ids[q].size();
        }
        std::vector<mmgpu_sw_hit> hitsDev(totalDev);
        if (totalDev && backend->align(&par, dqDev.data(), (uint32_t)dqDev.size(), mode, hitsDev.data()) != 0) {
            err = backend->lastError();
            return false;
        }
        for (size_t p = 0; p < total; p++)
            if (devPair[p] != 0xFFFFFFFFu) hits[p] = hitsDev[devPair[p]];
    }

    watch.lap("device alignment (prepare, run, fetch)");
    // ---- host part of ssw_align_private (StripedSmithWaterman.cpp:846-890) per pair; pairs that go on to the
    // backtrace are collected for one traceback call
    // (plain data, every field written by the gate loop below: no 400 MB of zero-fill and string destructors on one thread)
    std::unique_ptr<Pending[]> alnStore(new Pending[total ? total : 1]);
    Pending *aln = alnStore.get();
    std::vector<std::string> btStore;      // block-aligner backtraces: host hook (rare, appended under a lock) + the device call's
    std::mutex btLock;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (size_t q = 0; q < nq; q++) {
        unsigned int thread = 0;
#ifdef OPENMP
        thread = (unsigned int)omp_get_thread_num();
#endif
        const int qlen = queries[q].L;
        size_t p = firstPair[q];
        for (size_t t = 0; t < queries[q].targets.size(); t++) {
            const Target &tg = queries[q].targets[t];
            if (tg.isIdentity) continue;
            const mmgpu_sw_hit &h = hits[p];
            const int dbLen = tg.length;
            Pending &pe = aln[p];
            s_align a;
            memset(&a, 0, sizeof(a));
            a.score1 = (uint32_t)h.score;
            a.qEndPos1 = h.q_end;
            a.dbEndPos1 = h.t_end;
            a.qStartPos1 = -1;
            a.dbStartPos1 = -1;
            a.word = h.word;
            pe.wantsBacktrace = false;
            pe.blockBt = -1;
            pe.blockBtLen = 0;
            pe.blockDone = false;
            pe.needsBlock = false;
            pe.needsReverse = false;
            pe.refuse = hostQuery[q] != 0;
            if (a.dbEndPos1 != -1 && !pe.refuse) {
                a.qCov = SmithWaterman::computeCov(0, a.qEndPos1, qlen);
                a.tCov = SmithWaterman::computeCov(0, a.dbEndPos1, dbLen);
                const bool lowCov = !Util::hasCoverage(covThr, covMode, a.qCov, a.tCov);
                a.evalue = evaluer->computeEvalue(a.score1, qlen);
                const bool lowEval = a.evalue > (queries[q].evalThr >= 0.0 ? queries[q].evalThr : evalThr);
                if (!(alignmentMode == 0 || ((alignmentMode == 2 || alignmentMode == 1) && (lowEval || lowCov)))) {
                    // word == 1: the stock reference asks the block aligner first (:865-882)
                    if (a.word == 1 && deviceBlockAligner) {
                        // one device call for all of them, below - profile queries included (round 5: the kernel takes the query's
                        // score rows in place of matrix + bias, alignStartPosBacktraceBlock<PROFILE_SEQ>)
                        pe.needsBlock = true;
                    } else if (a.word == 1 && blockHook != NULL && queries[q].profile != NULL) {
                        // the host-side hook below speaks sequences only: the pair goes back to the host's own Matcher::getSWResult
                        pe.refuse = true;
                    } else if (a.word == 1 && blockHook != NULL) {
                        s_align b = a;
                        std::string bt;
                        if (blockHook->run(thread, q, queries[q].numSequence, qlen, targetLookup(targetLookupCtx, tg.id), dbLen, b, bt)) {
                            pe.blockDone = true;
                            {
                                std::lock_guard<std::mutex> g(btLock);
                                pe.blockBt = (int32_t)btStore.size();
                                btStore.push_back(std::string());
                                btStore.back().swap(bt);
                            }
                            a = b;
                        }
                    }
                    if (!pe.blockDone && !pe.refuse && !pe.needsBlock) {
                        // alignStartPosBacktrace (:1129-1258): start positions from the reverse scan
                        a.qStartPos1 = h.q_start;
                        a.dbStartPos1 = h.t_start;
                        a.qCov = SmithWaterman::computeCov(a.qStartPos1, a.qEndPos1, qlen);
                        a.tCov = SmithWaterman::computeCov(a.dbStartPos1, a.dbEndPos1, dbLen);
                        const bool lowCov2 = !Util::hasCoverage(covThr, covMode, a.qCov, a.tCov);
                        if (!(alignmentMode == 1 || lowCov2)) pe.wantsBacktrace = true;
                    }
                }
            }
            pe.a = a;
            p++;
        }
    }
    watch.lap("E-value / coverage gates + block hook");
    // ---- int16-range pairs: the block aligner on the device (mmgpu_sw_block_backtrace; StripedSmithWaterman.cpp:865-882,
    // 943-1127).  OK: start / identities / backtrace are the block aligner's.  DECLINED ("Block alignment failed"): the
    // reference's own fallback, i.e. the reverse scan's start positions and the banded traceback.  TOO_LARGE: the host's
    // block aligner if one is installed, else the host's Matcher for the whole pair.
    if (deviceBlockAligner) {
        std::vector<uint32_t> blkPairs, blkBlockPair;
        for (size_t p = 0; p < total; p++)
            if (aln[p].needsBlock) {
                blkPairs.push_back(devPair.empty() ? (uint32_t)p : devPair[p]);
                blkBlockPair.push_back((uint32_t)p);
            }
        std::vector<mmgpu_sw_block> blk(blkPairs.size());
        std::string blkStrings;
        if (!blkPairs.empty() && backend->blockBacktrace(blkPairs.data(), (uint32_t)blkPairs.size(), blk.data(), blkStrings,
                                                         needBacktraceStrings ? MMGpuAlignBackend::BLOCK_STRINGS
                                                                              : (alignmentMode == Matcher::SCORE_COV_SEQID ? MMGpuAlignBackend::BLOCK_IDENT : MMGpuAlignBackend::BLOCK_STARTS)) != 0) {
            err = backend->lastError();
            return false;
        }
        std::vector<uint32_t> pairQuery(blkPairs.empty() ? 0 : total), pairTarget(blkPairs.empty() ? 0 : total);
        if (!blkPairs.empty())
            for (size_t q = 0; q < nq; q++) {
                size_t p = firstPair[q];
                for (size_t t = 0; t < queries[q].targets.size(); t++) {
                    if (queries[q].targets[t].isIdentity) continue;
                    pairQuery[p] = (uint32_t)q;
                    pairTarget[p] = (uint32_t)t;
                    p++;
                }
            }
        const size_t btBase = btStore.size();
        btStore.resize(btBase + blkBlockPair.size());
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
        for (size_t k = 0; k < blkBlockPair.size(); k++) {
            unsigned int thread = 0;
#ifdef OPENMP
            thread = (unsigned int)omp_get_thread_num();
#endif
            const size_t p = blkBlockPair[k];
            Pending &pe = aln[p];
            const size_t q = pairQuery[p];
            const Target &tg = queries[q].targets[pairTarget[p]];
            const int qlen = queries[q].L, dbLen = tg.length;
            const mmgpu_sw_block &bk = blk[k];
            s_align &a = pe.a;
            if (bk.status == MMGPU_BLOCK_OK) {
                a.qStartPos1 = bk.q_start;
                a.dbStartPos1 = bk.t_start;
                a.identicalAACnt = bk.ident;
                a.qCov = SmithWaterman::computeCov(a.qStartPos1, a.qEndPos1, qlen);       // :1114-1115
                a.tCov = SmithWaterman::computeCov(a.dbStartPos1, a.dbEndPos1, dbLen);
                if (!blkStrings.empty()) btStore[btBase + k].assign(blkStrings, (size_t)bk.bt_off, (size_t)bk.bt_len);
                pe.blockBt = (int32_t)(btBase + k);
                pe.blockBtLen = bk.bt_len;
                pe.blockDone = true;
                continue;
            }
            if (bk.status == MMGPU_BLOCK_TOO_LARGE) {
                if (blockHook == NULL) { pe.refuse = true; continue; }
                s_align b = a;
                std::string bt;
                if (blockHook->run(thread, q, queries[q].numSequence, qlen, targetLookup(targetLookupCtx, tg.id), dbLen, b, bt)) {
                    pe.blockDone = true;
                    btStore[btBase + k].swap(bt);
                    pe.blockBt = (int32_t)(btBase + k);
                    a = b;
                    continue;
                }
            }
            pe.needsReverse = true;      // DECLINED (or the host's block aligner declined as well)
        }
        // alignStartPosBacktrace (:1129-1258) for those: the batch ran without their reverse scan (MMGPU_SW_START_NOT_WORD)
        std::vector<uint32_t> revPairs, revBlockPair;
        for (size_t k = 0; k < blkBlockPair.size(); k++)
            if (aln[blkBlockPair[k]].needsReverse) {
                revPairs.push_back(blkPairs[k]);
                revBlockPair.push_back(blkBlockPair[k]);
            }
        if (!revPairs.empty()) {
            std::vector<mmgpu_sw_hit> rev(revPairs.size());
            if (backend->reversePairs(revPairs.data(), (uint32_t)revPairs.size(), rev.data()) != 0) {
                err = backend->lastError();
                return false;
            }
            for (size_t k = 0; k < revBlockPair.size(); k++) {
                const size_t p = revBlockPair[k];
                Pending &pe = aln[p];
                const size_t q = pairQuery[p];
                const int qlen = queries[q].L, dbLen = queries[q].targets[pairTarget[p]].length;
                hits[p] = rev[k];
                s_align &a = pe.a;
                a.qStartPos1 = rev[k].q_start;
                a.dbStartPos1 = rev[k].t_start;
                a.qCov = SmithWaterman::computeCov(a.qStartPos1, a.qEndPos1, qlen);
                a.tCov = SmithWaterman::computeCov(a.dbStartPos1, a.dbEndPos1, dbLen);
                const bool lowCov2 = !Util::hasCoverage(covThr, covMode, a.qCov, a.tCov);
                if (!(alignmentMode == 1 || lowCov2)) pe.wantsBacktrace = true;
            }
        }
        watch.lap("device block aligner (int16-range pairs)");
    }
    std::vector<uint32_t> btPairs, btBlockPair;      // device pair index / pair index inside the block
    for (size_t p = 0; p < total; p++)
        if (aln[p].wantsBacktrace) {
            btPairs.push_back(devPair.empty() ? (uint32_t)p : devPair[p]);
            btBlockPair.push_back((uint32_t)p);
        }
    std::vector<mmgpu_sw_bt> btInfo(btPairs.size());
    std::string btStrings;
    if (!btPairs.empty() && backend->traceback(btPairs.data(), (uint32_t)btPairs.size(), btInfo.data(), btStrings) != 0) {
        err = backend->lastError();
        return false;
    }
    watch.lap("device traceback");
    std::vector<int> btOf(total, -1);
    for (size_t i = 0; i < btBlockPair.size(); i++) btOf[btBlockPair[i]] = (int)i;

    // ---- Matcher::getSWResult's tail (Matcher.cpp:93-143) ----
    bool refused = false;
    std::vector<unsigned char> refusedFlag(total, 0);
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for (size_t q = 0; q < nq; q++) {
        const Query &qs = queries[q];
        const int origQueryLen = qs.L;
        size_t p = firstPair[q];
        results[q].reserve(qs.targets.size());
        for (size_t t = 0; t < qs.targets.size(); t++) {
            const Target &tg = qs.targets[t];
            s_align a;
            std::string backtrace;
            size_t btLen = 0;      // backtrace.size(), also when the string was left on the device
            if (tg.isIdentity) {      // (callers do not mark identity hits for profile queries: MMGpuAlignRun::usable)
                // SmithWaterman::scoreIdentical (StripedSmithWaterman.cpp:1770-1805): the diagonal of the word profile
                memset(&a, 0, sizeof(a));
                a.qStartPos1 = alignmentMode == 0 ? -1 : 0;
                a.dbStartPos1 = a.qStartPos1;
                a.qEndPos1 = tg.length - 1;
                a.dbEndPos1 = tg.length - 1;
                a.qCov = 1.0f;
                a.tCov = 1.0f;
                short score = 0;
                for (int pos = 0; pos < tg.length; pos++) {
                    score += (short)(tinySubMat[qs.numSequence[pos] * m->alphabetSize + tg.numSequence[pos]] + bias[q][pos]);
                    backtrace.push_back('M');
                }
                a.score1 = (uint32_t)score;
                a.evalue = evaluer->computeEvalue(a.score1, origQueryLen);
                a.identicalAACnt = (uint32_t)tg.length;
            } else {
                a = aln[p].a;
                if (aln[p].refuse) {
                    refused = true;
                    refusedFlag[p] = 1;
                } else if (aln[p].blockDone) {
                    if (aln[p].blockBt >= 0) backtrace.swap(btStore[(size_t)aln[p].blockBt]);
                    btLen = backtrace.empty() ? aln[p].blockBtLen : backtrace.size();
                } else if (btOf[p] >= 0) {
                    const mmgpu_sw_bt &bi = btInfo[btOf[p]];
                    if (bi.status == MMGPU_BT_OK) {
                        backtrace.assign(btStrings, (size_t)bi.bt_off, (size_t)bi.bt_len);
                        a.identicalAACnt = bi.ident;
                        if (correlationScoreWeight > 0.0f) {
                            // computerBacktrace's scorePerCol (:1291) and computeCorrelationScore (:1338-1362)
                            std::vector<int8_t> col;
                            col.reserve(backtrace.size());
                            int qp = a.qStartPos1, tp = a.dbStartPos1;
                            for (size_t c = 0; c < backtrace.size(); c++) {
                                if (backtrace[c] == 'M') {
                                    col.push_back((int8_t)(tinySubMat[qs.numSequence[qp] * m->alphabetSize + tg.numSequence[tp]] + bias[q][qp]));
                                    qp++;
                                    tp++;
                                } else if (backtrace[c] == 'I') {
                                    qp++;
                                } else {
                                    tp++;
                                }
                            }
                            int c1 = 0, c2 = 0, c3 = 0, c4 = 0;
                            const size_t len = col.size();
                            for (size_t st = 1; st < len; st++) {
                                c1 += col[st] * col[st - 1];
                                if (st >= 2) c2 += col[st] * col[st - 2];
                                if (st >= 3) c3 += col[st] * col[st - 3];
                                if (st >= 4) c4 += col[st] * col[st - 4];
                            }
                            a.score1 += static_cast<float>(c1 + c2 + c3 + c4) * correlationScoreWeight;     // the statement of :1251
                            // (`query_length` is the aligned query span by then: the function re-uses the variable at :1221)
                            a.evalue = evaluer->computeEvalue(a.score1, a.qEndPos1 - a.qStartPos1 + 1);
                        }
                    } else {
                        refused = true;          // every thread writes the same value
                        refusedFlag[p] = 1;
                    }
                }
                p++;
            }
            float qcov = 0.0f, dbcov = 0.0f, seqId = 0.0f;
            const unsigned int qStartPos = a.qStartPos1, dbStartPos = a.dbStartPos1, qEndPos = a.qEndPos1, dbEndPos = a.dbEndPos1;
            if (alignmentMode == Matcher::SCORE_COV || alignmentMode == Matcher::SCORE_COV_SEQID) {
                qcov = a.qCov;
                dbcov = a.tCov;
            }
            unsigned int alnLength = Matcher::computeAlnLength(qStartPos, qEndPos, dbStartPos, dbEndPos);
            if (backtrace.size() > 0) btLen = backtrace.size();
            if (alignmentMode == Matcher::SCORE_COV_SEQID) {
                if (btLen > 0) alnLength = btLen;
                seqId = Util::computeSeqId(seqIdMode, a.identicalAACnt, origQueryLen, tg.length, alnLength);
            } else if (alignmentMode == Matcher::SCORE_COV) {
                const unsigned int qAlnLen = std::max(qEndPos - qStartPos, static_cast<unsigned int>(1));
                const unsigned int dbAlnLen = std::max(dbEndPos - dbStartPos, static_cast<unsigned int>(1));
                seqId = Matcher::estimateSeqIdByScorePerCol(a.score1, qAlnLen, dbAlnLen);
            } else if (alignmentMode == Matcher::SCORE_ONLY) {
                const unsigned int qAlnLen = std::max(qEndPos, static_cast<unsigned int>(1));
                const unsigned int dbAlnLen = std::max(dbEndPos, static_cast<unsigned int>(1));
                seqId = Matcher::estimateSeqIdByScorePerCol(a.score1, qAlnLen, dbAlnLen);
            }
            const int bitScore = static_cast<int>(evaluer->computeBitScore(a.score1) + 0.5);
            results[q].push_back(Matcher::result_t(tg.dbKey, bitScore, qcov, dbcov, seqId, a.evalue, alnLength, qStartPos, qEndPos,
                                                   origQueryLen, dbStartPos, dbEndPos, tg.length, backtrace));
        }
    }
    watch.lap("result_t records");
    if (refused) {
        // MMGPU_BT_TOO_LARGE / MMGPU_BT_FAILED: the caller runs Matcher::getSWResult for these pairs
        if (refusedPairs == NULL) {
            err = "a backtrace was refused by the device (band too large): run Matcher::getSWResult for this pair";
            return false;
        }
        for (size_t q = 0; q < nq; q++) {
            size_t p = firstPair[q];
            for (size_t t = 0; t < queries[q].targets.size(); t++) {
                if (queries[q].targets[t].isIdentity) continue;
                if (refusedFlag[p]) refusedPairs->push_back(std::make_pair(q, t));
                p++;
            }
        }
    }
    return true;
}
