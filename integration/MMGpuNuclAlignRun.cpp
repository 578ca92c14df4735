// The hook behind Alignment::run (MMGpuAlignRun.cpp) for nucleotide databases: BandedNucleotideAligner::align
// (src/alignment/BandedNucleotideAligner.cpp:76-263, called by Matcher::getSWResult, Matcher.cpp:72-79) on libmmgpu - one
// mmgpu_nucl_align call per bucket of queries, getSWResult's tail (Matcher.cpp:91-146) on the results; the reference's loop
// takes them at getSWResult's call site (MMGpuAlignRun::take).
//
// The letter past the end.  The reference reverses both sequences with SmithWaterman::seq_reverse(dst, src, L) where the
// function expects L - 1 (BandedNucleotideAligner.cpp:61,68,93): the reversed copies start with src[L], one residue past
// the end of Sequence::numSequence (query strand, target) or of the aligner's queryRevCompSeq buffer (reverse strand) -
// whatever an earlier, longer sequence left at that index of the same buffer (buffers are per thread, malloc'ed:
// Sequence.cpp:44, BandedNucleotideAligner.cpp:20-27).  The letter decides the first column of the left extension, so it can
// change a result.  The stock loop therefore depends on which sequences ONE thread mapped before, in order; with one thread
// that order is the order of the prefilter database, with several it depends on the OpenMP schedule.  This file replays the
// one-thread history (BufferHistory below: what sits at an index of a buffer after a sequence of overwrites) and hands every
// pair its two letters (mmgpu_nucl_pair::past_end); memory no sequence has written yet counts as 0 (= 'A'), which is what a
// fresh heap gives the reference.  Output = `mmseqs align --threads 1` of the stock binary, whatever --threads is here.
//
// --wrapped-scoring (circular sequences) is served: the query goes to the device written twice, as Alignment.cpp:332-337 hands it to
// the matcher (mmgpu_nucl_params::wrapped), coverage / E-value / identity use the original length (Matcher.cpp:68).
// Not served (the reference's loop computes, MMGpuAlignRun::usableNucleotide): --realign, --alt-ali, lcaalign
// and finite --max-accept / --max-rejected (where the loop stops, and what it aligns a second time, decides what the buffers
// hold for the next query).
#include <climits>
#include <cstring>
#include <string>
#include <vector>

#include "Alignment.h"
#include "Debug.h"
#include "EvalueComputation.h"
#include "NucleotideMatrix.h"
#include "QueryMatcher.h"
#include "StripedSmithWaterman.h"
#include "Util.h"

#include "MMGpuAlignSession.h"
#include "MMGpuBufferHistory.h"

#ifdef OPENMP
#include <omp.h>
#endif

namespace {

struct Entry {          // one line of a prefilter list (:345-360)
    DBKeyType dbKey;
    unsigned int dbId;
    unsigned short diagonal;
    bool reverse;
    bool covered;       // Util::canBeCovered (:370)
};

}  // namespace

struct MMGpuNuclState {
    int8_t mat[25];
    uint8_t rev[5];
    mmgpu_nucl_params par;
    // what one thread of the reference would have in its three buffers
    BufferHistory qHistory, rcHistory, tHistory;      // (the query histories keep copies: a bucket's queries are freed with the bucket)
    MMGpuNuclState() : qHistory(true), rcHistory(true), tHistory(false) {}
};

bool MMGpuAlignRun::usableNucleotide(const Alignment &a) {
    const bool nucl = Parameters::isEqualDbtype(a.querySeqType, Parameters::DBTYPE_NUCLEOTIDES) &&
                      Parameters::isEqualDbtype(a.targetSeqType, Parameters::DBTYPE_NUCLEOTIDES);
    if (!nucl) return false;
    if (getenv("MMGPU_NUCL_ALIGN") != NULL && getenv("MMGPU_NUCL_ALIGN")[0] == '0') return false;
    // (--wrapped-scoring runs on the device since round 4: mmgpu_nucl_params::wrapped)
    return !a.realign && a.altAlignment == 0 && !a.lcaAlign && a.maxAccept == INT_MAX && a.maxReject == INT_MAX && a.m->alphabetSize == 5;
}

void MMGpuAlignRun::beginNucleotide(MMGpuAlignSession *s) {
    Alignment &al = s->al;
    Debug(Debug::INFO) << "MMGPU: nucleotide alignment on the device (results = the reference's loop with one thread; MMGPU_NUCL_ALIGN=0 keeps the CPU loop)\n";
    MMGpuNuclState *n = new MMGpuNuclState();
    NucleotideMatrix *nm = static_cast<NucleotideMatrix *>(al.m);
    for (int i = 0; i < 5; i++) {
        for (int j = 0; j < 5; j++) n->mat[i * 5 + j] = (int8_t)al.m->subMatrix[i][j];      // BandedNucleotideAligner.cpp:28-34
        n->rev[i] = (uint8_t)nm->reverseResidue(i);
    }
    n->par.mat = n->mat;
    n->par.reverse = n->rev;
    n->par.gap_open = al.gapOpen;
    n->par.gap_extend = al.gapExtend;
    n->par.zdrop = al.zdrop;
    n->par.past_end_query = 0;
    n->par.past_end_target = 0;
    n->par.wrapped = al.wrappedScoring ? 1 : 0;
    s->nucl = n;
}

void MMGpuAlignRun::endNucleotide(MMGpuAlignSession *s) {
    if (s->nucl == NULL) return;
    delete s->nucl;
    s->nucl = NULL;
}

void MMGpuAlignRun::planNucleotide(MMGpuAlignSession *s) {
    Alignment &al = s->al;
    MMGpuNuclState &N = *s->nucl;
    NucleotideMatrix *nm = static_cast<NucleotideMatrix *>(al.m);
    const std::vector<uint64_t> &tOff = s->targetOffsets;
    const std::vector<unsigned char> &tRes = s->targetResidues;
    const size_t nq = s->size, next = s->start;
    std::vector<std::vector<unsigned char> > queryNum(nq);
    std::vector<std::vector<Entry> > lists(nq);
    std::vector<DBKeyType> queryKeys(nq, 0);

    // ---- parse: the list walk of :316-375 without the alignment
#pragma omp parallel num_threads(al.threads)
    {
        unsigned int thread_idx = 0;
#ifdef OPENMP
        thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
        Sequence qSeq(al.maxSeqLen, al.querySeqType, al.m, 0, false, false);
        char buffer[1024];
        const char *words[10];
#pragma omp for schedule(dynamic, 5)
        for (size_t b = 0; b < nq; b++) {
            const size_t id = next + b;
            char *data = al.prefdbr->getData(id, thread_idx);
            const DBKeyType queryDbKey = al.prefdbr->getDbKey(id);
            queryKeys[b] = queryDbKey;
            size_t origQueryLen = 0;
            if (*data != '\0') {
                const size_t qId = al.qdbr->getId(queryDbKey);
                char *querySeqData = al.qdbr->getData(qId, thread_idx);
                if (querySeqData == NULL) continue;      // (the loop reports it)
                origQueryLen = al.qdbr->getSeqLen(qId);
                if (al.wrappedScoring) {      // :332-337: the query written twice
                    std::string twice(querySeqData, origQueryLen);
                    twice += twice;
                    qSeq.mapSequence(qId, queryDbKey, twice.c_str(), origQueryLen * 2);
                } else {
                    qSeq.mapSequence(qId, queryDbKey, querySeqData, origQueryLen);
                }
                queryNum[b].assign(qSeq.numSequence, qSeq.numSequence + qSeq.L);
            }
            while (*data != '\0') {
                Util::parseKey(data, buffer);
                Entry e;
                e.dbKey = Util::fast_atoi<DBKeyType>(buffer);
                const size_t elements = Util::getWordsOfLine(data, words, 10);
                e.diagonal = 0;
                e.reverse = false;
                if (elements == 3) {
                    hit_t hit = QueryMatcher::parsePrefilterHit(data);
                    e.reverse = al.reversePrefilterResult && (hit.prefScore < 0);
                    e.diagonal = (unsigned short)static_cast<short>(hit.diagonal);
                }
                data = Util::skipLine(data);
                const size_t dbId = al.tdbr->getId(e.dbKey);
                if (dbId >= al.tdbr->getSize() || al.tdbr->getData(dbId, thread_idx) == NULL) break;      // (the loop reports it and ends the run)
                e.dbId = (unsigned int)dbId;
                e.covered = Util::canBeCovered(al.canCovThr, al.covMode, static_cast<float>(origQueryLen),
                                               static_cast<float>(tOff[dbId + 1] - tOff[dbId]));
                lists[b].push_back(e);
            }
        }
    }
    s->watch.lap("parse bucket");

    // ---- the buffers of one reference thread, query by query and line by line (:337-338 mapSequence + initQuery, :368 mapSequence)
    std::vector<mmgpu_nucl_query> dq;
    std::vector<mmgpu_nucl_pair> pairs;
    std::vector<size_t> firstPair(nq + 1, 0);
    uint64_t btCap = 16;
    for (size_t b = 0; b < nq; b++) {
        firstPair[b] = pairs.size();
        if (lists[b].empty()) continue;
        const std::vector<unsigned char> &q = queryNum[b];
        const size_t L = q.size();
        N.qHistory.map(q.data(), L);
        N.rcHistory.map(q.data(), L);
        const unsigned char *os;
        size_t ol;
        unsigned int pastForward = 0, pastReverse = 0;
        if (N.qHistory.owner(L, &os, &ol)) pastForward = os[L];
        // queryRevCompSeq[(len - 1) - pos] = reverseResidue(numSequence[pos]) (:64-67)
        if (N.rcHistory.owner(L, &os, &ol)) pastReverse = (unsigned int)nm->reverseResidue(os[ol - 1 - L]);
        mmgpu_nucl_query nqy;
        nqy.q = q.data();
        nqy.qlen = (uint32_t)L;
        const uint32_t qIndex = (uint32_t)dq.size();
        dq.push_back(nqy);
        for (size_t k = 0; k < lists[b].size(); k++) {
            Entry &e = lists[b][k];
            const size_t tl = (size_t)(tOff[e.dbId + 1] - tOff[e.dbId]);
            N.tHistory.map(tRes.data() + tOff[e.dbId], tl);
            if (!e.covered) continue;
            unsigned int pastTarget = 0;
            if (N.tHistory.owner(tl, &os, &ol)) pastTarget = os[tl];
            mmgpu_nucl_pair p;
            p.query = qIndex;
            p.target = e.dbId;
            p.diagonal = e.diagonal;
            p.reverse = e.reverse ? 1 : 0;
            p.past_end = MMGPU_NUCL_PAST_END(e.reverse ? pastReverse : pastForward, pastTarget);
            pairs.push_back(p);
            btCap += (uint64_t)L + tl + 2;
        }
    }
    firstPair[nq] = pairs.size();
    s->watch.lap("buffer history");

    // ---- one device call for the bucket
    std::vector<mmgpu_nucl_hit> hits(pairs.size());
    std::vector<char> bt(btCap);
    uint64_t btUsed = 0;
    if (!pairs.empty() && mmgpu_nucl_align(s->gpu, &N.par, dq.data(), (uint32_t)dq.size(), pairs.data(), (uint32_t)pairs.size(), hits.data(),
                                           bt.data(), btCap, &btUsed) != 0) {
        Debug(Debug::ERROR) << "MMGPU: " << mmgpu_last_error() << "\n";
        EXIT(EXIT_FAILURE);
    }
    s->watch.lap("mmgpu_nucl_align");

    // ---- getSWResult's tail per pair: BandedNucleotideAligner.cpp:224-231, Matcher.cpp:91-137 with alignmentMode = SCORE_COV_SEQID (:78)
    s->results.assign(nq, std::vector<Matcher::result_t>());
    EvalueComputation &evaluer = s->evaluer;
#pragma omp parallel for schedule(dynamic, 5) num_threads(al.threads)
    for (size_t b = 0; b < nq; b++) {
        const int queryLen = (int)queryNum[b].size();      // (the doubled length with wrapped scoring)
        const int origQueryLen = al.wrappedScoring ? queryLen / 2 : queryLen;      // Matcher.cpp:68
        size_t pi = firstPair[b];
        std::vector<Matcher::result_t> &out = s->results[b];
        out.reserve(firstPair[b + 1] - firstPair[b]);
        for (size_t k = 0; k < lists[b].size(); k++) {
            const Entry &e = lists[b][k];
            if (!e.covered) continue;
            const mmgpu_nucl_hit &h = hits[pi++];
            if (h.status != MMGPU_NUCL_OK) {
                Debug(Debug::ERROR) << "MMGPU: backtrace buffer too small for query " << queryKeys[b] << "\n";
                EXIT(EXIT_FAILURE);
            }
            const int dbLen = (int)(tOff[e.dbId + 1] - tOff[e.dbId]);
            float qcov = SmithWaterman::computeCov(h.q_start, h.q_end, queryLen);
            if (al.wrappedScoring) qcov = std::min(1.0f, qcov * 2);      // BandedNucleotideAligner.cpp:146-147,225-227
            const float dbcov = SmithWaterman::computeCov(h.t_start, h.t_end, dbLen);
            const double evalue = evaluer.computeEvalue(h.score, origQueryLen);
            std::string backtrace(bt.data() + h.bt_off, h.bt_len);
            unsigned int alnLength = Matcher::computeAlnLength(h.q_start, h.q_end, h.t_start, h.t_end);
            if (backtrace.size() > 0) alnLength = backtrace.size();
            const float seqId = Util::computeSeqId(al.seqIdMode, h.ident, origQueryLen, dbLen, alnLength);
            const int bitScore = static_cast<int>(evaluer.computeBitScore(h.score) + 0.5);
            if (e.reverse) out.emplace_back(Matcher::result_t(e.dbKey, bitScore, qcov, dbcov, seqId, evalue, alnLength, h.q_start, h.q_end, origQueryLen,
                                                               h.t_end, h.t_start, dbLen, backtrace));
            else out.emplace_back(Matcher::result_t(e.dbKey, bitScore, qcov, dbcov, seqId, evalue, alnLength, h.q_start, h.q_end, origQueryLen,
                                                    h.t_start, h.t_end, dbLen, backtrace));
        }
    }
    s->watch.lap("result_t records");
}
