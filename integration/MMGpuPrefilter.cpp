#include <cstring>
#include "MMGpuPrefilter.h"
#include "MMGpuRun.h"

#include <climits>

#include "BaseMatrix.h"
#include "Debug.h"
#include "SubstitutionMatrix.h"
#include "Util.h"

unsigned int MMGpuPrefilter::referenceBins(size_t dbsize) {
    const uint64_t l2CacheSize = Util::getL2CacheSize();
    for (unsigned int b = 2; b <= 1024; b *= 2)
        if (dbsize / b < l2CacheSize) return b;
    return 2048;
}

MMGpuPrefilter::MMGpuPrefilter(mmgpu_ctx *gpu, BaseMatrix *kmerSubMat, BaseMatrix *ungappedSubMat, bool aaBiasCorrection,
                               float aaBiasCorrectionScale)
    : gpu(gpu), multi(NULL), kmerSubMat(kmerSubMat), ungappedSubMat(ungappedSubMat), aaBiasCorrection(aaBiasCorrection),
      aaBiasCorrectionScale(aaBiasCorrectionScale), dbSize(0), exactKmerMatching(false), nucleotideSearch(false), kmerScore(false) {
    memset(handedBack, 0, sizeof(handedBack));
    rerunUnsplit = 0;
}

bool MMGpuPrefilter::loadIndex(IndexTable *indexTable, SequenceLookup *sequenceLookup, ScoreMatrix &threeMer, ScoreMatrix &twoMer,
                               bool spacedKmer) {
    // targets: the (masked) numeric residues the ungapped scorer reads, exactly SequenceLookup's arrays
    const size_t n = sequenceLookup->getSequenceCount();
    std::vector<uint64_t> offsets(n + 1);
    for (size_t i = 0; i <= n; i++) offsets[i] = sequenceLookup->getOffsets()[i];
    if (mmgpu_load_targets(gpu, reinterpret_cast<const uint8_t *>(sequenceLookup->getData()), offsets.data(), (uint32_t)n,
                           kmerSubMat->alphabetSize) != 0) {
        err = mmgpu_last_error();
        return false;
    }
    dbSize = n;
    const int a = ungappedSubMat->alphabetSize;
    std::vector<int8_t> ungapped(a * a);
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) ungapped[i * a + j] = (int8_t)ungappedSubMat->subMatrix[i][j];
    static_assert(sizeof(size_t) == sizeof(uint64_t), "IndexTable::getOffsets() is handed over as it is");
    mmgpu_pf_index ix;
    memset(&ix, 0, sizeof(ix));
    ix.kmer_size = indexTable->getKmerSize();
    ix.alphabet = kmerSubMat->alphabetSize;
    ix.spaced = spacedKmer ? 1 : 0;
    // (no similar-k-mer tables where Prefiltering built none - nucleotide searches - or where the queries match exactly anyway
    // - profile targets, --target-search-mode 1, --exact-kmer-matching: the library then serves exact k-mers only, for k = 4 .. 15)
    const bool tables = !exactKmerMatching && threeMer.isValid();
    ix.score3 = tables ? threeMer.score : NULL;
    ix.index3 = tables ? threeMer.index : NULL;
    ix.row3 = tables ? threeMer.rowSize : 0;
    ix.score2 = tables && twoMer.isValid() ? twoMer.score : NULL;
    ix.index2 = tables && twoMer.isValid() ? twoMer.index : NULL;
    ix.row2 = tables && twoMer.isValid() ? twoMer.rowSize : 0;
    ix.kmer_alphabet = indexTable->getAlphabetSize();      // (the full alphabet where the targets are profiles, Prefiltering.cpp:560-563)
    ix.offsets = reinterpret_cast<const uint64_t *>(indexTable->getOffsets());       // tableSize + 1 entries
    ix.entries6 = indexTable->getEntries();            // packed 6-byte IndexEntryLocal records
    ix.n_entries = indexTable->getTableEntriesNum();
    ix.ungapped_mat = ungapped.data();
    if (mmgpu_pf_load_index(gpu, &ix) != 0) {
        err = mmgpu_last_error();
        return false;
    }
    return true;
}

uint64_t MMGpuPrefilter::fingerprint(const void *p, size_t n, uint64_t h) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

uint64_t MMGpuPrefilter::indexFingerprint(int kmerSize, bool spacedKmer, int indexKmerThr, bool maskOnDevice, double maskProb,
                                          bool similarKmerTables, const int32_t *more, size_t nMore) const {
    const int a = ungappedSubMat->alphabetSize;
    std::vector<int16_t> flat((size_t)2 * a * a);
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) {
            flat[(size_t)i * a + j] = (int16_t)kmerSubMat->subMatrix[i][j];
            flat[(size_t)a * a + (size_t)i * a + j] = (int16_t)ungappedSubMat->subMatrix[i][j];
        }
    const int32_t scal[6] = {kmerSize, spacedKmer ? 1 : 0, indexKmerThr, maskOnDevice ? 1 : 0, kmerSubMat->alphabetSize, similarKmerTables ? 1 : 0};
    uint64_t fp = fingerprint(scal, sizeof(scal));
    fp = fingerprint(&maskProb, sizeof(maskProb), fp);
    if (nMore) fp = fingerprint(more, nMore * sizeof(int32_t), fp);
    return fingerprint(flat.data(), flat.size() * sizeof(int16_t), fp) | 1ull;      // (0 = "no index" in the library's calls)
}

namespace {
void scoreTables(mmgpu_pf_index &ix, int kmerSize, int alphabet, bool spacedKmer, ScoreMatrix &threeMer, ScoreMatrix &twoMer, const int8_t *ungapped) {
    memset(&ix, 0, sizeof(ix));
    ix.kmer_size = kmerSize;
    ix.alphabet = alphabet;
    ix.spaced = spacedKmer ? 1 : 0;
    ix.score3 = threeMer.isValid() ? threeMer.score : NULL;
    ix.index3 = threeMer.isValid() ? threeMer.index : NULL;
    ix.row3 = threeMer.isValid() ? threeMer.rowSize : 0;
    ix.score2 = twoMer.isValid() ? twoMer.score : NULL;
    ix.index2 = twoMer.isValid() ? twoMer.index : NULL;
    ix.row2 = twoMer.isValid() ? twoMer.rowSize : 0;
    ix.ungapped_mat = ungapped;
}
}

bool MMGpuPrefilter::loadPersisted(const Persisted &file, size_t nTargets, int kmerSize, ScoreMatrix &threeMer, ScoreMatrix &twoMer, bool spacedKmer) {
    if (multi != NULL) {
        err = "a persisted layout holds one context's database (shards are dealt per run)";
        return false;
    }
    const int a = ungappedSubMat->alphabetSize;
    std::vector<int8_t> ungapped(a * a);
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) ungapped[i * a + j] = (int8_t)ungappedSubMat->subMatrix[i][j];
    mmgpu_pf_index ix;
    scoreTables(ix, kmerSize, kmerSubMat->alphabetSize, spacedKmer, threeMer, twoMer, ungapped.data());
    if (mmgpu_db_load(gpu, file.path.c_str(), file.sourceFp, file.indexFp, &ix) != 0) {
        err = mmgpu_last_error();
        return false;
    }
    dbSize = nTargets;
    return true;
}

bool MMGpuPrefilter::buildIndex(SequenceLookup *sequenceLookup, int kmerSize, int indexKmerThr, ScoreMatrix &threeMer,
                                ScoreMatrix &twoMer, bool spacedKmer, bool maskOnDevice, double maskProb, bool logMasked, const Persisted *saveAs) {
    const size_t n = sequenceLookup->getSequenceCount();
    static_assert(sizeof(size_t) == sizeof(uint64_t), "SequenceLookup::getOffsets() is handed over as it is");
    const uint8_t *res = reinterpret_cast<const uint8_t *>(sequenceLookup->getData());
    const uint64_t *off = reinterpret_cast<const uint64_t *>(sequenceLookup->getOffsets());
    const int a = ungappedSubMat->alphabetSize;
    std::vector<int8_t> ungapped(a * a);
    std::vector<int16_t> kmer16(a * a);
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) {
            ungapped[i * a + j] = (int8_t)ungappedSubMat->subMatrix[i][j];
            kmer16[i * a + j] = (int16_t)kmerSubMat->subMatrix[i][j];
        }
    mmgpu_pf_index ix;
    scoreTables(ix, kmerSize, kmerSubMat->alphabetSize, spacedKmer, threeMer, twoMer, ungapped.data());
    // MMGPU_DB_FILE (saveAs): a file made from this database with these index parameters is loaded - no upload of the lookup, no
    // masking, no index build; anything else is built as ever and saved for the next run.  (Where the caller could tell before it
    // filled the lookup, it loaded the file itself and this function is not called: MMGpuPrefilterRun::loadPersisted.)
    if (saveAs != NULL && multi == NULL) {
        if (mmgpu_db_load(gpu, saveAs->path.c_str(), saveAs->sourceFp, saveAs->indexFp, &ix) == 0) {
            dbSize = n;
            Debug(Debug::INFO) << "MMGPU: targets, masked view and k-mer index loaded from " << saveAs->path << "\n";
            return true;
        }
        Debug(Debug::INFO) << "MMGPU: " << saveAs->path << " not usable (" << mmgpu_last_error() << "): building\n";
    }
    if ((multi ? mmgpu_multi_load_targets(multi, res, off, (uint32_t)n, kmerSubMat->alphabetSize)
               : mmgpu_load_targets(gpu, res, off, (uint32_t)n, kmerSubMat->alphabetSize)) != 0) {
        err = mmgpu_last_error();
        return false;
    }
    dbSize = n;
    if (maskOnDevice) {
        // Masker's likelihood ratios (ProbabilityMatrix over the k-mer matrix, IndexBuilder.cpp:99, BaseMatrix.h:83-101)
        ProbabilityMatrix pm(*kmerSubMat);
        const int ka = kmerSubMat->alphabetSize;
        std::vector<double> lr((size_t)ka * ka);
        for (int i = 0; i < ka; i++)
            for (int j = 0; j < ka; j++) lr[(size_t)i * ka + j] = pm.probMatrixPointers[i][j];
        uint64_t masked = 0;
        const int maskLetter = (int)kmerSubMat->aa2num[(int)'X'];
        if ((multi ? mmgpu_multi_pf_mask_targets(multi, lr.data(), ka, maskProb, maskLetter, &masked)
                   : mmgpu_pf_mask_targets(gpu, lr.data(), ka, maskProb, maskLetter, &masked)) != 0) {
            err = mmgpu_last_error();
            return false;
        }
        if (logMasked) Debug(Debug::INFO) << "Index table: Masked residues: " << masked << " (tantan on the device)\n";
    }
    if ((multi ? mmgpu_multi_pf_build_index(multi, &ix, kmer16.data(), indexKmerThr) : mmgpu_pf_build_index(gpu, &ix, kmer16.data(), indexKmerThr)) != 0) {
        err = mmgpu_last_error();
        return false;
    }
    if (saveAs != NULL && multi == NULL) {
        if (mmgpu_db_save(gpu, saveAs->path.c_str(), saveAs->sourceFp, saveAs->indexFp) == 0) Debug(Debug::INFO) << "MMGPU: device layout saved to " << saveAs->path << "\n";
        else Debug(Debug::WARNING) << "MMGPU: could not save " << saveAs->path << ": " << mmgpu_last_error() << "\n";
    }
    return true;
}

// One block in flight: what submitBlock() enqueued and finishBlock() collects.
struct MMGpuPrefilter::Pending {
    const std::vector<Query> *queries;
    std::vector<std::vector<float> > bias;      // composition bias of the queries that came without one
    std::vector<mmgpu_pf_query> dq;
    mmgpu_pf_params par;
    uint32_t stride;
    mmgpu_pf_batch_t *batch;
    mmgpu_multi_pf_batch *mb;
    bool enqueued;      // the whole block is on the device (otherwise finishBlock runs it in pieces)
    Pending() : queries(NULL), stride(0), batch(NULL), mb(NULL), enqueued(false) {}
};

void MMGpuPrefilter::compositionBias(const Query &s, std::vector<float> &bias) const {
    bias.assign(s.L, 0.0f);
    // no correction for profile and nucleotide queries (QueryMatcher.cpp:110-114: amino-acid sequences only)
    if (aaBiasCorrection && s.profile == NULL && !nucleotideSearch)
        SubstitutionMatrix::calcLocalAaBiasCorrection(kmerSubMat, s.numSequence, s.L, bias.data(), aaBiasCorrectionScale);
}

MMGpuPrefilter::Pending *MMGpuPrefilter::submitBlock(const std::vector<Query> &queries, int kmerThr, size_t maxResListLen,
                                                     unsigned int minDiagScoreThr) {
    Pending *P = new Pending();
    const size_t nq = queries.size();
    P->queries = &queries;
    if (nq == 0) return P;
    // QueryMatcher::matchQuery's composition bias over the k-mer matrix (QueryMatcher.cpp:109-117), floats
    P->bias.resize(nq);
    P->dq.resize(nq);
    for (size_t q = 0; q < nq; q++) {
        const Query &s = queries[q];
        if (s.compBias == NULL && s.profile == NULL) compositionBias(s, P->bias[q]);
        mmgpu_pf_query &d = P->dq[q];
        d.q = s.numSequence;
        d.qlen = (uint32_t)s.L;
        d.comp_bias = s.profile ? NULL : (s.compBias ? s.compBias : P->bias[q].data());
        d.identity_id = s.identityId;
        d.profile_score = s.profileScore;
        d.profile_index = s.profileIndex;
        d.profile_row = s.profileRow;
        d.profile = s.profile;
    }
    mmgpu_pf_params &par = P->par;
    par.kmer_thr = kmerThr;
    par.max_hits = (uint32_t)maxResListLen;
    par.min_diag_score = minDiagScoreThr;
    par.ref_bins = referenceBins(dbSize);
    par.exact_kmer = exactKmerMatching ? 1u : 0u;
    par.nucleotide = nucleotideSearch ? 1u : 0u;
    par.kmer_score = kmerScore ? 1u : 0u;
    P->stride = (uint32_t)std::min(maxResListLen, dbSize);
    // One device batch for the block, enqueued here and collected by finishBlock(): the caller submits the next block before it
    // collects this one, so that the device never waits for the host between blocks.
    MMGpuStopwatch watch("prefilter block");
    int rc;
    if (multi) {
        // every shard's prefilter, the exchange of the lists over the library's communicator, the merge (== unsplit lists);
        // MMGPU_PF_SHARD_INEXACT queries come back through needsCpu.  (Per-query statistics - log output only - are not
        // gathered over the shards.)
        rc = mmgpu_multi_pf_prepare(multi, &par, P->dq.data(), (uint32_t)nq, &P->mb);
        if (rc == 0) rc = mmgpu_multi_pf_run(multi, P->mb);
    } else {
        rc = mmgpu_pf_prepare(gpu, &par, P->dq.data(), (uint32_t)nq, &P->batch);
        watch.lap("mmgpu_pf_prepare");
        if (rc == 0) rc = mmgpu_pf_run(gpu, P->batch);
        if (rc == 0 && getenv("MMGPU_TRACE") != NULL && getenv("MMGPU_TRACE")[0] == '2') mmgpu_synchronize(gpu);     // the lap shows the kernels, not the enqueue
        watch.lap("mmgpu_pf_run");
    }
    P->enqueued = rc == 0;
    if (rc != 0) {      // finishBlock() runs the block in pieces
        if (P->mb) mmgpu_multi_pf_free(multi, P->mb);
        if (P->batch) mmgpu_pf_free(gpu, P->batch);
        P->mb = NULL;
        P->batch = NULL;
    }
    return P;
}

bool MMGpuPrefilter::finishBlock(Pending *P, std::vector<std::vector<hit_t> > &results, std::vector<bool> &needsCpu,
                                 std::vector<mmgpu_pf_qstat> *stats) {
    const std::vector<Query> &queries = *P->queries;
    const size_t nq = queries.size();
    results.assign(nq, std::vector<hit_t>());
    needsCpu.assign(nq, false);
    if (nq == 0) {
        delete P;
        return true;
    }
    const uint32_t stride = P->stride;
    std::vector<mmgpu_pf_hit> hits(nq * (size_t)stride);
    std::vector<uint32_t> counts(nq);
    std::vector<int32_t> status(nq);
    if (stats) stats->assign(nq, mmgpu_pf_qstat());
    MMGpuStopwatch watch("prefilter block");
    // A batch the device cannot hold (out of HBM, or 2^32 index entries and more - a large database with long or repetitive
    // queries) is cut in halves and retried, down to single queries.
    std::vector<std::pair<size_t, size_t> > todo(1, std::make_pair((size_t)0, nq));
    bool ok = true;
    while (!todo.empty() && ok) {
        const size_t lo = todo.back().first, hi = todo.back().second;
        todo.pop_back();
        const bool whole = lo == 0 && hi == nq && P->enqueued;
        int rc = 0;
        if (multi) {
            mmgpu_multi_pf_batch *mb = whole ? P->mb : NULL;
            if (!whole) {
                rc = mmgpu_multi_pf_prepare(multi, &P->par, P->dq.data() + lo, (uint32_t)(hi - lo), &mb);
                if (rc == 0) rc = mmgpu_multi_pf_run(multi, mb);
            }
            if (rc == 0) rc = mmgpu_multi_pf_fetch(multi, mb, hits.data() + lo * (size_t)stride, stride, counts.data() + lo, status.data() + lo);
            if (rc != 0) err = mmgpu_last_error();
            if (rc == 0 && mb != NULL) {
                uint32_t redone = 0;
                if (mmgpu_multi_pf_redone(mb, &redone, NULL) == 0) rerunUnsplit += redone;
            }
            if (mb) mmgpu_multi_pf_free(multi, mb);
            if (whole) P->mb = NULL;
        } else {
            mmgpu_pf_batch_t *batch = whole ? P->batch : NULL;
            if (!whole) {
                rc = mmgpu_pf_prepare(gpu, &P->par, P->dq.data() + lo, (uint32_t)(hi - lo), &batch);
                if (rc == 0) rc = mmgpu_pf_run(gpu, batch);
            }
            if (rc == 0) rc = mmgpu_pf_fetch(gpu, batch, hits.data() + lo * (size_t)stride, stride, counts.data() + lo, status.data() + lo,
                                             stats ? stats->data() + lo : NULL);
            watch.lap("mmgpu_pf_fetch");
            if (rc != 0) err = mmgpu_last_error();
            if (batch) mmgpu_pf_free(gpu, batch);
            if (whole) P->batch = NULL;
        }
        P->enqueued = false;
        if (rc != 0) {
            if ((rc == MMGPU_ERR_HIP || rc == MMGPU_ERR_UNSUPPORTED) && hi - lo > 1) {
                const size_t mid = lo + (hi - lo) / 2;
                todo.push_back(std::make_pair(mid, hi));
                todo.push_back(std::make_pair(lo, mid));
                continue;
            }
            ok = false;
        }
    }
    if (P->mb) mmgpu_multi_pf_free(multi, P->mb);
    if (P->batch) mmgpu_pf_free(gpu, P->batch);
    delete P;
    if (!ok) return false;
    for (size_t q = 0; q < nq; q++) {
        if (status[q] != MMGPU_PF_OK) {     // MMGPU_PF_OVERFLOW / MMGPU_PF_LONG_SEQ: the host's own matcher runs this query
            needsCpu[q] = true;
            handedBack[status[q] & 7]++;
            continue;
        }
        results[q].resize(counts[q]);
        for (uint32_t k = 0; k < counts[q]; k++) {
            const mmgpu_pf_hit &h = hits[q * (size_t)stride + k];
            hit_t &o = results[q][k];
            o.seqId = h.id;
            o.prefScore = h.score;
            o.diagonal = h.diagonal;
        }
    }
    watch.lap("hit_t lists");
    return true;
}

bool MMGpuPrefilter::matchBlock(const std::vector<Query> &queries, int kmerThr, size_t maxResListLen, unsigned int minDiagScoreThr,
                                std::vector<std::vector<hit_t> > &results, std::vector<bool> &needsCpu,
                                std::vector<mmgpu_pf_qstat> *stats) {
    return finishBlock(submitBlock(queries, kmerThr, maxResListLen, minDiagScoreThr), results, needsCpu, stats);
}
