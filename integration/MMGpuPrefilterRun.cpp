// The query loop of Prefiltering::runSplit (src/prefiltering/Prefiltering.cpp:820-918) with QueryMatcher::matchQuery on
// libmmgpu.
//
// Everything before the loop stays the reference's: the target split, getIndexTable (IndexBuilder::fillDatabase with
// tantan masking, or the precomputed index), the similar-k-mer score matrices.  The IndexTable, the (masked)
// SequenceLookup and the ScoreMatrix tables are handed to the device once per split (MMGpuPrefilter::loadIndex); the loop
// then runs over blocks of queries: every thread maps sequences, ONE device call returns the hit_t lists of the block
// (MMGpuPrefilter::matchBlock = the batch form of matchQuery), every thread finishes its queries with the reference's own
// statements (:876-917: local id -> key, canBeCovered, prefilterHitToBuffer, DBWriter, statistics).  Queries the device
// declines (more than 62 flushes of the databaseHits buffer; a sequence of 32768 residues or more involved, which the
// reference scores with computeLongScore) go through a host QueryMatcher.
//
// Compiled into MMseqs2 by integration/build_mmseqs.sh (HAVE_MMGPU); Prefiltering.h declares the class a friend.
#include <algorithm>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <list>
#include <mutex>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>

#include "DBWriter.h"
#include "ExtendedSubstitutionMatrix.h"
#include "FileUtil.h"
#include "IndexBuilder.h"
#include "Debug.h"
#include "Parameters.h"
#include "Prefiltering.h"
#include "QueryMatcher.h"
#include "Util.h"
#include "simd.h"

#include "MMGpuFusedSearch.h"
#include "MMGpuPrefilter.h"
#include "MMGpuRun.h"

void mmgpuFusedPrepareCapture(size_t threads);      // MMGpuFusedSearch.cpp

#ifdef OPENMP
#include <omp.h>
#endif

bool MMGpuPrefilterRun::usable(Prefiltering &p) { return usableConfig(p, true); }

bool MMGpuPrefilterRun::usableConfig(Prefiltering &p, bool indexExists) {
    if (!MMGpuRun::enabled()) return false;
    const bool profileQuery = Parameters::isEqualDbtype(p.querySeqType, Parameters::DBTYPE_HMM_PROFILE);
    const bool nucl = Parameters::isEqualDbtype(p.querySeqType, Parameters::DBTYPE_NUCLEOTIDES) &&
                      Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_NUCLEOTIDES);
    // Profile TARGETS with amino-acid queries (e.g. sequences against a Pfam-like profile database): the index holds the similar
    // k-mers of the profiles' positions (IndexBuilder.cpp:63, isTargetSimiliarKmerSearch - built on the host and handed over), the
    // lookup their consensus sequences (:128), the queries match exactly (Prefiltering.cpp:187-190) - the path of
    // --target-search-mode 1.
    const bool profileTarget = Parameters::isEqualDbtype(p.querySeqType, Parameters::DBTYPE_AMINO_ACIDS) &&
                               Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_HMM_PROFILE) && p.takeOnlyBestKmer;
    const bool aa = ((Parameters::isEqualDbtype(p.querySeqType, Parameters::DBTYPE_AMINO_ACIDS) || profileQuery) &&
                     Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS)) || profileTarget;
    const char *why = NULL;
    if (!aa && !nucl) why = "mixed database types";
    else if (indexExists && (p.indexTable == NULL || (p.sequenceLookup == NULL && !p.mmgpuPersisted))) why = "no index table / sequence lookup in memory";
    else if (nucl && !p.takeOnlyBestKmer) why = "nucleotide search without exact k-mer matching";
    else if (profileQuery && p.takeOnlyBestKmer) why = "exact k-mer matching with profile queries";
    else if (!p.takeOnlyBestKmer && !profileQuery && (!p._3merSubMatrix.isValid() || !p._2merSubMatrix.isValid())) why = "no similar-k-mer score matrices";
    else if (p.diagonalScoring == 0 && (profileQuery || nucl)) why = "--diag-score 0 with profile queries / nucleotide databases";
    else if (p.minDiagScoreThr < 1 && p.diagonalScoring != 0) why = "--min-ungapped-score 0";     // (with --diag-score 0 it equals 1)
    else if (p.takeOnlyBestKmer ? (p.kmerSize < 4 || p.kmerSize > 15) : (p.kmerSize < 5 || p.kmerSize > 7)) why = "k-mer size not covered (5 / 6 / 7; 4..15 with exact k-mer matching)";
    else if (p.spacedKmerPattern.empty() == false) why = "user-defined spaced k-mer pattern";
    else if (p.ungappedSubMatAux != NULL) why = "auxiliary ungapped matrix";
    // (the constructor creates the taxonomy hook after it built the index: before that the parameter says whether it will)
    else if (indexExists ? p.taxonomyHook != NULL : Parameters::getInstance().taxonList.length() > 0) why = "taxonomy filter";
    else if (p.maxResListLen > MMGPU_PF_MAX_HITS) why = "--max-seqs above the device limit";
    if (why != NULL) {
        Debug(Debug::INFO) << "MMGPU: prefilter configuration not covered by the device path (" << why << "), using the CPU path\n";
        return false;
    }
    return true;
}

bool MMGpuPrefilterRun::deviceBuildsIndex(Prefiltering &p) {
    if (getenv("MMGPU_HOST_INDEX") != NULL && getenv("MMGPU_HOST_INDEX")[0] == '1') return false;
    if (p.templateDBIsIndex) return false;
    // --target-search-mode 1: the index holds the SIMILAR k-mers of every target position (IndexBuilder.cpp:63,
    // isTargetSimiliarKmerSearch) and the queries match exactly; that index is the host's to build, it is handed over as it is
    if (p.targetSearchMode != 0 || Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_HMM_PROFILE)) return false;
    // the same conditions run() will check at the seam, on what is known before the index exists
    const bool ok = usableConfig(p, false);
    if (ok) Debug(Debug::INFO) << "MMGPU: the k-mer index will be built on the device (MMGPU_HOST_INDEX=1 keeps the host's)\n";
    return ok;
}

ScoreMatrix MMGpuPrefilterRun::scoreMatrix(Prefiltering &p, const BaseMatrix &matrix, size_t kmerSize) {
    if (!MMGpuRun::enabled() || p.templateDBIsIndex || (kmerSize != 2 && kmerSize != 3) || matrix.alphabetSize + 1 > 32)
        return p.getScoreMatrix(matrix, kmerSize);
    // the caller took X out of the alphabet (Prefiltering.cpp:221): alphabetSize letters, subMatrix rows of the full matrix
    const int ka = matrix.alphabetSize, a = ka + 1;
    std::vector<int16_t> flat((size_t)a * a, 0);
    for (int i = 0; i < ka; i++)
        for (int j = 0; j < ka; j++) flat[(size_t)i * a + j] = (int16_t)matrix.subMatrix[i][j];
    size_t size = 1;
    for (size_t i = 0; i < kmerSize; i++) size *= (size_t)ka;
    const size_t rowSize = (size / MAX_ALIGN_INT + 1) * MAX_ALIGN_INT;      // ExtendedSubstitutionMatrix.cpp:24-25
    short *score = (short *)mem_align(MAX_ALIGN_INT, size * rowSize * sizeof(short));
    unsigned int *index = (unsigned int *)mem_align(MAX_ALIGN_INT, size * rowSize * sizeof(unsigned int));
    static_assert(sizeof(short) == sizeof(int16_t) && sizeof(unsigned int) == sizeof(uint32_t), "ScoreMatrix element types");
    if (mmgpu_host_score_matrix_rows(flat.data(), a, (int)kmerSize, rowSize, reinterpret_cast<int16_t *>(score), reinterpret_cast<uint32_t *>(index)) != 0) {
        free(score);
        free(index);
        return p.getScoreMatrix(matrix, kmerSize);
    }
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < size; r++)
        for (size_t z = size; z < rowSize; z++) {      // :50-53
            score[r * rowSize + z] = -255;
            index[r * rowSize + z] = 0;
        }
    return ScoreMatrix(score, index, size, rowSize);
}

bool MMGpuPrefilterRun::deviceMasks(Prefiltering &p) {
    const char *e = getenv("MMGPU_DEVICE_MASK");
    if (e != NULL && e[0] == '0') return false;
    // what the kernel restates: Masker::maskSequence with tantan alone (Masker.cpp:20-32) on amino-acid sequences
    if (p.maskMode != 1 || p.maskLowerCaseMode != 0 || p.maskNrepeats > 0) return false;
    if (!Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS)) return false;
    return true;
}

namespace {
// a target split of more sequences than one context indexes is dealt to several contexts on the device - where the configuration
// allows it (MMGpuPrefilter::multiCapable); 0 = no such split
int contextsForLargeSplit(size_t dbSize) {
    const size_t maxTargets = MMGpuRun::envSize("MMGPU_TEST_MAX_TARGETS", MMGPU_PF_MAX_TARGETS);      // (tests: the path with a small database)
    return MMGpuRun::deviceIds().empty() && dbSize > maxTargets ? (int)((dbSize + maxTargets - 1) / maxTargets) : 0;
}
bool largeSplitNeedsHost(Prefiltering &p, size_t dbSize, bool deviceIndex, size_t maxResListLen, int querySeqType, int targetSeqType, int diagonalScoring) {
    const int n = contextsForLargeSplit(dbSize);
    if (n <= 1) return false;
    (void)p;
    return !(deviceIndex && MMGpuPrefilter::multiCapable(Parameters::isEqualDbtype(querySeqType, Parameters::DBTYPE_HMM_PROFILE),
                                                         Parameters::isEqualDbtype(targetSeqType, Parameters::DBTYPE_NUCLEOTIDES),
                                                         diagonalScoring == 0, maxResListLen, n));
}
}

namespace {
// MMGPU_DB_FILE: the file and the fingerprints of what this split would put into it.  The source fingerprint is read off the
// DBReader - keys, lengths, the first bytes of every 64th entry - so that it can be compared before any sequence is mapped.
bool persistedLayout(Prefiltering &p, DBReader<unsigned int> *tdbr, size_t dbFrom, size_t dbSize, bool maskOnDevice, int indexKmerThr, int kmerSize,
                     bool spacedKmer, double maskProb, int maskMode, int maskLowerCaseMode, int maskNrepeats, int targetSearchMode, bool tables,
                     const MMGpuPrefilter &device, MMGpuPrefilter::Persisted *out) {
    (void)p;
    const char *file = getenv("MMGPU_DB_FILE");
    if (file == NULL || file[0] == '\0') return false;
    out->path = file;
    std::vector<uint64_t> ident(2 * dbSize + 2);
    ident[0] = dbSize;
    ident[1] = tdbr->getSize();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < dbSize; i++) {
        ident[2 + 2 * i] = tdbr->getDbKey(dbFrom + i);
        ident[3 + 2 * i] = tdbr->getSeqLen(dbFrom + i);
    }
    uint64_t fp = MMGpuPrefilter::fingerprint(ident.data(), ident.size() * sizeof(uint64_t));
    // a compressed database is never served from a persisted layout (nothing of its content could be sampled here, and the alignment
    // module of a fused search reads single sequences on demand)
    if (tdbr->isCompressed()) return false;
    // the data files themselves: size and modification time (a database rewritten in place with the same keys and lengths - edited
    // residues, another masking, another translation table - is another database), and a sample of the content
    {
        const std::vector<std::string> files = FileUtil::findDatafiles(tdbr->getDataFileName());
        std::vector<uint64_t> stamp;
        for (size_t i = 0; i < files.size(); i++) {
            struct stat st;
            if (stat(files[i].c_str(), &st) != 0) return false;
            stamp.push_back((uint64_t)st.st_size);
            stamp.push_back((uint64_t)st.st_mtim.tv_sec);
            stamp.push_back((uint64_t)st.st_mtim.tv_nsec);
        }
        stamp.push_back(files.size());
        fp = MMGpuPrefilter::fingerprint(stamp.data(), stamp.size() * sizeof(uint64_t), fp);
    }
    for (size_t i = 0; i < dbSize; i += 16) {
        const char *data = tdbr->getDataUncompressed(dbFrom + i);
        if (data != NULL) fp = MMGpuPrefilter::fingerprint(data, std::min<size_t>(tdbr->getSeqLen(dbFrom + i), 64), fp);
    }
    out->sourceFp = fp | 1ull;
    const int32_t more[5] = {maskMode, maskLowerCaseMode, maskNrepeats, targetSearchMode, (int32_t)dbFrom};
    out->indexFp = device.indexFingerprint(kmerSize, spacedKmer, indexKmerThr, maskOnDevice, maskProb, tables, more, 5);
    return true;
}
}

bool MMGpuPrefilterRun::loadPersisted(Prefiltering &p, size_t dbFrom, size_t dbSize) {
    const char *file = getenv("MMGPU_DB_FILE");
    if (file == NULL || file[0] == '\0' || !p.mmgpuDeviceIndex) return false;
    // one context, one unsplit database; sequence queries (a profile run computes its score tables in run()); an uncompressed
    // database (the alignment module of a fused search reads single sequences on demand, see MMGpuAlignRun)
    if (p.splits != 1 || !MMGpuRun::deviceIds().empty() || contextsForLargeSplit(dbSize) > 1) return false;
    if (Parameters::isEqualDbtype(p.querySeqType, Parameters::DBTYPE_HMM_PROFILE)) return false;
    // (amino-acid target databases: the layout of a nucleotide index - 4^15 offsets - is served by the library calls, but this
    // binding has only been exercised with protein databases)
    if (!Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS)) return false;
    mmgpu_db_info info;
    if (mmgpu_db_probe(file, &info) != 0) return false;      // (no file yet: run() builds and saves it)
    MMGpuPrefilter device(NULL, p.kmerSubMat, p.ungappedSubMat, p.aaBiasCorrection, p.aaBiasCorrectionScale);
    MMGpuPrefilter::Persisted layout;
    if (!persistedLayout(p, p.tdbr, dbFrom, dbSize, p.mmgpuDeviceMask, p.mmgpuIndexKmerThr, p.kmerSize, p.spacedKmer, (double)p.maskProb,
                         p.mmgpuDeviceMask ? 0 : p.maskMode, p.maskLowerCaseMode, p.maskNrepeats, p.targetSearchMode, p._3merSubMatrix.isValid(), device, &layout))
        return false;
    if (info.source_fingerprint != layout.sourceFp || info.index_fingerprint != layout.indexFp || info.n_targets != dbSize) {
        Debug(Debug::INFO) << "MMGPU: " << file << " was made from another database or with other index parameters: building (and replacing it)\n";
        return false;
    }
    MMGpuPrefilter onDevice(MMGpuRun::context(), p.kmerSubMat, p.ungappedSubMat, p.aaBiasCorrection, p.aaBiasCorrectionScale);
    if (!onDevice.loadPersisted(layout, dbSize, p.kmerSize, p._3merSubMatrix, p._2merSubMatrix, p.spacedKmer)) {
        Debug(Debug::INFO) << "MMGPU: " << file << " not usable (" << onDevice.error() << "): building\n";
        return false;
    }
    Debug(Debug::INFO) << "MMGPU: targets, masked view and k-mer index loaded from " << file << " (no sequence lookup on the host)\n";
    return true;
}

// `mmseqs makemmgpudb <targetDB> <layoutFile> [prefilter options]` (integration/MMGpuMakeDb.cpp; the shape of the reference's
// makepaddedseqdb / createindex, src/util/makepaddedseqdb.cpp:14): what the first `prefilter` / `search` with MMGPU_DB_FILE would do on
// its way - fill the lookup, hand it over, mask and index on the device, save - as a command of its own, so that no search pays for it.
// The Prefiltering object was made with the target database on both sides and MMGPU_DB_FILE = the file to write.
bool MMGpuPrefilterRun::buildAndSave(Prefiltering &p) {
    const size_t dbSize = p.tdbr->getSize();
    if (p.splits != 1 || !MMGpuRun::deviceIds().empty() || contextsForLargeSplit(dbSize) > 1) {
        Debug(Debug::ERROR) << "MMGPU: makemmgpudb persists ONE unsplit database on one device (this one needs " << p.splits << " split(s))\n";
        return false;
    }
    if (p.indexTable == NULL && !p.mmgpuPersisted) p.getIndexTable(0, 0, dbSize);      // (target-split mode builds it in runSplit)
    if (p.mmgpuPersisted) {
        Debug(Debug::INFO) << "MMGPU: the file already holds this database with these index parameters\n";
        return true;
    }
    if (!usable(p) || !p.mmgpuDeviceIndex || !Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS)) {
        Debug(Debug::ERROR) << "MMGPU: makemmgpudb covers amino-acid sequence databases whose index the device builds (see the message above)\n";
        return false;
    }
    MMGpuStopwatch watch("makemmgpudb");
    MMGpuPrefilter device(MMGpuRun::context(), p.kmerSubMat, p.ungappedSubMat, p.aaBiasCorrection, p.aaBiasCorrectionScale);
    watch.lap("open device");
    MMGpuPrefilter::Persisted layout;
    if (!persistedLayout(p, p.tdbr, 0, dbSize, p.mmgpuDeviceMask, p.mmgpuIndexKmerThr, p.kmerSize, p.spacedKmer, (double)p.maskProb,
                         p.mmgpuDeviceMask ? 0 : p.maskMode, p.maskLowerCaseMode, p.maskNrepeats, p.targetSearchMode, p._3merSubMatrix.isValid(), device,
                         &layout)) {
        Debug(Debug::ERROR) << "MMGPU: this database cannot be persisted (compressed, or its data files cannot be read)\n";
        return false;
    }
    device.setMode(p.takeOnlyBestKmer, false, p.diagonalScoring == 0);
    if (!device.buildIndex(p.sequenceLookup, p.kmerSize, p.mmgpuIndexKmerThr, p._3merSubMatrix, p._2merSubMatrix, p.spacedKmer, p.mmgpuDeviceMask,
                           (double)p.maskProb, true, &layout)) {
        Debug(Debug::ERROR) << "MMGPU: " << device.error() << "\n";
        return false;
    }
    watch.lap("hand over targets, mask, build the index on the device, save");
    mmgpu_db_info info;
    if (mmgpu_db_probe(layout.path.c_str(), &info) != 0 || info.source_fingerprint != layout.sourceFp || info.index_fingerprint != layout.indexFp) {
        Debug(Debug::ERROR) << "MMGPU: " << layout.path << " was not written (" << mmgpu_last_error() << ")\n";
        return false;
    }
    Debug(Debug::INFO) << "MMGPU: " << info.n_targets << " targets, " << info.n_entries << " index entries (k = " << info.kmer_size << "), "
                       << (info.file_bytes >> 20) << " MB in " << layout.path << "; searches find it with MMGPU_DB_FILE=" << layout.path << "\n";
    return true;
}

bool MMGpuPrefilterRun::keepsEntriesInMemory(Prefiltering &p, const std::string &resultDB, size_t dbSize) {
    return p.splits == 1 && MMGpuFusedSearch::capturing(resultDB) && usable(p) &&
           !largeSplitNeedsHost(p, dbSize, p.mmgpuDeviceIndex, p.maxResListLen, p.querySeqType, p.targetSeqType, p.diagonalScoring);
}

bool MMGpuPrefilterRun::runsUnsplitWithResidentTargets(Prefiltering &p, size_t *maxResListLen) {
    *maxResListLen = p.maxResListLen;
    // (index and lookup of an unsplit run exist once the constructor has returned: Prefiltering.cpp:196-199)
    return p.splits == 1 && (p.sequenceLookup != NULL || p.mmgpuPersisted) && p.mmgpuDeviceIndex && p.mmgpuDeviceMask && MMGpuRun::deviceIds().empty() && usable(p);
}

void MMGpuPrefilterRun::ensureHostIndex(Prefiltering &p, size_t dbFrom, size_t dbSize) {
    if (!p.mmgpuDeviceIndex) return;
    // IndexBuilder::fillDatabase as Prefiltering::getIndexTable calls it (:564-569); it also fills a second SequenceLookup,
    // which replaces the first (same content)
    Debug(Debug::INFO) << "MMGPU: building the host index for queries handed back by the device\n";
    {   // getIndexTable created the table without its offset array (the device was going to build the index): a real one now
        IndexTable *table = new IndexTable(p.indexTable->getAlphabetSize(), p.kmerSize, false);
        delete p.indexTable;
        p.indexTable = table;
    }
    Sequence tseq(p.maxSeqLen, p.targetSeqType, p.kmerSubMat, p.kmerSize, p.spacedKmer, p.aaBiasCorrection, true, p.spacedKmerPattern);
    SequenceLookup *second = NULL;
    IndexBuilder::fillDatabase(p.indexTable, &second, *p.kmerSubMat, p._3merSubMatrix, p._2merSubMatrix, &tseq, p.tdbr, dbFrom,
                               dbFrom + dbSize, p.mmgpuIndexKmerThr, p.maskMode, p.maskLowerCaseMode, p.maskProb, p.maskNrepeats,
                               p.targetSearchMode);
    if (MMGpuFusedSearch::holdsLookup(p.sequenceLookup)) p.sequenceLookup = NULL;      // (the alignment module of the fused search reads it)
    delete p.sequenceLookup;
    p.sequenceLookup = second;
    p.mmgpuDeviceIndex = false;
}

bool MMGpuPrefilterRun::run(Prefiltering &p, DBWriter &tmpDbw, size_t dbFrom, size_t dbSize, size_t queryFrom, size_t querySize,
                            char *notEmpty, std::list<int> **reslens, size_t localThreads, Debug::Progress &progress,
                            MMGpuPrefilterStats &st) {
    if (!usable(p)) {
        // the index may have been left to the device by getIndexTable (deviceBuildsIndex): the reference's loop that runs now
        // needs the host's
        ensureHostIndex(p, dbFrom, dbSize);
        return false;
    }
    MMGpuStopwatch watch("prefilter");
    mmgpu_ctx *gpu = MMGpuRun::context();
    watch.lap("open device");
    const bool profileQuery = Parameters::isEqualDbtype(p.querySeqType, Parameters::DBTYPE_HMM_PROFILE);
    // MMGPU_DEVICES: G query groups x S target shards (MMGpuRun::queryGroups).  A group is one MMGpuPrefilter: the targets of this
    // split dealt to its S contexts (sequence queries, diagonal scoring, device-built index - a host index is one table - and
    // --max-seqs x S <= 4096), or, with S = 1, the whole split and its index on the one context; query blocks are dealt to the groups.
    std::vector<MMGpuPrefilter *> devices;
    const bool nuclDb = Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_NUCLEOTIDES);
    // one device, but more targets in this split than a context indexes: as many contexts on the device as it takes (one group of
    // shards, the lists merged like those of several devices; usable() has checked that the configuration can be sharded)
    const int virtualShards = contextsForLargeSplit(dbSize);
    if (largeSplitNeedsHost(p, dbSize, p.mmgpuDeviceIndex, p.maxResListLen, p.querySeqType, p.targetSeqType, p.diagonalScoring)) {
        Debug(Debug::INFO) << "MMGPU: " << dbSize << " targets in this split - a device context indexes " << MMGPU_PF_MAX_TARGETS
                           << ", and this configuration cannot be dealt to several (sequence queries, diagonal scoring, device-built index, "
                              "--max-seqs x contexts <= 4096) - using the CPU path\n";
        ensureHostIndex(p, dbFrom, dbSize);
        return false;
    }
    if (!MMGpuRun::deviceIds().empty() || virtualShards > 1) {
        const int n = virtualShards > 1 ? virtualShards : (int)MMGpuRun::deviceIds().size();
        const bool shards = p.mmgpuDeviceIndex && MMGpuPrefilter::multiCapable(profileQuery, nuclDb, p.diagonalScoring == 0, p.maxResListLen, 2);
        int g = virtualShards > 1 ? 1 : MMGpuRun::queryGroups(n, shards);
        while (virtualShards <= 1 && n / g > 1 && !MMGpuPrefilter::multiCapable(profileQuery, nuclDb, p.diagonalScoring == 0, p.maxResListLen, n / g)) {
            do g++; while (n % g != 0);      // fewer shards per group until the merged list fits the exchange
        }
        if (!shards)
            Debug(Debug::INFO) << "MMGPU: this prefilter configuration runs without target shards (shards: sequence queries, diagonal "
                                  "scoring, device-built index, --max-seqs x shards <= 4096): every device holds the whole split\n";
        const std::vector<mmgpu_multi *> &groups = MMGpuRun::groups(g, virtualShards);
        for (size_t i = 0; i < groups.size(); i++) {
            const bool single = mmgpu_multi_size(groups[i]) == 1;
            devices.push_back(new MMGpuPrefilter(single ? mmgpu_multi_ctx(groups[i], 0) : gpu, p.kmerSubMat, p.ungappedSubMat, p.aaBiasCorrection,
                                                 p.aaBiasCorrectionScale));
            if (!single) devices.back()->useDevices(groups[i]);
        }
    } else {
        devices.push_back(new MMGpuPrefilter(gpu, p.kmerSubMat, p.ungappedSubMat, p.aaBiasCorrection, p.aaBiasCorrectionScale));
    }
    const size_t nGroups = devices.size();
    // The library's index hand-over carries the 3-mer / 2-mer score tables of sequence queries; Prefiltering only builds
    // them for amino-acid queries (Prefiltering.cpp:218-225), so a profile run computes them here the same way.
    ScoreMatrix local3, local2;
    if (profileQuery && (!p._3merSubMatrix.isValid() || !p._2merSubMatrix.isValid())) {
        const int alph = p.kmerSubMat->alphabetSize;
        p.kmerSubMat->alphabetSize = alph - 1;
        local2 = ExtendedSubstitutionMatrix::calcScoreMatrix(*p.kmerSubMat, 2);
        local3 = ExtendedSubstitutionMatrix::calcScoreMatrix(*p.kmerSubMat, 3);
        p.kmerSubMat->alphabetSize = alph;
    }
    const bool nuclSearch = Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_NUCLEOTIDES);
    ScoreMatrix &three = local3.isValid() ? local3 : p._3merSubMatrix;
    ScoreMatrix &two = local2.isValid() ? local2 : p._2merSubMatrix;
    {
        // MMGPU_DB_FILE: an unsplit run on one context keeps its device layout in a file (loaded here if getIndexTable could not
        // tell - profile queries -, else built and saved; a run that loaded it there, p.mmgpuPersisted, has no lookup to hand over)
        MMGpuPrefilter::Persisted layout;
        const bool persist = p.mmgpuDeviceIndex && !p.mmgpuPersisted && nGroups == 1 && MMGpuRun::deviceIds().empty() && virtualShards <= 1 && p.splits == 1 &&
                             Parameters::isEqualDbtype(p.targetSeqType, Parameters::DBTYPE_AMINO_ACIDS) &&
                             persistedLayout(p, p.tdbr, dbFrom, dbSize, p.mmgpuDeviceMask, p.mmgpuIndexKmerThr, p.kmerSize, p.spacedKmer, (double)p.maskProb,
                                             p.mmgpuDeviceMask ? 0 : p.maskMode, p.maskLowerCaseMode, p.maskNrepeats, p.targetSearchMode, three.isValid(),
                                             *devices[0], &layout);
        std::vector<char> handedOver(nGroups, 0);
        auto handOver = [&](size_t g) {
            MMGpuPrefilter &device = *devices[g];
            device.setMode(p.takeOnlyBestKmer, nuclSearch, p.diagonalScoring == 0);
            if (p.mmgpuPersisted) {
                device.adoptResident(dbSize);
                handedOver[g] = 1;
                return;
            }
            handedOver[g] = p.mmgpuDeviceIndex ? device.buildIndex(p.sequenceLookup, p.kmerSize, p.mmgpuIndexKmerThr, three, two, p.spacedKmer,
                                                                   p.mmgpuDeviceMask, (double)p.maskProb, g == 0, persist ? &layout : NULL)
                                               : device.loadIndex(p.indexTable, p.sequenceLookup, three, two, p.spacedKmer);
        };
        std::vector<std::thread> helpers;
        for (size_t g = 1; g < nGroups; g++) helpers.push_back(std::thread(handOver, g));
        handOver(0);
        for (size_t g = 0; g < helpers.size(); g++) helpers[g].join();
        for (size_t g = 0; g < nGroups; g++)
            if (!handedOver[g]) {
                Debug(Debug::ERROR) << "MMGPU: " << devices[g]->error() << "\n";
                EXIT(EXIT_FAILURE);
            }
    }
    watch.lap(p.mmgpuPersisted ? "targets + index already resident (persisted layout)" : p.mmgpuDeviceIndex ? "hand over targets, build the index on the device" : "hand over targets + index");
    if (local3.isValid()) ExtendedSubstitutionMatrix::freeScoreMatrix(local3);     // the library copied the tables
    if (local2.isValid()) ExtendedSubstitutionMatrix::freeScoreMatrix(local2);
    // Block size: the library's working buffers grow with the index entries a block touches (about 15 MB per query at 1 M
    // targets) and on some hosts the driver maps fresh device memory at only ~27 GB/s - blocks of 1024 queries and 4 GB candidate
    // stages cost nothing measurable in kernel time and keep a module's first device call short.  The buffers follow the query
    // residues rather than the query count: a block is MMGPU_PREF_BLOCK_QUERIES queries of 384 residues' worth, 16 times as many
    // queries at most (the ORFs of a translated search are ~45 residues long: blocks of 1024 of them spent 17 ms each on 2 ms of kernels)
    const size_t maxBlockQueries = MMGpuRun::envSize("MMGPU_PREF_BLOCK_QUERIES", 1024);
    const size_t maxBlockResidues = maxBlockQueries * 384;
    setenv("MMGPU_PF_STAGE_GB", "4", 0);
    // fused search (MMGpuFusedSearch): the entries of an unsplit run stay in memory for the alignment module of this process;
    // split runs merge their parts through files (mergePrefilterSplits / mergeTargetSplits) and are written as ever
    const bool capture = p.splits == 1 && MMGpuFusedSearch::capturing(tmpDbw.getDataFileName());
    if (capture) mmgpuFusedPrepareCapture(localThreads);
    // fused search: the unmasked lookup (device masking) is what the alignment module would map and upload again - it stays
    // resident and the lookup goes to the fused run instead of being freed with this Prefiltering object.  An overlapped run
    // (the alignment module is already waiting) gets it now, the others when this run is over.
    const bool leaveTargets = capture && p.mmgpuDeviceMask && p.mmgpuDeviceIndex && MMGpuRun::deviceIds().empty() && virtualShards <= 1 && dbFrom == 0 &&
                              dbSize == p.tdbr->getSize() && MMGpuFusedSearch::keepsTargets() && (p.sequenceLookup != NULL || p.mmgpuPersisted);
    // (a persisted layout leaves no host copy: the alignment module is told that the device holds the database's sequences, unmasked)
    if (MMGpuFusedSearch::overlappedRun()) {
        if (leaveTargets && p.mmgpuPersisted) MMGpuFusedSearch::keepResidentTargetsOnDevice(p.tdbr, gpu);
        else MMGpuFusedSearch::keepResidentTargets(leaveTargets ? p.sequenceLookup : NULL, p.tdbr, gpu);
    }

    std::vector<Sequence *> seqs(localThreads, NULL);
    std::vector<QueryMatcher *> cpuMatchers(localThreads, NULL);
    // Four blocks per query group are in flight.  A group's helper thread submits block b (descriptors, uploads, kernels enqueued)
    // and only then collects block b - 1 (downloads, hit_t lists), so the device always has the next block queued; this thread
    // maps the sequences of the blocks ahead (with their composition bias) and finishes the collected ones in order (key mapping,
    // coverage gate, DBWriter).
    struct Block {
        size_t first, nq;
        std::vector<std::vector<unsigned char> > queryNum;
        std::vector<std::vector<short> > queryProfScore;         // profile queries: copies of the Sequence's profile arrays
        std::vector<std::vector<unsigned int> > queryProfIndex;
        std::vector<std::vector<int8_t> > queryProfAln;
        std::vector<std::vector<float> > queryBias;
        std::vector<MMGpuPrefilter::Query> block;
        std::vector<std::vector<hit_t> > results;
        std::vector<bool> needsCpu;
        std::vector<mmgpu_pf_qstat> qstats;
        bool ok;
    };
    const size_t ringSlots = 4 * nGroups;
    std::vector<Block> ring(ringSlots);
    std::vector<double> deviceSeconds(nGroups, 0.0);
    double kmersPerPos = 0;
    size_t dbMatches = 0, doubleMatches = 0, querySeqLenSum = 0, resSize = 0, diagonalOverflow = 0;

    std::vector<size_t> starts;      // first query of every block, and the end of the last
    {
        size_t residues = 0, count = 0;
        for (size_t next = queryFrom; next < queryFrom + querySize; next++) {
            const size_t len = p.qdbr->getSeqLen(next);
            const bool full = count >= 16 * maxBlockQueries || (count >= maxBlockQueries && residues + len > maxBlockResidues);
            if (count == 0 || full) {
                starts.push_back(next);
                residues = 0;
                count = 0;
            }
            residues += len;
            count++;
        }
        starts.push_back(queryFrom + querySize);
    }

    auto mapBlock = [&](Block &B, size_t next, size_t nq) {
        const double t0 = watch.now();
        B.first = next;
        B.nq = nq;
        B.queryNum.assign(nq, std::vector<unsigned char>());
        B.queryProfScore.assign(nq, std::vector<short>());
        B.queryProfIndex.assign(nq, std::vector<unsigned int>());
        B.queryProfAln.assign(nq, std::vector<int8_t>());
        B.queryBias.assign(nq, std::vector<float>());
        B.block.assign(nq, MMGpuPrefilter::Query());
        std::vector<std::vector<unsigned char> > &queryNum = B.queryNum;
        std::vector<std::vector<short> > &queryProfScore = B.queryProfScore;
        std::vector<std::vector<unsigned int> > &queryProfIndex = B.queryProfIndex;
        std::vector<std::vector<int8_t> > &queryProfAln = B.queryProfAln;
        std::vector<MMGpuPrefilter::Query> &block = B.block;
#pragma omp parallel num_threads(localThreads)
        {
            unsigned int thread_idx = 0;
#ifdef OPENMP
            thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
            if (seqs[thread_idx] == NULL)
                seqs[thread_idx] = new Sequence(p.qdbr->getMaxSeqLen(), p.querySeqType, p.kmerSubMat, p.kmerSize, p.spacedKmer,
                                                p.aaBiasCorrection, true, p.spacedKmerPattern);
            Sequence &seq = *seqs[thread_idx];
#pragma omp for schedule(dynamic, 16)
            for (size_t b = 0; b < nq; b++) {
                const size_t id = next + b;
                char *seqData = p.qdbr->getData(id, thread_idx);
                DBKeyType qKey = p.qdbr->getDbKey(id);
                seq.mapSequence(id, qKey, seqData, p.qdbr->getSeqLen(id));
                queryNum[b].assign(seq.numSequence, seq.numSequence + seq.L);
                block[b].numSequence = queryNum[b].data();
                block[b].L = seq.L;
                if (profileQuery) {
                    const size_t row = seq.profile_row_size;
                    queryProfScore[b].assign(seq.profile_score, seq.profile_score + (size_t)seq.L * row);
                    queryProfIndex[b].assign(seq.profile_index, seq.profile_index + (size_t)seq.L * row);
                    const int8_t *ap = seq.getAlignmentProfile();
                    queryProfAln[b].assign(ap, ap + Sequence::PROFILE_AA_SIZE * (size_t)seq.L);
                    block[b].profileScore = queryProfScore[b].data();
                    block[b].profileIndex = queryProfIndex[b].data();
                    block[b].profileRow = (unsigned int)row;
                    block[b].profile = queryProfAln[b].data();
                }
                // :855-868
                DBLocalId targetSeqId = DB_LOCAL_ID_INVALID;
                if (p.sameQTDB || p.includeIdentical) {
                    size_t foundTargetSeqId = p.tdbr->getId(seq.getDbKey());
                    if (foundTargetSeqId >= dbFrom && foundTargetSeqId < (dbFrom + dbSize) && foundTargetSeqId != DB_ENTRY_NOT_FOUND) {
                        targetSeqId = static_cast<DBLocalId>(foundTargetSeqId - dbFrom);
                        if (targetSeqId > p.tdbr->getSize()) {
                            Debug(Debug::ERROR) << "targetSeqId: " << targetSeqId << " > target database size: " << p.tdbr->getSize() << "\n";
                            EXIT(EXIT_FAILURE);
                        }
                    }
                }
                block[b].identityId = targetSeqId == DB_LOCAL_ID_INVALID ? UINT_MAX : (unsigned int)targetSeqId;
                if (!profileQuery) {
                    devices[0]->compositionBias(block[b], B.queryBias[b]);
                    block[b].compBias = B.queryBias[b].data();
                }
            }
        }
        watch.add(0, watch.now() - t0);
    };

    auto writeBlock = [&](Block &B) {
        const double t0 = watch.now();
        const size_t nq = B.nq, next = B.first;
        std::vector<MMGpuPrefilter::Query> &block = B.block;
        std::vector<std::vector<hit_t> > &results = B.results;
        std::vector<bool> &needsCpu = B.needsCpu;
        std::vector<mmgpu_pf_qstat> &qstats = B.qstats;
        for (size_t b = 0; b < nq; b++)
            if (needsCpu[b]) {      // the reference's matcher needs the reference's index
                ensureHostIndex(p, dbFrom, dbSize);
                break;
            }
#pragma omp parallel num_threads(localThreads)
        {
            unsigned int thread_idx = 0;
#ifdef OPENMP
            thread_idx = static_cast<unsigned int>(omp_get_thread_num());
#endif
            Sequence &seq = *seqs[thread_idx];
            char buffer[128];
            std::string result;
            result.reserve(1000000);
            std::vector<unsigned int> resultKeys;      // (fused search: the target keys of the entry's lines)
#pragma omp for schedule(dynamic, 16) reduction(+ : kmersPerPos, resSize, dbMatches, doubleMatches, querySeqLenSum, diagonalOverflow)
            for (size_t b = 0; b < nq; b++) {
                progress.updateProgress();
                const size_t id = next + b;
                const DBKeyType qKey = p.qdbr->getDbKey(id);
                hit_t *hits = results[b].data();
                size_t resultSize = results[b].size();
                statistics_t cpuStats;
                bool haveCpuStats = false;
                if (needsCpu[b]) {
                    // the reference's own matcher for the queries the device declined (QueryMatcher.cpp:310-346 with > 62 flushes)
                    if (cpuMatchers[thread_idx] == NULL) {
                        cpuMatchers[thread_idx] = new QueryMatcher(p.indexTable, p.sequenceLookup, p.kmerSubMat, p.ungappedSubMat, p.kmerThr,
                                                                   p.kmerSize, dbSize, std::max(p.tdbr->getMaxSeqLen(), p.qdbr->getMaxSeqLen()),
                                                                   p.maxResListLen, p.aaBiasCorrection, p.aaBiasCorrectionScale, p.diagonalScoring,
                                                                   p.minDiagScoreThr, p.takeOnlyBestKmer, nuclSearch, p.ungappedSubMatAux, p.targetSeqType);
                        if (seq.profile_matrix != NULL) cpuMatchers[thread_idx]->setProfileMatrix(seq.profile_matrix);     // :832-836
                        else cpuMatchers[thread_idx]->setSubstitutionMatrix(&p._3merSubMatrix, &p._2merSubMatrix);
                    }
                    seq.mapSequence(id, qKey, p.qdbr->getData(id, thread_idx), p.qdbr->getSeqLen(id));
                    const DBLocalId identityId = block[b].identityId == UINT_MAX ? DB_LOCAL_ID_INVALID : (DBLocalId)block[b].identityId;
                    std::pair<hit_t *, size_t> r = cpuMatchers[thread_idx]->matchQuery(&seq, identityId, nuclSearch);
                    hits = r.first;
                    resultSize = r.second;
                    cpuStats = *cpuMatchers[thread_idx]->getStatistics();
                    haveCpuStats = true;
                }
                const float queryLength = static_cast<float>(p.qdbr->getSeqLen(id));
                for (size_t i = 0; i < resultSize; i++) {
                    hit_t *res = hits + i;
                    // correct the 0 indexed sequence id again to its real identifier
                    size_t targetSeqId1 = res->seqId + dbFrom;
                    // replace id with key
                    res->seqId = p.tdbr->getDbKey(targetSeqId1);
                    if (UNLIKELY(targetSeqId1 >= p.tdbr->getSize())) {
                        Debug(Debug::WARNING) << "Wrong prefiltering result for query: " << p.qdbr->getDbKey(id) << " -> " << targetSeqId1 << "\t" << res->prefScore << "\n";
                    }
                    if (p.covThr > 0.0 && (p.covMode == Parameters::COV_MODE_BIDIRECTIONAL || p.covMode == Parameters::COV_MODE_QUERY ||
                                           p.covMode == Parameters::COV_MODE_LENGTH_SHORTER)) {
                        const float targetLength = static_cast<float>(p.tdbr->getSeqLen(targetSeqId1));
                        if (Util::canBeCovered(p.covThr, p.covMode, queryLength, targetLength) == false) {
                            continue;
                        }
                    }
                    int len = QueryMatcher::prefilterHitToBuffer(buffer, *res);
                    result.append(buffer, len);
                    if (capture) resultKeys.push_back(res->seqId);
                }
                if (capture) {
                    MMGpuFusedSearch::capture(qKey, result.c_str(), result.length(), thread_idx, resultKeys.data(), resultKeys.size());
                    resultKeys.clear();
                }
                else tmpDbw.writeData(result.c_str(), result.length(), qKey, thread_idx);
                result.clear();
                if (resultSize != 0) {
                    notEmpty[id - queryFrom] = 1;
                }
                if (Debug::debugLevel >= Debug::INFO) {
                    if (haveCpuStats) {
                        kmersPerPos += cpuStats.kmersPerPos;
                        dbMatches += cpuStats.dbMatches;
                        doubleMatches += cpuStats.doubleMatches;
                        diagonalOverflow += cpuStats.diagonalOverflow;
                    } else {
                        // QueryMatcher::match's counters (QueryMatcher.cpp:366-374)
                        kmersPerPos += (double)qstats[b].kmer_list_len / (double)block[b].L;
                        dbMatches += qstats[b].db_matches;
                        // statistics_t::doubleMatches is only counted with --diag-score 0 (QueryMatcher.cpp:366-371)
                        if (p.diagonalScoring == 0) doubleMatches += qstats[b].double_hits;
                    }
                    querySeqLenSum += block[b].L;
                    resSize += resultSize;
                    reslens[thread_idx]->emplace_back(resultSize);
                }
            }
        }
        watch.add(2, watch.now() - t0);
    };

    const size_t nBlocks = starts.size() - 1;
    std::mutex lock;
    std::condition_variable changed;
    size_t mapped = 0;                        // blocks [0, mapped) are ready for the device
    std::vector<char> collected(nBlocks, 0);  // the device's answer for block k is in its ring slot
    auto groupWorker = [&](size_t g) {        // blocks g, g + G, g + 2 G, ...
        MMGpuPrefilter::Pending *before = NULL;
        size_t beforeK = 0;
        auto collect = [&]() {
            Block &B = ring[beforeK % ringSlots];
            B.ok = devices[g]->finishBlock(before, B.results, B.needsCpu, &B.qstats);
            {
                std::lock_guard<std::mutex> guard(lock);
                collected[beforeK] = 1;
            }
            changed.notify_all();
            before = NULL;
        };
        for (size_t k = g; k < nBlocks; k += nGroups) {
            {
                std::unique_lock<std::mutex> guard(lock);
                changed.wait(guard, [&]() { return mapped > k; });
            }
            const double t0 = watch.now();
            MMGpuPrefilter::Pending *cur = devices[g]->submitBlock(ring[k % ringSlots].block, p.kmerThr, p.maxResListLen, p.minDiagScoreThr);
            if (before != NULL) collect();
            before = cur;
            beforeK = k;
            deviceSeconds[g] += watch.now() - t0;
        }
        if (before != NULL) {
            const double t0 = watch.now();
            collect();
            deviceSeconds[g] += watch.now() - t0;
        }
    };
    std::vector<std::thread> workers;
    for (size_t g = 0; g < nGroups && g < nBlocks; g++) workers.push_back(std::thread(groupWorker, g));
    for (size_t nextMap = 0, nextWrite = 0; nextWrite < nBlocks;) {
        if (nextMap < nBlocks && nextMap - nextWrite < ringSlots) {
            mapBlock(ring[nextMap % ringSlots], starts[nextMap], starts[nextMap + 1] - starts[nextMap]);
            {
                std::lock_guard<std::mutex> guard(lock);
                mapped = ++nextMap;
            }
            changed.notify_all();
            continue;
        }
        {
            std::unique_lock<std::mutex> guard(lock);
            changed.wait(guard, [&]() { return collected[nextWrite] != 0; });
        }
        Block &B = ring[nextWrite % ringSlots];
        if (!B.ok) {
            Debug(Debug::ERROR) << "MMGPU: " << devices[nextWrite % nGroups]->error() << "\n";
            EXIT(EXIT_FAILURE);
        }
        writeBlock(B);
        if (capture) MMGpuFusedSearch::publish();
        nextWrite++;
    }
    for (size_t g = 0; g < workers.size(); g++) workers[g].join();
    watch.add(1, *std::max_element(deviceSeconds.begin(), deviceSeconds.end()));
    {
        static const char *const names[3] = {"map queries", "device blocks (prepare, run, fetch; the busiest query group)", "serialise + write"};
        watch.report(names, 3);
    }
    {
        size_t handedBack[8] = {0, 0, 0, 0, 0, 0, 0, 0}, back = 0, rerun = 0;
        for (size_t g = 0; g < nGroups; g++) {
            rerun += devices[g]->rerunUnsplit;
            for (size_t i = 0; i < 8; i++) {
                handedBack[i] += devices[g]->handedBack[i];
                back += devices[g]->handedBack[i];
            }
        }
        if (rerun != 0)
            Debug(Debug::INFO) << "MMGPU: " << rerun << " of " << querySize << " queries ran once more against the unsplit database on the device "
                               << "(a shard reached its share of the database-hit buffer)\n";
        if (back != 0)
            Debug(Debug::INFO) << "MMGPU: " << back << " of " << querySize << " queries ran through the host's matcher (database-hit buffer flushes beyond the device's: "
                               << handedBack[MMGPU_PF_OVERFLOW] << ", sequences of 32768 residues or more: " << handedBack[MMGPU_PF_LONG_SEQ]
                               << ", candidate array / saturated-diagonal ties: " << handedBack[MMGPU_PF_SAT_TIE]
                               << ", shard-dependent order: " << handedBack[MMGPU_PF_SHARD_INEXACT] << ")\n";
    }
    for (size_t i = 0; i < localThreads; i++) {
        delete seqs[i];
        delete cpuMatchers[i];
    }
    if (leaveTargets) {
        // A query handed back to the CPU made ensureHostIndex replace the unmasked lookup by a host-masked copy (it clears
        // mmgpuDeviceIndex): that copy is not what the device aligned against, so it is not handed over - the alignment module maps
        // and uploads its own targets.  (An overlapped run handed the unmasked lookup over before the loop; ensureHostIndex leaves
        // a lookup the fused search holds alone.)
        if (!MMGpuFusedSearch::overlappedRun() && p.mmgpuDeviceIndex) {
            if (p.mmgpuPersisted) MMGpuFusedSearch::keepResidentTargetsOnDevice(p.tdbr, gpu);
            else MMGpuFusedSearch::keepResidentTargets(p.sequenceLookup, p.tdbr, gpu);
        }
        if (MMGpuFusedSearch::holdsLookup(p.sequenceLookup)) p.sequenceLookup = NULL;
    }
    for (size_t g = 0; g < nGroups; g++) delete devices[g];
    st.kmersPerPos = kmersPerPos;
    st.dbMatches = dbMatches;
    st.doubleMatches = doubleMatches;
    st.querySeqLenSum = querySeqLenSum;
    st.resSize = resSize;
    st.diagonalOverflow = diagonalOverflow;
    return true;
}
