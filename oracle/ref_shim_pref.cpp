// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
//
// Prefilter half of oracle/_ref/libmmref.so: drives the REAL reference classes
//   ExtendedSubstitutionMatrix::calcScoreMatrix   src/prefiltering/ExtendedSubstitutionMatrix.cpp:20-71
//   IndexTable::addKmerCount/addSequence/...      src/prefiltering/IndexTable.h:135-191,350-403
//   SequenceLookup::addSequence                   src/prefiltering/SequenceLookup.cpp
//   KmerGenerator::generateKmerList               src/prefiltering/KmerGenerator.cpp:108-184
//   QueryMatcher::matchQuery                      src/prefiltering/QueryMatcher.cpp:103-241
// in the order Prefiltering's constructor / getIndexTable / runSplit use them
// (Prefiltering.cpp:68-69, 220-225, 544-583, 826-842, 873), from numeric sequences, without DBReader and
// without tantan masking (IndexBuilder.cpp:146-156; `--mask 0`), so that the plain-C restatement in
// oracle/prefilter_oracle.c can be fuzzed against it.
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "Debug.h"
#include "ExtendedSubstitutionMatrix.h"
#include "tantan.h"
#include "IndexTable.h"
#include "Indexer.h"
#include "KmerGenerator.h"
#include "Parameters.h"
#include "QueryMatcher.h"
#include "Sequence.h"
#include "SequenceLookup.h"
#include "NucleotideMatrix.h"
#include "SubstitutionMatrix.h"

namespace {

// QueryMatcher keeps its bin count (chosen from the host's L2 size, QueryMatcher.cpp:460-488) protected.
struct MatcherProbe : public QueryMatcher {
    using QueryMatcher::QueryMatcher;
    unsigned int bins() const { return activeCounter; }
    // Emulate a host with a different L2 size: replace the CacheFriendlyOperations<BINS> instance that
    // initDiagonalMatcher (QueryMatcher.cpp:460-488) picked by the one another host would have picked.
    void forceBins(unsigned int b) {
        deleteDiagonalMatcher(activeCounter);
#define MMREF_FORCE(x) case x: cachedOperation##x = new CacheFriendlyOperations<x>(dbSize, maxDbMatches / x); activeCounter = x; break;
        switch (b) {
            MMREF_FORCE(2) MMREF_FORCE(4) MMREF_FORCE(8) MMREF_FORCE(16) MMREF_FORCE(32) MMREF_FORCE(64)
            MMREF_FORCE(128) MMREF_FORCE(256) MMREF_FORCE(512) MMREF_FORCE(1024) MMREF_FORCE(2048)
            default: cachedOperation2 = new CacheFriendlyOperations<2>(dbSize, maxDbMatches / 2); activeCounter = 2;
        }
#undef MMREF_FORCE
    }
};

struct PrefCtx {
    SubstitutionMatrix *kmerMat;      // VTML80, bit factor 8, bias -0.2   (Prefiltering.cpp:68)
    SubstitutionMatrix *ungappedMat;  // blosum62, bit factor 2, bias -0.2 (Prefiltering.cpp:69)
    ScoreMatrix two, three;
    IndexTable *index;
    SequenceLookup *lookup;
    MatcherProbe *matcher;
    KmerGenerator *gen;
    Sequence *qseq;
    int kmerSize;
    size_t dbSize;
    unsigned maxLen;
};

}  // namespace

extern "C" {

void *mmref_pref_new(const char *kmer_matrix, const char *ungapped_matrix, int kmer_size) {
    Debug::setDebugLevel(Debug::ERROR);
    PrefCtx *c = new PrefCtx();
    c->kmerMat = new SubstitutionMatrix(kmer_matrix, 8.0f, -0.2f);
    c->ungappedMat = new SubstitutionMatrix(ungapped_matrix, 2.0f, -0.2f);
    c->kmerSize = kmer_size;
    const int alph = c->kmerMat->alphabetSize;
    c->kmerMat->alphabetSize = alph - 1;   // Prefiltering.cpp:221-224: X is not part of the k-mer alphabet
    c->two = ExtendedSubstitutionMatrix::calcScoreMatrix(*c->kmerMat, 2);
    c->three = ExtendedSubstitutionMatrix::calcScoreMatrix(*c->kmerMat, 3);
    c->kmerMat->alphabetSize = alph;
    c->index = NULL;
    c->lookup = NULL;
    c->matcher = NULL;
    c->gen = NULL;
    c->qseq = NULL;
    return c;
}

int mmref_pref_alphabet(void *h) { return ((PrefCtx *)h)->kmerMat->alphabetSize; }

// tantan masking exactly as IndexBuilder::fillDatabase applies it to amino-acid targets (IndexBuilder.cpp:148, Masker.cpp:14-57
// with maskTantan only): tantan::maskSequences over Sequence::numSequence with the k-mer matrix's likelihood ratios
// (ProbabilityMatrix, BaseMatrix.h:83-101), then finalizeMasking.  seqs are masked in place; lr_out (may be NULL) receives the
// alphabet x alphabet likelihood-ratio table, probs_out (may be NULL) the per-letter repeat probabilities of sequence 0.
uint64_t mmref_tantan_mask(void *h, uint8_t *tdata, const uint64_t *toff, uint32_t n, double mask_prob, double *lr_out, float *probs_out) {
    PrefCtx *c = (PrefCtx *)h;
    ProbabilityMatrix pm(*c->kmerMat);
    const int a = c->kmerMat->alphabetSize;
    if (lr_out)
        for (int i = 0; i < a; i++)
            for (int j = 0; j < a; j++) lr_out[i * a + j] = pm.probMatrixPointers[i][j];
    if (probs_out && n > 0)
        tantan::getProbabilities(tdata + toff[0], tdata + toff[1], 50, pm.probMatrixPointers, 0.005, 0.05, 0.9, 0, 0, probs_out);
    uint64_t masked = 0;
    const unsigned char maskLetter = c->kmerMat->aa2num[(int)'X'];
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : masked)
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *s = tdata + toff[i];
        const size_t len = toff[i + 1] - toff[i];
        masked += tantan::maskSequences(s, s + len, 50, pm.probMatrixPointers, 0.005, 0.05, 0.9, 0, 0, mask_prob, pm.hardMaskTable);
        const unsigned char maskChar = pm.hardMaskTable[0];      // Masker::finalizeMasking (Masker.cpp:119-126)
        for (size_t k = 0; k < len; k++) s[k] = (s[k] == maskChar || s[k] == maskLetter) ? maskLetter : s[k];
    }
    return masked;
}

void mmref_pref_get_matrices(void *h, int8_t *kmer_mat, int8_t *ungapped_mat, int16_t *kmer_mat16, double *kmer_pback) {
    PrefCtx *c = (PrefCtx *)h;
    int a = c->kmerMat->alphabetSize;
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) {
            kmer_mat[i * a + j] = (int8_t)c->kmerMat->subMatrix[i][j];
            kmer_mat16[i * a + j] = c->kmerMat->subMatrix[i][j];
            ungapped_mat[i * a + j] = (int8_t)c->ungappedMat->subMatrix[i][j];
        }
    for (int i = 0; i < a; i++) kmer_pback[i] = c->kmerMat->pBack[i];
}

// which: 2 or 3; returns elementSize, *row_size; copies when out pointers are non-NULL (elementSize*elementSize each,
// i.e. without the SIMD padding columns)
uint64_t mmref_pref_score_matrix(void *h, int which, uint64_t *row_size, int16_t *score, uint32_t *index) {
    PrefCtx *c = (PrefCtx *)h;
    const ScoreMatrix &m = which == 2 ? c->two : c->three;
    if (row_size) *row_size = m.rowSize;
    if (score && index)
        for (size_t r = 0; r < m.elementSize; r++)
            for (size_t z = 0; z < m.elementSize; z++) {
                score[r * m.elementSize + z] = m.score[r * m.rowSize + z];
                index[r * m.elementSize + z] = m.index[r * m.rowSize + z];
            }
    return m.elementSize;
}

// Build index + sequence lookup over numeric targets exactly as IndexBuilder::fillDatabase does for amino-acid
// targets with masking off (IndexBuilder.cpp:118-166, 226-270).
void mmref_pref_build_index(void *h, const uint8_t *tdata, const uint64_t *toff, uint32_t n, int kmer_thr, int spaced) {
    PrefCtx *c = (PrefCtx *)h;
    const int alph = c->kmerMat->alphabetSize;
    unsigned maxLen = 1;
    for (uint32_t i = 0; i < n; i++) maxLen = std::max<unsigned>(maxLen, (unsigned)(toff[i + 1] - toff[i]));
    c->maxLen = maxLen + 1;
    c->dbSize = n;
    delete c->index;
    delete c->lookup;
    c->index = new IndexTable(alph - 1, c->kmerSize, false);          // Prefiltering.cpp:560-563
    c->lookup = new SequenceLookup(n, toff[n]);
    char *idScore = new char[alph];
    for (int a = 0; a < alph; a++) idScore[a] = (char)c->kmerMat->subMatrix[a][a];   // IndexBuilder.cpp:11-22
    size_t tableSize = 0;
    // same structure as IndexBuilder::fillDatabase: thread-private Sequence/Indexer/buffers, atomic adds inside
    // IndexTable (IndexBuilder.cpp:118-166)
#pragma omp parallel reduction(+ : tableSize)
    {
        Sequence s(c->maxLen, Parameters::DBTYPE_AMINO_ACIDS, c->kmerMat, c->kmerSize, spaced != 0, false, true);
        Indexer idxer(alph - 1, c->kmerSize);
        std::vector<unsigned int> buffer(c->maxLen + 8);
#pragma omp for schedule(dynamic, 100)
        for (uint32_t i = 0; i < n; i++) {
            unsigned len = (unsigned)(toff[i + 1] - toff[i]);
            s.mapSequence(i, i, std::make_pair((const unsigned char *)(tdata + toff[i]), (const unsigned int)len));
            c->index->addKmerCount(&s, &idxer, buffer.data(), kmer_thr, idScore);
            c->lookup->addSequence(s.numSequence, s.L, i, toff[i]);
            tableSize += len;
        }
    }
    c->index->initMemory(tableSize);
    c->index->init();
#pragma omp parallel
    {
        Sequence s(c->maxLen, Parameters::DBTYPE_AMINO_ACIDS, c->kmerMat, c->kmerSize, spaced != 0, false, true);
        Indexer idxer(alph - 1, c->kmerSize);
        IndexEntryLocalTmp *tmp = (IndexEntryLocalTmp *)malloc((c->maxLen + 8) * sizeof(IndexEntryLocalTmp));
#pragma omp for schedule(dynamic, 100)
        for (uint32_t i = 0; i < n; i++) {
            s.mapSequence(i, i, c->lookup->getSequence(i));
            c->index->addSequence(&s, &idxer, &tmp, c->maxLen + 8, kmer_thr, idScore);
        }
        free(tmp);
    }
    delete[] idScore;
    c->index->revertPointer();
    c->index->sortDBSeqLists();
}

uint64_t mmref_pref_index_entries(void *h) { return ((PrefCtx *)h)->index->getTableEntriesNum(); }
uint64_t mmref_pref_index_table_size(void *h) { return ((PrefCtx *)h)->index->getTableSize(); }

void mmref_pref_index_dump(void *h, uint64_t *offsets /*tableSize+1*/, uint32_t *seq_id, uint16_t *pos_j) {
    PrefCtx *c = (PrefCtx *)h;
    size_t ts = c->index->getTableSize();
    for (size_t k = 0; k <= ts; k++) offsets[k] = c->index->getOffsets()[k];
    IndexEntryLocal *e = c->index->getEntries();
    for (uint64_t k = 0; k < c->index->getTableEntriesNum(); k++) {
        seq_id[k] = e[k].seqId;
        pos_j[k] = e[k].position_j;
    }
}

// similar k-mer list for one k-mer (already extracted residues) at threshold thr; returns count
uint64_t mmref_pref_kmer_list(void *h, const uint8_t *kmer, int thr, uint64_t *out, uint64_t cap) {
    PrefCtx *c = (PrefCtx *)h;
    if (!c->gen) {
        c->gen = new KmerGenerator(c->kmerSize, c->kmerMat->alphabetSize - 1, (short)thr);
        c->gen->setDivideStrategy(&c->three, &c->two);
    }
    c->gen->setThreshold((short)thr);
    std::pair<size_t *, size_t> r = c->gen->generateKmerList(kmer);
    for (size_t i = 0; i < r.second && i < cap; i++) out[i] = r.first[i];
    return r.second;
}

// create the matcher as Prefiltering::runSplit does (Prefiltering.cpp:826-842)
unsigned mmref_pref_make_matcher(void *h, int kmer_thr, unsigned max_seq_len, uint64_t max_hits, int comp_bias,
                                 float comp_bias_scale, int diag_scoring, unsigned min_diag_score, int spaced,
                                 unsigned force_bins) {
    PrefCtx *c = (PrefCtx *)h;
    delete c->matcher;
    delete c->qseq;
    unsigned ml = std::max(max_seq_len, c->maxLen);
    c->matcher = new MatcherProbe(c->index, c->lookup, c->kmerMat, c->ungappedMat, (short)kmer_thr, c->kmerSize,
                                  c->dbSize, ml, max_hits, comp_bias != 0, comp_bias_scale, diag_scoring != 0,
                                  min_diag_score, false, false);
    if (force_bins) c->matcher->forceBins(force_bins);
    c->matcher->setSubstitutionMatrix(&c->three, &c->two);
    c->qseq = new Sequence(ml, Parameters::DBTYPE_AMINO_ACIDS, c->kmerMat, c->kmerSize, spaced != 0, comp_bias != 0, true);
    return c->matcher->bins();
}

// matchQuery: returns number of hits; identity_id = UINT32_MAX for none
uint64_t mmref_pref_match(void *h, const uint8_t *q, uint32_t qlen, uint32_t identity_id, uint32_t *ids, int32_t *scores,
                          uint16_t *diags, uint64_t cap, uint64_t *db_matches, double *kmers_per_pos) {
    PrefCtx *c = (PrefCtx *)h;
    c->qseq->mapSequence(0, 0, std::make_pair((const unsigned char *)q, (const unsigned int)qlen));
    DBLocalId ident = identity_id == UINT32_MAX ? DB_LOCAL_ID_INVALID : (DBLocalId)identity_id;
    std::pair<hit_t *, size_t> r = c->matcher->matchQuery(c->qseq, ident, false);
    for (size_t i = 0; i < r.second && i < cap; i++) {
        ids[i] = (uint32_t)r.first[i].seqId;
        scores[i] = r.first[i].prefScore;
        diags[i] = r.first[i].diagonal;
    }
    if (db_matches) *db_matches = c->matcher->getStatistics()->dbMatches;
    if (kmers_per_pos) *kmers_per_pos = c->matcher->getStatistics()->kmersPerPos;
    return r.second;
}

// Profile query (DBTYPE_HMM_PROFILE): `entry` = one profile-database entry (25 bytes per position, Sequence.cpp:301-325).
// Prefiltering::runSplit maps it with a Sequence of that type built WITH a k-mer size (mapProfile then sorts the rows,
// :343-351) and hands Sequence::profile_matrix to the matcher (Prefiltering.cpp:832-834); matchQuery as above.
// Outputs for the caller (any may be NULL): what the reference derived from the entry - the sorted score rows and their
// letters [qlen][row_size], the alignment profile [20][qlen], the query letters [qlen].
uint64_t mmref_pref_match_profile(void *h, const char *entry, uint32_t qlen, int kmer_thr, unsigned max_seq_len, uint64_t max_hits,
                                  unsigned min_diag_score, int spaced, unsigned force_bins, uint32_t identity_id,
                                  uint32_t *ids, int32_t *scores, uint16_t *diags, uint64_t cap, uint64_t *db_matches,
                                  int16_t *pscore, uint32_t *pindex, uint32_t *row_size, int8_t *aln, uint8_t *letters) {
    PrefCtx *c = (PrefCtx *)h;
    unsigned ml = std::max(max_seq_len, c->maxLen);
    MatcherProbe matcher(c->index, c->lookup, c->kmerMat, c->ungappedMat, (short)kmer_thr, c->kmerSize, c->dbSize, ml, max_hits,
                         true, 1.0f, true, min_diag_score, false, false);
    if (force_bins) matcher.forceBins(force_bins);
    Sequence seq(ml, Parameters::DBTYPE_HMM_PROFILE, c->kmerMat, c->kmerSize, spaced != 0, true, true);
    matcher.setProfileMatrix(seq.profile_matrix);
    seq.mapSequence(0, 0, entry, qlen);
    if (row_size) *row_size = (uint32_t)seq.profile_row_size;
    if (pscore) memcpy(pscore, seq.profile_score, (size_t)seq.L * seq.profile_row_size * sizeof(short));
    if (pindex) memcpy(pindex, seq.profile_index, (size_t)seq.L * seq.profile_row_size * sizeof(unsigned int));
    if (aln) memcpy(aln, seq.getAlignmentProfile(), Sequence::PROFILE_AA_SIZE * (size_t)seq.L);
    if (letters) memcpy(letters, seq.numSequence, (size_t)seq.L);
    DBLocalId ident = identity_id == UINT32_MAX ? DB_LOCAL_ID_INVALID : (DBLocalId)identity_id;
    std::pair<hit_t *, size_t> r = matcher.matchQuery(&seq, ident, false);
    for (size_t i = 0; i < r.second && i < cap; i++) {
        ids[i] = (uint32_t)r.first[i].seqId;
        scores[i] = r.first[i].prefScore;
        diags[i] = r.first[i].diagonal;
    }
    if (db_matches) *db_matches = matcher.getStatistics()->dbMatches;
    return r.second;
}

// ---- nucleotide prefilter (Search.cpp:180-198: exact k-mers, k = 15; Prefiltering.cpp:62-66: one NucleotideMatrix for seeding and
// ungapped scoring, k-mer threshold 0, :555-563 index over the 4-letter alphabet; QueryMatcher::matchQuery's isNucleotide
// branch :147-177) - the same classes, driven for nucleotide sequences given as numeric codes (A C T G N = 0..4).
struct NPrefCtx {
    NucleotideMatrix *mat;
    IndexTable *index;
    SequenceLookup *lookup;
    int kmerSize;
    size_t dbSize;
    unsigned maxLen;
};

void *mmref_npref_new(const char *nucl_matrix, int kmer_size) {
    Debug::setDebugLevel(Debug::ERROR);
    NPrefCtx *c = new NPrefCtx();
    c->mat = new NucleotideMatrix(nucl_matrix, 1.0f, 0.0f);
    c->kmerSize = kmer_size;
    c->index = NULL;
    c->lookup = NULL;
    return c;
}

void mmref_npref_matrix(void *h, int8_t *out /* alphabet^2 */) {
    NPrefCtx *c = (NPrefCtx *)h;
    const int a = c->mat->alphabetSize;
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) out[i * a + j] = (int8_t)c->mat->subMatrix[i][j];
}

void mmref_npref_build_index(void *h, const uint8_t *tdata, const uint64_t *toff, uint32_t n, int spaced) {
    NPrefCtx *c = (NPrefCtx *)h;
    const int alph = c->mat->alphabetSize;
    unsigned maxLen = 1;
    for (uint32_t i = 0; i < n; i++) maxLen = std::max<unsigned>(maxLen, (unsigned)(toff[i + 1] - toff[i]));
    c->maxLen = maxLen + 1;
    c->dbSize = n;
    delete c->index;
    delete c->lookup;
    c->index = new IndexTable(alph - 1, c->kmerSize, false);
    c->lookup = new SequenceLookup(n, toff[n]);
    char *idScore = new char[alph];
    for (int a = 0; a < alph; a++) idScore[a] = (char)c->mat->subMatrix[a][a];
    size_t tableSize = 0;
    {
        Sequence s(c->maxLen, Parameters::DBTYPE_NUCLEOTIDES, c->mat, c->kmerSize, spaced != 0, false, true);
        Indexer idxer(alph - 1, c->kmerSize);
        std::vector<unsigned int> buffer(c->maxLen + 8);
        for (uint32_t i = 0; i < n; i++) {
            unsigned len = (unsigned)(toff[i + 1] - toff[i]);
            s.mapSequence(i, i, std::make_pair((const unsigned char *)(tdata + toff[i]), (const unsigned int)len));
            c->index->addKmerCount(&s, &idxer, buffer.data(), 0, idScore);
            c->lookup->addSequence(s.numSequence, s.L, i, toff[i]);
            tableSize += len;
        }
    }
    c->index->initMemory(tableSize);
    c->index->init();
    {
        Sequence s(c->maxLen, Parameters::DBTYPE_NUCLEOTIDES, c->mat, c->kmerSize, spaced != 0, false, true);
        Indexer idxer(alph - 1, c->kmerSize);
        IndexEntryLocalTmp *tmp = (IndexEntryLocalTmp *)malloc((c->maxLen + 8) * sizeof(IndexEntryLocalTmp));
        for (uint32_t i = 0; i < n; i++) {
            s.mapSequence(i, i, c->lookup->getSequence(i));
            c->index->addSequence(&s, &idxer, &tmp, c->maxLen + 8, 0, idScore);
        }
        free(tmp);
    }
    delete[] idScore;
    c->index->revertPointer();
    c->index->sortDBSeqLists();
}

uint64_t mmref_npref_index_entries(void *h) { return ((NPrefCtx *)h)->index->getTableEntriesNum(); }
void mmref_npref_index_dump(void *h, uint64_t *offsets, uint32_t *seq_id, uint16_t *pos_j) {
    NPrefCtx *c = (NPrefCtx *)h;
    size_t ts = c->index->getTableSize();
    for (size_t k = 0; k <= ts; k++) offsets[k] = c->index->getOffsets()[k];
    IndexEntryLocal *e = c->index->getEntries();
    for (uint64_t k = 0; k < c->index->getTableEntriesNum(); k++) {
        seq_id[k] = e[k].seqId;
        pos_j[k] = e[k].position_j;
    }
}

uint64_t mmref_npref_match(void *h, const uint8_t *q, uint32_t qlen, unsigned max_seq_len, uint64_t max_hits, unsigned min_diag_score,
                           int spaced, unsigned force_bins, uint32_t identity_id, uint32_t *ids, int32_t *scores, uint16_t *diags,
                           uint64_t cap, uint64_t *db_matches) {
    NPrefCtx *c = (NPrefCtx *)h;
    unsigned ml = std::max(max_seq_len, c->maxLen);
    // Prefiltering.cpp:826-831: kmerThr 0, composition bias as configured (zero for nucleotides inside matchQuery),
    // diagonal scoring, takeOnlyBestKmer (exact k-mers), isNucleotide
    MatcherProbe matcher(c->index, c->lookup, c->mat, c->mat, 0, c->kmerSize, c->dbSize, ml, max_hits, true, 1.0f, true, min_diag_score,
                         true, true);
    if (force_bins) matcher.forceBins(force_bins);
    ScoreMatrix none3, none2;      // Prefiltering::runSplit hands its (empty) 3-mer / 2-mer tables over for nucleotides as well (:835)
    matcher.setSubstitutionMatrix(&none3, &none2);
    Sequence seq(ml, Parameters::DBTYPE_NUCLEOTIDES, c->mat, c->kmerSize, spaced != 0, true, true);
    seq.mapSequence(0, 0, std::make_pair((const unsigned char *)q, (const unsigned int)qlen));
    DBLocalId ident = identity_id == UINT32_MAX ? DB_LOCAL_ID_INVALID : (DBLocalId)identity_id;
    std::pair<hit_t *, size_t> r = matcher.matchQuery(&seq, ident, true);
    for (size_t i = 0; i < r.second && i < cap; i++) {
        ids[i] = (uint32_t)r.first[i].seqId;
        scores[i] = r.first[i].prefScore;
        diags[i] = r.first[i].diagonal;
    }
    if (db_matches) *db_matches = matcher.getStatistics()->dbMatches;
    return r.second;
}

// CPU baseline: the query loop of Prefiltering::runSplit (Prefiltering.cpp:820-917) - one QueryMatcher + Sequence per
// OpenMP thread, dynamic schedule - over nq queries; returns the wall time of the loop (matcher construction excluded,
// as the reference's own "Time for processing" excludes setup) and the total number of hits / index matches.
double mmref_pref_match_batch(void *h, const uint8_t *qdata, const uint64_t *qoff, uint32_t nq, int n_threads, int kmer_thr,
                              unsigned max_seq_len, uint64_t max_hits, int comp_bias, unsigned min_diag_score, int spaced,
                              uint64_t *total_hits, uint64_t *total_db_matches, uint32_t *hit_counts,
                              uint32_t *hit_ids, int32_t *hit_scores, uint16_t *hit_diags, unsigned *bins_used) {
    PrefCtx *c = (PrefCtx *)h;
    unsigned ml = std::max(max_seq_len, c->maxLen);
    uint64_t hits = 0, dbm = 0;
    double t_loop = 0;
#pragma omp parallel num_threads(n_threads) reduction(+ : hits, dbm)
    {
        MatcherProbe matcher(c->index, c->lookup, c->kmerMat, c->ungappedMat, (short)kmer_thr, c->kmerSize, c->dbSize, ml,
                             max_hits, comp_bias != 0, 1.0f, true, min_diag_score, false, false);
#pragma omp master
        if (bins_used) *bins_used = matcher.bins();
        matcher.setSubstitutionMatrix(&c->three, &c->two);
        Sequence seq(ml, Parameters::DBTYPE_AMINO_ACIDS, c->kmerMat, c->kmerSize, spaced != 0, comp_bias != 0, true);
#pragma omp barrier
        double t0 = omp_get_wtime();
#pragma omp for schedule(dynamic, 1)
        for (uint32_t i = 0; i < nq; i++) {
            seq.mapSequence(i, i, std::make_pair((const unsigned char *)(qdata + qoff[i]), (const unsigned int)(qoff[i + 1] - qoff[i])));
            std::pair<hit_t *, size_t> r = matcher.matchQuery(&seq, DB_LOCAL_ID_INVALID, false);
            hits += r.second;
            dbm += matcher.getStatistics()->dbMatches;
            if (hit_counts) hit_counts[i] = (uint32_t)r.second;
            if (hit_ids)   // full lists, for the full-size parity check of bench.py
                for (size_t z = 0; z < r.second && z < max_hits; z++) {
                    hit_ids[(size_t)i * max_hits + z] = (uint32_t)r.first[z].seqId;
                    hit_scores[(size_t)i * max_hits + z] = r.first[z].prefScore;
                    hit_diags[(size_t)i * max_hits + z] = r.first[z].diagonal;
                }
        }
        double t1 = omp_get_wtime();   // after the implicit barrier of the omp for
#pragma omp master
        t_loop = t1 - t0;
    }
    if (total_hits) *total_hits = hits;
    if (total_db_matches) *total_db_matches = dbm;
    return t_loop;
}

}  // extern "C"
