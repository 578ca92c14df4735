// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
//
// oracle/_ref/libmmref.so: a thin extern "C" shim over the *real* reference classes,
// compiled by oracle/Makefile from the sources where they lie under /root/reference
// (no reference source is copied into this repository).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; it is used to
//   (1) pin the plain-C restatement in oracle/*.c,
//   (2) generate the committed fixtures under tests/golden/ (tests/golden/make_golden.py),
//   (3) serve as the "reference" CPU baseline (AVX2 striped Smith-Waterman, uint8->int16).
//
// Reference entry points driven here:
//   SubstitutionMatrix(blosum62.out, 2.0, 0.0)              src/alignment/Alignment.cpp:152
//   Sequence::mapSequence                                    src/alignment/Alignment.cpp:339,367
//   SmithWaterman::ssw_init / ssw_align                      src/alignment/Matcher.cpp:58,82
//   EvalueComputation(dbResidues, m, gapOpen, gapExtend)     src/alignment/Alignment.cpp:263
//   UngappedAlignment::createProfile / scoreSingelSequence…  src/prefiltering/QueryMatcher.cpp:119
//   SubstitutionMatrix::calcLocalAaBiasCorrection            src/commons/SubstitutionMatrix.cpp:79-112
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <omp.h>
#include <string>
#include <vector>

#include "Debug.h"
#include "EvalueComputation.h"
#include "Parameters.h"
#include "Sequence.h"
#include "StripedSmithWaterman.h"
#include "SubstitutionMatrix.h"

extern "C" {

struct mmref_sw_result {
    uint32_t score;
    int32_t q_start, q_end, t_start, t_end;
    int32_t word;            // 0: uint8 pass sufficed, 1: int16 re-run (StripedSmithWaterman.cpp:916-920)
    float q_cov, t_cov;
    double evalue;
    uint32_t ident;          // identicalAACnt (only with backtrace)
    int32_t bt_len;          // length of backtrace string written to bt (0 if none)
};

struct mmref_ctx {
    SubstitutionMatrix *m;
    EvalueComputation *evaluer;
    SmithWaterman *sw;
    Sequence *q;
    Sequence *t;
    Sequence *qprof;      // profile query (DBTYPE_HMM_PROFILE), created on first use
    bool profileQuery;
    int gapOpen, gapExtend;
    size_t maxLen;
    bool compBias;
};

// matrix_file: path to <reference>/data/blosum62.out (read at run time by the reference's own
// SubstitutionMatrix; tests that need it are skipped when /root/reference is absent).
mmref_ctx *mmref_new(const char *matrix_file, float bit_factor, float score_bias, int max_len, int gap_open,
                     int gap_extend, int comp_bias, uint64_t db_residues) {
    Debug::setDebugLevel(Debug::ERROR);
    mmref_ctx *c = new mmref_ctx();
    c->m = new SubstitutionMatrix(matrix_file, bit_factor, score_bias);
    c->maxLen = max_len;
    c->compBias = comp_bias != 0;
    c->evaluer = NULL;
    c->sw = NULL;
    if (gap_open > 0) {   // gap_open <= 0: matrix-only context (k-mer / ungapped matrices have no Gumbel table)
        c->evaluer = new EvalueComputation(db_residues, c->m, gap_open, gap_extend);
        c->sw = new SmithWaterman(max_len, c->m->alphabetSize, c->compBias, 1.0f, c->m);
    }
    c->q = new Sequence(max_len, Parameters::DBTYPE_AMINO_ACIDS, c->m, 0, false, c->compBias);
    c->t = new Sequence(max_len, Parameters::DBTYPE_AMINO_ACIDS, c->m, 0, false, c->compBias);
    c->qprof = NULL;
    c->profileQuery = false;
    c->gapOpen = gap_open;
    c->gapExtend = gap_extend;
    return c;
}

void mmref_free(mmref_ctx *c) {
    delete c->qprof;
    delete c->q;
    delete c->t;
    delete c->sw;
    delete c->evaluer;
    delete c->m;
    delete c;
}

int mmref_alphabet_size(mmref_ctx *c) { return c->m->alphabetSize; }

// int matrix as the aligner sees it (Matcher::setSubstitutionMatrix, Matcher.cpp:29-36)
void mmref_get_matrix(mmref_ctx *c, int8_t *out /*alphabet^2*/) {
    int a = c->m->alphabetSize;
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) out[i * a + j] = (int8_t)c->m->subMatrix[i][j];
}

// ASCII -> numeric with the reference's aa2num table
void mmref_aa2num(mmref_ctx *c, const char *seq, int len, uint8_t *out) {
    for (int i = 0; i < len; i++) out[i] = c->m->aa2num[(unsigned char)seq[i]];
}

void mmref_num2aa(mmref_ctx *c, char *out /*alphabet*/) {
    for (int i = 0; i < c->m->alphabetSize; i++) out[i] = c->m->num2aa[i];
}

// float composition bias as the reference computes it (SubstitutionMatrix.cpp:79-112)
void mmref_comp_bias(mmref_ctx *c, const uint8_t *num, int len, float scale, float *out) {
    SubstitutionMatrix::calcLocalAaBiasCorrection(c->m, num, len, out, scale);
}

// query given as numeric codes (SequenceLookup-style mapSequence overload, Sequence.h:88)
void mmref_sw_set_query(mmref_ctx *c, const uint8_t *qnum, int qlen) {
    c->profileQuery = false;
    c->q->mapSequence(0, 0, std::make_pair((const unsigned char *)qnum, (const unsigned int)qlen));
    int a = c->m->alphabetSize;
    std::vector<int8_t> tiny(a * a);
    mmref_get_matrix(c, tiny.data());
    c->sw->ssw_init(c->q, tiny.data(), c->m);
}

// Profile query: `data` is one entry of a profile database, Sequence::PROFILE_READIN_SIZE (25) bytes per position
// (20 scores x 4, query letter, consensus letter, Neff, 2 reserved; Sequence.cpp:301-340).  Alignment::run maps it with
// Sequence::mapSequence (-> mapProfile) and Matcher::initQuery hands getAlignmentProfile() to ssw_init (Matcher.cpp:49-60).
// prof_out [PROFILE_AA_SIZE * qlen] / cons_out [qlen] receive what the reference aligns with (may be NULL).
void mmref_sw_set_profile_query(mmref_ctx *c, const char *data, int qlen, int8_t *prof_out, uint8_t *cons_out) {
    if (c->qprof == NULL) c->qprof = new Sequence(c->maxLen, Parameters::DBTYPE_HMM_PROFILE, c->m, 0, false, false);
    c->profileQuery = true;
    c->qprof->mapSequence(0, 0, data, (unsigned int)qlen);
    c->sw->ssw_init(c->qprof, c->qprof->getAlignmentProfile(), c->m);
    if (prof_out) memcpy(prof_out, c->qprof->getAlignmentProfile(), Sequence::PROFILE_AA_SIZE * (size_t)c->qprof->L);
    if (cons_out) memcpy(cons_out, c->qprof->numSequence, (size_t)c->qprof->L);
}

// mode: Matcher::SCORE_ONLY=0, SCORE_COV=1, SCORE_COV_SEQID=2 (Matcher.h:24-26)
void mmref_sw_align(mmref_ctx *c, const uint8_t *tnum, int tlen, int mode, double evalue_thr, int cov_mode,
                    float cov_thr, mmref_sw_result *res, char *bt, int bt_cap) {
    std::string backtrace;
    int32_t maskLen = (c->profileQuery ? c->qprof->L : c->q->L) / 2;
    s_align a = c->sw->ssw_align(tnum, tlen, backtrace, c->gapOpen, c->gapExtend, mode, evalue_thr, c->evaluer,
                                 cov_mode, cov_thr, 0.0f, maskLen);
    res->score = a.score1;
    res->q_start = a.qStartPos1;
    res->q_end = a.qEndPos1;
    res->t_start = a.dbStartPos1;
    res->t_end = a.dbEndPos1;
    res->word = a.word;
    res->q_cov = a.qCov;
    res->t_cov = a.tCov;
    res->evalue = a.evalue;
    res->ident = a.identicalAACnt;
    res->bt_len = 0;
    if (bt != NULL && (int)backtrace.size() < bt_cap) {
        memcpy(bt, backtrace.data(), backtrace.size());
        bt[backtrace.size()] = 0;
        res->bt_len = (int)backtrace.size();
    }
    delete[] a.cigar;
}

// batch: one query against n targets, score/end only; returns wall-clock-free results.
// Used by bench.py's cpu_baseline ("reference" kind) from several threads, one ctx per thread.
void mmref_sw_batch_score(mmref_ctx *c, const uint8_t *tdata, const uint64_t *toff, const uint32_t *ids, int n,
                          uint32_t *score, int32_t *qend, int32_t *tend) {
    std::string backtrace;
    int32_t maskLen = c->q->L / 2;
    for (int i = 0; i < n; i++) {
        uint32_t id = ids[i];
        const uint8_t *t = tdata + toff[id];
        int tlen = (int)(toff[id + 1] - toff[id]);
        s_align a = c->sw->ssw_align(t, tlen, backtrace, c->gapOpen, c->gapExtend, 0, 1e300, c->evaluer, 0, 0.0f,
                                     0.0f, maskLen);
        score[i] = a.score1;
        qend[i] = a.qEndPos1;
        tend[i] = a.dbEndPos1;
    }
}

// Alignment::run's inner loop as ONE native call (bench.py's cpu_baseline): OpenMP over the queries,
// one SmithWaterman + one query Sequence per thread exactly as Alignment.cpp:279-295, dynamic schedule like the reference's
// `#pragma omp for schedule(dynamic, 5)` (:313); per query ssw_init (Matcher::initQuery, Matcher.cpp:49-60) and then
// ssw_align of every list entry in list order (Matcher::getSWResult, :62-144).  mode 0 = score + end positions
// (Matcher::SCORE_ONLY); mode 1 = start positions too for pairs whose E-value passes evalue_thr (SCORE_COV), which is what
// `mmseqs search` runs.  Nothing is compared or converted inside the timed region; results land in caller arrays.
// Returns the wall seconds between the barrier after thread set-up and the end of the loop.
double mmref_sw_lists_omp(const char *matrix_file, int max_len, int gap_open, int gap_extend, int comp_bias,
                          uint64_t db_residues, int n_threads, int mode, double evalue_thr,
                          const uint8_t *qdata, const uint64_t *qoff, uint32_t q_from, uint32_t q_to,
                          const uint32_t *list_ids, const uint64_t *list_off,
                          const uint8_t *tdata, const uint64_t *toff,
                          uint32_t *score, int32_t *qend, int32_t *tend, int32_t *qstart, int32_t *tstart,
                          int *threads_used) {
    Debug::setDebugLevel(Debug::ERROR);
    SubstitutionMatrix m(matrix_file, 2.0f, 0.0f);
    EvalueComputation evaluer(db_residues, &m, gap_open, gap_extend);
    int a = m.alphabetSize;
    std::vector<int8_t> tiny(a * a);
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) tiny[i * a + j] = (int8_t)m.subMatrix[i][j];
    double seconds = 0.0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        SmithWaterman sw(max_len, a, comp_bias != 0, 1.0f, &m);
        Sequence q(max_len, Parameters::DBTYPE_AMINO_ACIDS, &m, 0, false, comp_bias != 0);
        std::string backtrace;
#pragma omp master
        *threads_used = omp_get_num_threads();
#pragma omp barrier
        double t0 = omp_get_wtime();
#pragma omp for schedule(dynamic, 5)
        for (int64_t qi = (int64_t)q_from; qi < (int64_t)q_to; qi++) {
            uint64_t b = list_off[qi], e = list_off[qi + 1];
            if (b == e) continue;
            q.mapSequence(0, 0, std::make_pair((const unsigned char *)(qdata + qoff[qi]),
                                               (const unsigned int)(qoff[qi + 1] - qoff[qi])));
            sw.ssw_init(&q, tiny.data(), &m);
            int32_t maskLen = q.L / 2;
            for (uint64_t k = b; k < e; k++) {
                uint32_t id = list_ids[k];
                int tlen = (int)(toff[id + 1] - toff[id]);
                s_align r = sw.ssw_align(tdata + toff[id], tlen, backtrace, gap_open, gap_extend, mode, evalue_thr, &evaluer,
                                         0, 0.0f, 0.0f, maskLen);
                score[k] = r.score1;
                qend[k] = r.qEndPos1;
                tend[k] = r.dbEndPos1;
                if (qstart) { qstart[k] = r.qStartPos1; tstart[k] = r.dbStartPos1; }
                delete[] r.cigar;
            }
        }
        double t1 = omp_get_wtime();   // after the implicit barrier of the omp for
#pragma omp master
        seconds = t1 - t0;
    }
    return seconds;
}

double mmref_evalue(mmref_ctx *c, double score, double qlen) { return c->evaluer->computeEvalue(score, qlen); }
double mmref_bitscore(mmref_ctx *c, double score) { return c->evaluer->computeBitScore(score); }

}  // extern "C"

extern "C" void mmref_get_pback(mmref_ctx *c, double *out) {
    for (int i = 0; i < c->m->alphabetSize; i++) out[i] = c->m->pBack[i];
}
extern "C" void mmref_get_matrix16(mmref_ctx *c, int16_t *out) {
    int a = c->m->alphabetSize;
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) out[i * a + j] = c->m->subMatrix[i][j];
}

// "name.out:DATA" form accepted by the SubstitutionMatrix constructor (SubstitutionMatrix.cpp:13-19,
// BaseMatrix::serialize, BaseMatrix.cpp:176-187).  tests/golden/make_golden.py stores it so the prebuilt
// libmmref.so can construct its matrices on the GPU box, where /root/reference/data does not exist.
extern "C" int mmref_serialized_matrix(mmref_ctx *c, char *out, int cap) {
    char *s = BaseMatrix::serialize(c->m->matrixName, c->m->matrixData);
    int n = (int)strlen(s);
    if (n + 1 > cap) { free(s); return -n; }
    memcpy(out, s, (size_t)n + 1);
    free(s);
    return n;
}
