// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
//
// oracle/_ref/libmmref.so, nucleotide part: extern "C" shim over the REAL reference classes
//   BandedNucleotideAligner::initQuery / align          src/alignment/BandedNucleotideAligner.cpp:52-263
//   ksw_extz2_sse (+ ksw_backtrack, ksw_apply_zdrop)     lib/ksw2/ksw2_extz2_sse.cpp:44-285, lib/ksw2/ksw2.h:116-199
//   NucleotideMatrix(nucleotide.out, 1.0, 0.0)           src/alignment/Alignment.cpp:147-150
// compiled where the sources lie (oracle/Makefile); used by tests/ and tests/golden/make_golden.py only, to pin
// oracle/nucl_oracle.c and to generate committed fixtures.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "Debug.h"
#include "EvalueComputation.h"
#include "NucleotideMatrix.h"
#include "Parameters.h"
#include "Sequence.h"
#include "StripedSmithWaterman.h"
#include "ksw2.h"
#include <sstream>
#include <iostream>
#include <vector>
#include <map>

// The reference reads ONE residue past the end of the query and of the target: SmithWaterman::seq_reverse is called
// with L where it expects L - 1 (BandedNucleotideAligner.cpp:61,68,93; StripedSmithWaterman.h:224-233), so the
// reversed copies are shifted by one and start with numSequence[L] - stale buffer content in a real run.  To make
// that input explicit the shim writes a caller-chosen letter there (`past_end`), which needs the aligner's private
// reverse-complement buffer.
#define private public
#include "BandedNucleotideAligner.h"
#undef private

extern "C" {

struct mmref_nucl_result {
    int32_t score;
    int32_t q_start, q_end, t_start, t_end;
    uint32_t ident;
    int32_t bt_len;
    int32_t cigar_len;
};

struct mmref_nucl_ctx {
    NucleotideMatrix *m;
    EvalueComputation *evaluer;
    BandedNucleotideAligner *al;
    Sequence *q;
    Sequence *t;
};

mmref_nucl_ctx *mmref_nucl_new(const char *matrix_file, int max_len, int gap_open, int gap_extend, int zdrop,
                               uint64_t db_residues) {
    Debug::setDebugLevel(Debug::ERROR);
    mmref_nucl_ctx *c = new mmref_nucl_ctx();
    c->m = new NucleotideMatrix(matrix_file, 1.0f, 0.0f);
    c->evaluer = new EvalueComputation(db_residues, c->m, gap_open, gap_extend);
    c->al = new BandedNucleotideAligner(c->m, max_len, gap_open, gap_extend, zdrop);
    c->q = new Sequence(max_len, Parameters::DBTYPE_NUCLEOTIDES, c->m, 0, false, false);
    c->t = new Sequence(max_len, Parameters::DBTYPE_NUCLEOTIDES, c->m, 0, false, false);
    return c;
}

void mmref_nucl_free(mmref_nucl_ctx *c) {
    delete c->q;
    delete c->t;
    delete c->al;
    delete c->evaluer;
    delete c->m;
    delete c;
}

int mmref_nucl_alphabet(mmref_nucl_ctx *c) { return c->m->alphabetSize; }

void mmref_nucl_matrix(mmref_nucl_ctx *c, int8_t *out) {
    int a = c->m->alphabetSize;
    for (int i = 0; i < a; i++)
        for (int j = 0; j < a; j++) out[i * a + j] = (int8_t)c->m->subMatrix[i][j];
}

void mmref_nucl_aa2num(mmref_nucl_ctx *c, const char *seq, int len, uint8_t *out) {
    for (int i = 0; i < len; i++) out[i] = c->m->aa2num[(unsigned char)seq[i]];
}

void mmref_nucl_reverse_lookup(mmref_nucl_ctx *c, uint8_t *out) {
    for (int i = 0; i < c->m->alphabetSize; i++) out[i] = (uint8_t)c->m->reverseResidue(i);
}

// the char data must outlive the alignments (Sequence keeps the pointer, BandedNucleotideAligner.cpp:82)
void mmref_nucl_set_query(mmref_nucl_ctx *c, const char *seq, int len, int past_end) {
    c->q->mapSequence(0, 0, seq, (unsigned int)len);
    c->q->numSequence[len] = (unsigned char)past_end;
    c->al->queryRevCompSeq[len] = (uint8_t)past_end;     // read by the second seq_reverse of initQuery (:68)
    c->al->initQuery(c->q);
}

void mmref_nucl_align(mmref_nucl_ctx *c, const char *tseq, int tlen, int past_end, int diagonal, int reverse, int wrapped,
                      mmref_nucl_result *res, char *bt, int bt_cap) {
    c->t->mapSequence(1, 1, tseq, (unsigned int)tlen);
    c->t->numSequence[tlen] = (unsigned char)past_end;
    std::string backtrace;
    s_align a = c->al->align(c->t, diagonal, reverse != 0, backtrace, c->evaluer, wrapped != 0);
    res->score = (int32_t)a.score1;
    res->q_start = a.qStartPos1;
    res->q_end = a.qEndPos1;
    res->t_start = a.dbStartPos1;
    res->t_end = a.dbEndPos1;
    res->ident = a.identicalAACnt;
    res->cigar_len = a.cigarLen;
    res->bt_len = 0;
    if (bt != NULL && (int)backtrace.size() < bt_cap) {
        memcpy(bt, backtrace.data(), backtrace.size());
        bt[backtrace.size()] = 0;
        res->bt_len = (int)backtrace.size();
    }
    delete[] a.cigar;
}

// the extension kernel alone: out[0..8] = max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score; returns n_cigar
int mmref_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo,
                    int gape, int w, int zdrop, int flag, int32_t *out, uint32_t *cigar, int cigar_cap) {
    ksw_extz_t ez;
    memset(&ez, 0, sizeof(ez));
    ksw_extz2_sse(0, qlen, query, tlen, target, (int8_t)m, mat, (int8_t)gapo, (int8_t)gape, w, zdrop, flag, &ez);
    out[0] = (int32_t)ez.max;
    out[1] = (int32_t)ez.zdropped;
    out[2] = ez.max_q;
    out[3] = ez.max_t;
    out[4] = ez.mqe;
    out[5] = ez.mqe_t;
    out[6] = ez.mte;
    out[7] = ez.mte_q;
    out[8] = ez.score;
    int n = ez.n_cigar;
    for (int i = 0; i < n && i < cigar_cap; i++) cigar[i] = ez.cigar[i];
    free(ez.cigar);
    return n;
}

}  // extern "C"

// Batch form for bench.py's cpu_baseline leg: pairs spread over OpenMP threads, one BandedNucleotideAligner and one
// pair of Sequence objects per thread (what Alignment::run keeps per thread, Alignment.cpp:279-295).  Sequences are
// ASCII (A C G T N); pairs must be grouped by query (the query is re-initialised when it changes).
#include <omp.h>
extern "C" double mmref_nucl_batch(const char *matrix_file, int max_len, int gap_open, int gap_extend, int zdrop, int past_end,
                                   int n_threads, const char *q_chars, const uint64_t *q_off, const char *t_chars,
                                   const uint64_t *t_off, const uint32_t *pair_q, const uint32_t *pair_t, const uint16_t *pair_diag,
                                   const uint8_t *pair_rev, uint32_t n_pairs, int32_t *out /* n_pairs x 6 */, uint32_t *bt_len) {
    Debug::setDebugLevel(Debug::ERROR);
    std::vector<mmref_nucl_ctx *> ctxs(n_threads);
    for (int t = 0; t < n_threads; t++) ctxs[t] = mmref_nucl_new(matrix_file, max_len, gap_open, gap_extend, zdrop, 100000000ull);
    const double t0 = omp_get_wtime();
#pragma omp parallel num_threads(n_threads)
    {
        mmref_nucl_ctx *c = ctxs[omp_get_thread_num()];
        uint32_t cur_q = 0xFFFFFFFFu;
        std::vector<char> bt;
#pragma omp for schedule(dynamic, 4)
        for (uint32_t i = 0; i < n_pairs; i++) {
            const uint32_t q = pair_q[i], t = pair_t[i];
            if (q != cur_q) {
                mmref_nucl_set_query(c, q_chars + q_off[q], (int)(q_off[q + 1] - q_off[q]), past_end);
                cur_q = q;
            }
            const int tlen = (int)(t_off[t + 1] - t_off[t]);
            bt.resize((size_t)(q_off[q + 1] - q_off[q]) + tlen + 8);
            mmref_nucl_result r;
            mmref_nucl_align(c, t_chars + t_off[t], tlen, past_end, pair_diag[i], pair_rev[i], 0, &r, bt.data(), (int)bt.size());
            out[i * 6 + 0] = r.score; out[i * 6 + 1] = r.q_start; out[i * 6 + 2] = r.q_end;
            out[i * 6 + 3] = r.t_start; out[i * 6 + 4] = r.t_end; out[i * 6 + 5] = (int32_t)r.ident;
            bt_len[i] = (uint32_t)r.bt_len;
        }
    }
    const double dt = omp_get_wtime() - t0;
    for (int t = 0; t < n_threads; t++) mmref_nucl_free(ctxs[t]);
    return dt;
}

// "name:data" form of the matrix (BaseMatrix::serialize) so that the prebuilt library can build its NucleotideMatrix
// on the GPU box, where /root/reference/data does not exist (stored in tests/golden/matrices.npz by make_golden.py)
extern "C" int mmref_nucl_serialized_matrix(mmref_nucl_ctx *c, char *out, int cap) {
    char *s = BaseMatrix::serialize(c->m->matrixName, c->m->matrixData);
    int n = (int)strlen(s);
    if (n + 1 > cap) { free(s); return -n; }
    memcpy(out, s, (size_t)n + 1);
    free(s);
    return n;
}
