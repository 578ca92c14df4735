// Entry points the patched reference calls (integration/mmseqs_mmgpu.patch): one static function per seam, each
// returning false when the configuration is not covered by the device path so that the reference's own CPU loop runs.
//
//   Alignment::run          -> MMGpuAlignRun::run        (src/alignment/Alignment.cpp:248)
//   Prefiltering::runSplit  -> MMGpuPrefilterRun::run    (src/prefiltering/Prefiltering.cpp:820, the query loop)
//
// Run-time switches (environment, read once):
//   MMGPU_DISABLE=1            keep the CPU path everywhere (the binary then behaves like the stock one)
//   MMGPU_DEVICE=<n>           HIP device of this process (default 0)
//   MMGPU_DEVICES=a,b,...      two or more devices for one module call.  `prefilter`: G query groups x S target shards - a
//                              group deals the target database to its S devices by length bucket (one k-mer index per
//                              device, the hit lists exchanged over the group's RCCL communicator, merged lists = the unsplit
//                              run's) and the query blocks are dealt to the groups; `align` keeps the targets on every device
//                              and deals the queries of each block to them
//   MMGPU_QUERY_GROUPS=G       G of that layout (default: the stage model of MMGpuRun::queryGroups, S >= 2)
//   MMGPU_BLOCK_ALIGNER        hits whose score left the uint8 range (s_align::word == 1) take start position, identities and
//                              backtrace from the block aligner (StripedSmithWaterman.cpp:865-882,943-1127):
//                                device (default)  the device's block aligner (mmgpu_sw_block_backtrace); what it declines as
//                                                  too large goes to the host's own alignStartPosBacktraceBlock
//                                host              the host's alignStartPosBacktraceBlock for every such pair: those fields
//                                                  are the stock binary's by construction (a Rust-linked build: the crate's)
//                                sw                the reverse scan + banded traceback for all of them (the reference's
//                                                  fallback; what rounds 1-2 called "device")
//   MMGPU_ALIGN_BLOCK_QUERIES, MMGPU_ALIGN_BLOCK_BYTES, MMGPU_PREF_BLOCK_QUERIES   block sizes of the device calls
#ifndef MMGPU_RUN_H
#define MMGPU_RUN_H

#include <sys/time.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <string>

#include "Debug.h"
#include "Matcher.h"
#include "ScoreMatrix.h"
#include "BaseMatrix.h"

#include <vector>

#include "mmgpu.h"

class Alignment;
class Prefiltering;
class DBWriter;

class MMGpuRun {
public:
    static bool enabled();
    static bool hostBlockAligner();
    static bool deviceBlockAligner();
    static size_t envSize(const char *name, size_t fallback);
    // the process-wide context; logs the library's message and EXITs if the device cannot be opened
    static mmgpu_ctx *context();
    // MMGPU_DEVICES (two ids or more; empty otherwise)
    static const std::vector<int> &deviceIds();
    // G of the G query groups x S target shards layout of a prefilter run over n contexts (MMGPU_QUERY_GROUPS, or the stage
    // model's choice); shardsPossible = the configuration can run with its targets dealt to several contexts
    static int queryGroups(int nDevices, bool shardsPossible);
    // the contexts of MMGPU_DEVICES as g multi-device objects of n / g contexts each (mmgpu_init_multi; g = 0: the layout
    // opened last, one group if none); empty without MMGPU_DEVICES.  Opening another layout closes the previous one.
    // shardsOfOneDevice > 1 without MMGPU_DEVICES: that many contexts on the one device (a target split of more than
    // MMGPU_PF_MAX_TARGETS sequences is dealt to them like to several devices)
    static const std::vector<mmgpu_multi *> &groups(int g, int shardsOfOneDevice = 0);
    static std::vector<mmgpu_ctx *> allContexts();
};

struct MMGpuAlignSession;
class EvalueComputation;
class Sequence;

// Alignment::run (integration/mmseqs_mmgpu.patch): begin before the bucket loop, plan before a bucket's OpenMP region, take at
// getSWResult's call site, end after the loop (MMGpuAlignRun.cpp; nucleotide databases: MMGpuNuclAlignRun.cpp)
class MMGpuAlignRun {
public:
    static bool usable(const Alignment &a);
    static MMGpuAlignSession *begin(Alignment &a, EvalueComputation &evaluer, size_t dbFrom, size_t dbSize);      // NULL: CPU path
    static size_t bucketQueries(MMGpuAlignSession *s);
    static void plan(MMGpuAlignSession *s, size_t start, size_t bucketSize);
    static Matcher::result_t take(MMGpuAlignSession *s, size_t id, size_t entry, Matcher &matcher, Sequence *dbSeq, int diagonal, bool isReverse,
                                  bool isIdentity);
    static void end(MMGpuAlignSession *s);
    // Sequence::numSequence of target `id`: out of the session's host copy, or - where the targets came to the device from a persisted
    // layout and the host holds none - mapped now and kept for the bucket
    static const unsigned char *targetResidues(MMGpuAlignSession *s, size_t id);
    // nucleotide databases: BandedNucleotideAligner::align on the device (MMGPU_NUCL_ALIGN=0 keeps the CPU loop)
    static bool usableNucleotide(const Alignment &a);
    static void beginNucleotide(MMGpuAlignSession *s);
    static void planNucleotide(MMGpuAlignSession *s);
    static void endNucleotide(MMGpuAlignSession *s);
};

struct MMGpuPrefilterStats {
    double kmersPerPos;
    size_t dbMatches, doubleMatches, querySeqLenSum, resSize, diagonalOverflow;
};

class MMGpuPrefilterRun {
public:
    static bool usable(Prefiltering &p);
    static bool usableConfig(Prefiltering &p, bool indexExists);
    // Prefiltering::getIndexTable asks before it builds the index: true = the device will build it from the masked
    // SequenceLookup (mmgpu_pf_build_index), the host only masks (IndexBuilder::fillDatabase without an index table);
    // MMGPU_HOST_INDEX=1 keeps the host's index
    // Prefiltering's constructor: the score-sorted 2-mer / 3-mer tables (getScoreMatrix -> ExtendedSubstitutionMatrix::calcScoreMatrix,
    // 0.45 s of every prefilter process for the 8000 x 8000 table) by the library's stable counting sort - the same table
    // (mmgpu_host_score_matrix_rows); a precomputed index or a disabled device path keeps the reference's own function
    static ScoreMatrix scoreMatrix(Prefiltering &p, const BaseMatrix &matrix, size_t kmerSize);
    static bool deviceBuildsIndex(Prefiltering &p);
    // ... and whether the device also does the masking fillDatabase would do (tantan only, one device): mmgpu_pf_mask_targets;
    // MMGPU_DEVICE_MASK=0 keeps the host's Masker
    static bool deviceMasks(Prefiltering &p);
    // ... and, before fillDatabase maps a single sequence: MMGPU_DB_FILE names a persisted device layout (mmgpu_db_save) made from
    // this target database with this run's index parameters - it is loaded (targets, masked view, index) and the split runs without
    // a SequenceLookup on the host.  false: nothing loaded, the lookup is filled and handed over as ever (and the layout saved)
    static bool loadPersisted(Prefiltering &p, size_t dbFrom, size_t dbSize);
    // `mmseqs makemmgpudb` (MMGpuMakeDb.cpp): builds the device layout of p's target database and saves it to MMGPU_DB_FILE
    static bool buildAndSave(Prefiltering &p);
    // fused search: this object will run unsplit through the device path and leave its targets resident for the alignment
    // module (MMGpuFusedSearch::keepResidentTargets) - the alignment module can then start before the prefilter has finished
    static bool runsUnsplitWithResidentTargets(Prefiltering &p, size_t *maxResListLen);
    // Prefiltering::runSplit, before it opens its DBWriter: every entry of this split will go to the fused search's in-memory
    // store (MMGpuPrefilterRun::run decides the same way), none to the writer
    static bool keepsEntriesInMemory(Prefiltering &p, const std::string &resultDB, size_t dbSize);
    // the reference's own index for queries the device hands back (overflow, long sequences, ties), built when first needed
    static void ensureHostIndex(Prefiltering &p, size_t dbFrom, size_t dbSize);
    // the `omp parallel` block of Prefiltering::runSplit (:820-918): writes every query's entry to tmpDbw, fills the
    // statistics the caller prints
    static bool run(Prefiltering &p, DBWriter &tmpDbw, size_t dbFrom, size_t dbSize, size_t queryFrom, size_t querySize,
                    char *notEmpty, std::list<int> **reslens, size_t localThreads, Debug::Progress &progress,
                    MMGpuPrefilterStats &stats);
};

// MMGPU_TRACE=1: where the wall time of a hooked module goes (stderr), e.g. "[mmgpu prefilter] load index 1.234 s"
class MMGpuStopwatch {
public:
    explicit MMGpuStopwatch(const char *module) : module(module), on(getenv("MMGPU_TRACE") != NULL), last(0) {
        for (int i = 0; i < 8; i++) acc[i] = 0;
        last = now();
    }
    void lap(const char *what) {         // time since the previous lap
        const double t = now();
        if (on) fprintf(stderr, "[mmgpu %s] %s %.3f s\n", module, what, t - last);
        last = t;
    }
    void add(int slot, double s) { acc[slot] += s; }
    double now() const {
        struct timeval tv;
        gettimeofday(&tv, NULL);
        return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
    }
    void report(const char *const *names, int n) {
        if (!on) return;
        for (int i = 0; i < n && i < 8; i++) fprintf(stderr, "[mmgpu %s] %s %.3f s (all blocks)\n", module, names[i], acc[i]);
    }
private:
    const char *module;
    bool on;
    double last;
    double acc[8];
};

#endif
