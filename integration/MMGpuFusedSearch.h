// `mmseqs search` without the two child processes: the plain sequence-search workflow
// (data/workflow/blastp.sh with PREFMODE=KMER, one sensitivity step: `prefilter` then `align`) run inside the `search` process.
// The precedent in the reference is the single module ungappedprefilter --prefilter-mode 3 (ungappedprefilter.cpp:282-299,
// blastp.sh:70,85); here the two modules keep their own code and parameters - each one is started through its Command entry
// with exactly the argument string blastp.sh would pass - and share what a second process would have to set up again:
//   * ONE device context (opened in the background while the prefilter module reads and masks the targets),
//   * the library's working-buffer cache (no second round of hipMalloc),
//   * the hit lists: the prefilter hook keeps its result entries in memory (MMGpuFusedSearch::PrefStore) and the alignment
//     module reads them from there - no pref_0 database is written or parsed from disk (MMGPU_FUSED_PREF_ON_DISK=1 keeps it).
// MMGPU_FUSED=0 keeps the stock workflow script.
#ifndef MMGPU_FUSED_SEARCH_H
#define MMGPU_FUSED_SEARCH_H

#include <string>
#include <vector>

#include "DBReader.h"

class Parameters;
class SequenceLookup;

class MMGpuFusedSearch {
public:
    // Search.cpp's generic branch (blastp.sh) only; false = run the workflow script as ever
    static bool usable(const Parameters &par, bool isUngappedMode, int searchMode, const std::string &program, const std::string &tmpDir);
    // prefilterPar / alignPar: the strings Search.cpp hands to blastp.sh as PREFILTER_PAR / ALIGNMENT_PAR; sens: SENSE_0
    static int run(Parameters &par, const std::string &query, const std::string &target, const std::string &result, const std::string &tmpDir,
                   const std::string &prefilterPar, const std::string &alignPar, const std::string &sens, bool removeTmp);

    // ---- the prefilter result of the fused run, kept in memory ----
    // active(): a fused run is in progress and the prefilter database named `db` is the one to keep in memory
    static bool capturing(const std::string &db);
    // prefilter hook: the serialised hit list of one query (what DBWriter::writeData would receive), any thread
    // (targetKeys / nTargets: the target keys of the entry's lines, in order - what a reader would parse out of them again)
    static void capture(unsigned int queryKey, const char *data, size_t len, unsigned int thread, const unsigned int *targetKeys,
                        size_t nTargets);
    // Alignment's constructor: a DBReader over the captured entries (NULL = none for this name: open the database on disk)
    static DBReader<unsigned int> *openCaptured(const std::string &db, int threads);
    // alignment hook: the target keys of entry `id` of that reader without parsing its text (false: not this reader / no keys kept)
    static bool capturedKeys(const DBReader<unsigned int> *reader, size_t id, const unsigned int **keys, size_t *n);

    // ---- the target set, resident once ----
    // With the masking on the device (mmgpu_pf_mask_targets) the prefilter module hands over the UNMASKED SequenceLookup - the very
    // residues the alignment module would map and upload again.  The prefilter hook leaves the lookup here (it takes it away from
    // the Prefiltering object, which would free it) together with the database keys of its ids; the alignment hook asks for it and
    // gets it only if its own reader numbers the same sequences the same way.
    static bool keepsTargets();      // a fused run is in progress: worth handing the lookup over
    static void keepResidentTargets(SequenceLookup *lookup, DBReader<unsigned int> *tdbr, void *gpu);
    // ... or, after a persisted device layout was loaded (MMGpuPrefilterRun::loadPersisted), nothing but the fact that the device
    // holds the database's sequences: residentTargets() then returns the offsets (lengths as the reader gives them) and data = NULL
    static void keepResidentTargetsOnDevice(DBReader<unsigned int> *tdbr, void *gpu);
    static bool residentTargets(DBReader<unsigned int> *tdbr, void *gpu, const unsigned char **data, const uint64_t **offsets);

    // ---- overlapped run: the alignment module works while the prefilter module still produces (MMGpuFusedSearch.cpp, bothModules) ----
    static bool overlappedRun();
    // the lookup was handed over by keepResidentTargets (early, in an overlapped run): the Prefiltering object must not free it
    static bool holdsLookup(const SequenceLookup *lookup);
    // prefilter hook, after a block's entries went through capture(): the alignment module may read them
    static void publish();
    // alignment hook, before it reads the entries of ids [firstId, firstId + count) of the prefilter database
    static void waitForEntries(size_t firstId, size_t count);
};

#endif
