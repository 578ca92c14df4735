// See MMGpuFusedSearch.h.  Compiled into MMseqs2 by integration/build_mmseqs.sh (HAVE_MMGPU); called from Search.cpp.
#include "MMGpuFusedSearch.h"

#include <algorithm>
#include <cstring>
#include <sstream>
#include <thread>

#include "Command.h"
#include "Debug.h"
#include "FileUtil.h"
#include "Parameters.h"
#include "SequenceLookup.h"
#include "Timer.h"
#include "Util.h"

#include "MMGpuRun.h"

extern const Command *getCommandByName(const char *s);      // src/commons/Application.cpp

namespace {

struct PrefStore {
    bool on;
    std::string db;                                  // the name the alignment module will ask for
    std::vector<std::vector<char> > perThread;       // serialised entries, appended by the prefilter hook's threads
    struct Entry { unsigned int key; unsigned int thread; size_t offset, length; };
    std::vector<std::vector<Entry> > entries;
    std::vector<char> data;                          // all entries back to back, NUL after each (DBWriter's layout), key order
    PrefStore() : on(false) {}
};
PrefStore store;

struct ResidentTargets {
    SequenceLookup *lookup;
    std::vector<unsigned int> keys;
    void *gpu;
    ResidentTargets() : lookup(NULL), gpu(NULL) {}
};
ResidentTargets resident;

std::vector<std::string> words(const std::string &s) {      // the shell's word splitting of an unquoted $PAR (values with
    std::vector<std::string> w;                             // white space are base64-encoded by createParameterString)
    std::istringstream in(s);
    std::string t;
    while (in >> t) w.push_back(t);
    return w;
}

// a child process starts from Parameters' defaults with no parameter marked as set (parseParameters refuses duplicates)
void freshParameters(Parameters &par, const Command &next) {
    par.setDefaults();
    for (size_t i = 0; i < par.searchworkflow.size(); i++) par.searchworkflow[i]->wasSet = false;
    for (size_t i = 0; i < next.params->size(); i++) (*next.params)[i]->wasSet = false;
}

int module(Parameters &par, const char *name, const std::vector<std::string> &args) {
    const Command *c = getCommandByName(name);
    if (c == NULL) {
        Debug(Debug::ERROR) << "MMGPU: no module " << name << "\n";
        EXIT(EXIT_FAILURE);
    }
    freshParameters(par, *c);
    std::vector<const char *> argv;
    for (size_t i = 0; i < args.size(); i++) argv.push_back(args[i].c_str());
    Timer timer;
    const int status = c->commandFunction((int)argv.size(), argv.data(), *c);
    Debug(Debug::INFO) << "Time for processing: " << timer.lap() << "\n";      // runCommand (Application.cpp:45-50)
    return status;
}

}  // namespace

bool MMGpuFusedSearch::usable(const Parameters &par, bool isUngappedMode, int searchMode, const std::string &program, const std::string &tmpDir) {
    if (!MMGpuRun::enabled()) return false;
    const char *e = getenv("MMGPU_FUSED");
    if (e != NULL && e[0] == '0') return false;
    if (program != tmpDir + "/blastp.sh") return false;      // translated / nucleotide / iterative / sliced searches wrap or replace it
    if (searchMode & (Parameters::SEARCH_MODE_FLAG_QUERY_TRANSLATED | Parameters::SEARCH_MODE_FLAG_TARGET_TRANSLATED)) return false;
    if ((searchMode & Parameters::SEARCH_MODE_FLAG_QUERY_NUCLEOTIDE) && (searchMode & Parameters::SEARCH_MODE_FLAG_TARGET_NUCLEOTIDE)) return false;
    if (par.prefMode != Parameters::PREF_MODE_KMER || par.sensSteps > 1 || isUngappedMode || par.lcaSearch) return false;
    if (par.runner.empty() == false) return false;
#ifdef HAVE_MPI
    return false;
#endif
    return true;
}

int MMGpuFusedSearch::run(Parameters &par, const std::string &query, const std::string &target, const std::string &result,
                          const std::string &tmpDir, const std::string &prefilterPar, const std::string &alignPar, const std::string &sens,
                          bool removeTmp) {
    Debug(Debug::INFO) << "MMGPU: prefilter and align run inside this process (MMGPU_FUSED=0 runs the workflow script)\n";
    // blastp.sh:42-44
    if (FileUtil::fileExists((result + ".dbtype").c_str())) {
        Debug(Debug::ERROR) << result << ".dbtype exists already!\n";
        EXIT(EXIT_FAILURE);
    }
    // the device is opened while the prefilter module parses, opens and masks the databases
    std::thread opener([]() { mmgpu_warmup(MMGpuRun::context()); });      // ... and the kernels' code objects loaded
    const bool onDisk = getenv("MMGPU_FUSED_PREF_ON_DISK") != NULL && getenv("MMGPU_FUSED_PREF_ON_DISK")[0] == '1';
    // Entries kept in memory: the module's DBWriter still creates its (then empty) database, and the alignment module's parameter
    // check wants to see it.  It gets a name blastp.sh does not know, so that a run that died half-way never leaves an empty
    // "pref_0" behind for a later search to resume from (blastp.sh:60), and is removed when the search is done.
    const std::string pref = tmpDir + (onDisk ? "/pref_0" : "/pref_0_mmgpu_in_memory");
    if (!onDisk && FileUtil::fileExists((pref + ".dbtype").c_str())) DBReader<unsigned int>::removeDb(pref);
    int status = EXIT_SUCCESS;
    if (!FileUtil::fileExists((pref + ".dbtype").c_str())) {      // blastp.sh:60 (a re-run after an interrupted search)
        store.on = !onDisk;
        store.db = pref;
        store.perThread.clear();
        store.entries.clear();
        std::vector<std::string> a;
        a.push_back(query); a.push_back(target); a.push_back(pref);
        const std::vector<std::string> p = words(prefilterPar);
        a.insert(a.end(), p.begin(), p.end());
        a.push_back("-s"); a.push_back(sens);
        status = module(par, "prefilter", a);
    }
    opener.join();
    if (status != EXIT_SUCCESS) {
        Debug(Debug::ERROR) << "Prefilter died\n";
        EXIT(EXIT_FAILURE);
    }
    {
        std::vector<std::string> a;
        a.push_back(query); a.push_back(target); a.push_back(pref); a.push_back(result);
        const std::vector<std::string> p = words(alignPar);
        a.insert(a.end(), p.begin(), p.end());
        status = module(par, "align", a);
    }
    store.on = false;
    delete resident.lookup;
    resident.lookup = NULL;
    if (status != EXIT_SUCCESS) {
        Debug(Debug::ERROR) << "Alignment died\n";
        EXIT(EXIT_FAILURE);
    }
    if (removeTmp || !onDisk) {      // blastp.sh:143-158; the in-memory run's placeholder database always goes
        if (FileUtil::fileExists((pref + ".dbtype").c_str())) DBReader<unsigned int>::removeDb(pref);
    }
    return EXIT_SUCCESS;
}

bool MMGpuFusedSearch::capturing(const std::string &db) { return store.on && db == store.db; }

bool MMGpuFusedSearch::keepsTargets() { return store.on; }

void MMGpuFusedSearch::keepResidentTargets(SequenceLookup *lookup, DBReader<unsigned int> *tdbr, void *gpu) {
    delete resident.lookup;
    resident.lookup = lookup;
    resident.gpu = gpu;
    const size_t n = lookup->getSequenceCount();
    resident.keys.resize(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) resident.keys[i] = tdbr->getDbKey(i);
}

bool MMGpuFusedSearch::residentTargets(DBReader<unsigned int> *tdbr, void *gpu, const unsigned char **data, const uint64_t **offsets) {
    if (resident.lookup == NULL || resident.gpu != gpu) return false;
    const size_t n = resident.lookup->getSequenceCount();
    if (tdbr->getSize() != n) return false;
    static_assert(sizeof(size_t) == sizeof(uint64_t), "SequenceLookup::getOffsets() is handed over as it is");
    const uint64_t *off = reinterpret_cast<const uint64_t *>(resident.lookup->getOffsets());
    bool same = true;
#pragma omp parallel for schedule(static) reduction(&& : same)
    for (size_t i = 0; i < n; i++) same = same && tdbr->getDbKey(i) == resident.keys[i] && tdbr->getSeqLen(i) == off[i + 1] - off[i];
    if (!same) return false;
    *data = reinterpret_cast<const unsigned char *>(resident.lookup->getData());
    *offsets = off;
    return true;
}

void MMGpuFusedSearch::capture(unsigned int queryKey, const char *data, size_t len, unsigned int thread) {
    // (called from inside the hook's parallel region: one slot per thread, sized by the hook's first call outside of it)
    std::vector<char> &buf = store.perThread[thread];
    PrefStore::Entry e;
    e.key = queryKey;
    e.thread = thread;
    e.offset = buf.size();
    e.length = len;
    buf.insert(buf.end(), data, data + len);
    buf.push_back('\0');
    store.entries[thread].push_back(e);
}

void mmgpuFusedPrepareCapture(size_t threads) {
    if (store.perThread.size() < threads) {
        store.perThread.resize(threads);
        store.entries.resize(threads);
    }
}

DBReader<unsigned int> *MMGpuFusedSearch::openCaptured(const std::string &db, int threads) {
    if (store.perThread.empty() || db != store.db) return NULL;
    std::vector<PrefStore::Entry> all;
    for (size_t t = 0; t < store.entries.size(); t++) all.insert(all.end(), store.entries[t].begin(), store.entries[t].end());
    std::sort(all.begin(), all.end(), [](const PrefStore::Entry &a, const PrefStore::Entry &b) { return a.key < b.key; });
    size_t total = 0;
    for (size_t i = 0; i < all.size(); i++) total += all[i].length + 1;
    store.data.resize(total + 1);
    // index entries as DBWriter + the index merge leave them: sorted by key, offsets into one data blob, length incl. the NUL
    DBReader<unsigned int>::Index *index = new DBReader<unsigned int>::Index[std::max<size_t>(all.size(), 1)];
    size_t off = 0;
    unsigned int maxLen = 0;
    for (size_t i = 0; i < all.size(); i++) {
        memcpy(store.data.data() + off, store.perThread[all[i].thread].data() + all[i].offset, all[i].length + 1);
        index[i].id = all[i].key;
        index[i].offset = off;
        index[i].length = (unsigned int)(all[i].length + 1);
        maxLen = std::max(maxLen, index[i].length);
        off += all[i].length + 1;
    }
    for (size_t t = 0; t < store.perThread.size(); t++) std::vector<char>().swap(store.perThread[t]);
    const unsigned int lastKey = all.empty() ? 0u : all.back().key;
    DBReader<unsigned int> *r = new DBReader<unsigned int>(index, all.size(), total, lastKey, Parameters::DBTYPE_PREFILTER_RES, maxLen, threads);
    r->open(DBReader<unsigned int>::NOSORT);
    r->setData(store.data.data(), total);
    r->setMode(DBReader<unsigned int>::USE_DATA);
    return r;
}
