// See MMGpuFusedSearch.h.  Compiled into MMseqs2 by integration/build_mmseqs.sh (HAVE_MMGPU); called from Search.cpp.
#include "MMGpuFusedSearch.h"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <iostream>
#include <cstring>
#include <mutex>
#include <sstream>
#include <thread>

#include "Alignment.h"
#include "Command.h"
#include "Debug.h"
#include "FileUtil.h"
#include "Parameters.h"
#include "Prefiltering.h"
#include "SequenceLookup.h"
#include "DBWriter.h"
#include "Timer.h"
#include "Util.h"

#include "MMGpuRun.h"

extern const Command *getCommandByName(const char *s);      // src/commons/Application.cpp

namespace {

struct PrefStore {
    bool on;
    std::string db;                                  // the name the alignment module will ask for
    std::vector<std::vector<char> > perThread;       // serialised entries, appended by the prefilter hook's threads
    struct Entry { unsigned int key; unsigned int thread; size_t offset, length, keyOffset, keyCount; };
    std::vector<std::vector<Entry> > entries;
    std::vector<std::vector<unsigned int> > perThreadKeys;      // the target keys of the entries, beside their text
    std::vector<unsigned int> targetKeys;            // ... of all entries back to back, in the reader's id order
    std::vector<size_t> targetKeyOffsets;            // [entries + 1]
    const DBReader<unsigned int> *reader;            // the DBReader openCaptured handed out
    std::vector<char> data;                          // all entries back to back, NUL after each (DBWriter's layout), key order
    // overlapped run: the alignment module reads while the prefilter module still writes.  Every query has a slot of fixed size
    // (the longest list the prefilter can write for it), so the reader's index is known before the first entry exists.
    bool progressive;
    std::vector<unsigned int> keys;                  // query keys, ascending = the ids of the prefilter database
    size_t slotBytes;
    char *slots;                                     // keys.size() * slotBytes, anonymous mapping (untouched slots cost nothing)
    size_t mapped;
    std::vector<char> produced;                      // per id
    size_t producedCount;
    bool producerDone;                               // the prefilter module returned
    PrefStore() : on(false), reader(NULL), progressive(false), slotBytes(0), slots(NULL), mapped(0), producedCount(0), producerDone(false) {}
};
PrefStore store;

struct ResidentTargets {
    SequenceLookup *lookup;
    std::vector<unsigned int> keys;
    std::vector<uint64_t> offsets;                   // device-only hand-over (persisted layout): no lookup, the lengths as the reader gave them
    bool deviceOnly;
    void *gpu;
    bool decided;                                    // overlapped run: the prefilter hook has offered the lookup, or never will
    ResidentTargets() : lookup(NULL), deviceOnly(false), gpu(NULL), decided(false) {}
};
ResidentTargets resident;

// producer = the prefilter module's thread(s), consumer = the alignment module's thread of an overlapped run
std::mutex fusedLock;
std::condition_variable fusedChanged;
bool overlapped = false;

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


void closeProgressive() {
    if (store.slots != NULL) munmap(store.slots, store.mapped);
    store.slots = NULL;
    store.progressive = false;
    overlapped = false;
    std::vector<unsigned int>().swap(store.keys);
    std::vector<char>().swap(store.produced);
}

// The two modules of blastp.sh side by side: what prefilter() (src/prefiltering/Main.cpp:13-62) and align() (src/alignment/
// Main.cpp:12-31) do, with the alignment module started as soon as the Prefiltering object exists.  Its reader of the prefilter
// result (MMGpuFusedSearch::openCaptured) has one slot per query; its hook waits for the resident targets the prefilter hook
// leaves behind after its hand-over (MMGpuAlignRun::begin) and, bucket by bucket, for the slots of the bucket's queries
// (MMGpuAlignRun::plan) - the device aligns the first hit lists while it still prefilters the later queries.
// The Parameters singleton is read by the two constructors only (Prefiltering.cpp:20-260, Alignment.cpp:20-215) and is parsed for
// the alignment module once the Prefiltering object is complete.  Not overlapped (the prefilter runs to its end, *aligned stays
// false and the caller starts the alignment module as a whole): split prefilter runs, configurations the prefilter hook leaves to
// the CPU loop or runs without the resident-target hand-over, more slots than a quarter of the memory.
int bothModules(Parameters &par, const std::vector<std::string> &prefArgs, const std::vector<std::string> &alnArgs, bool sideBySide, bool *aligned) {
    const Command *pc = getCommandByName("prefilter"), *ac = getCommandByName("align");
    if (pc == NULL || ac == NULL) {
        Debug(Debug::ERROR) << "MMGPU: no prefilter / align module\n";
        EXIT(EXIT_FAILURE);
    }
    // (index databases, profiles, nucleotides: the module's own checks and paths, Main.cpp:22-46)
    const int queryDbType = FileUtil::parseDbType(prefArgs[0].c_str());
    const int targetDbType = FileUtil::parseDbType(prefArgs[1].c_str());
    if (!Parameters::isEqualDbtype(queryDbType, Parameters::DBTYPE_AMINO_ACIDS) || !Parameters::isEqualDbtype(targetDbType, Parameters::DBTYPE_AMINO_ACIDS))
        return module(par, "prefilter", prefArgs);
    Timer prefTimer;
    freshParameters(par, *pc);
    std::vector<const char *> argv;
    for (size_t i = 0; i < prefArgs.size(); i++) argv.push_back(prefArgs[i].c_str());
    par.parseParameters((int)argv.size(), argv.data(), *pc, true, 0, MMseqsParameter::COMMAND_PREFILTER);
    const std::string prefDb = par.db3, prefDbIndex = par.db3Index, queryDb = par.db1, queryDbIndex = par.db1Index;
    const bool taxonFilter = par.taxonList.length() > 0;
    Prefiltering *pref = new Prefiltering(par.db1, par.db1Index, par.db2, par.db2Index, queryDbType, targetDbType, par);
    size_t maxResListLen = 0;
    bool overlap = sideBySide && !taxonFilter && MMGpuPrefilterRun::runsUnsplitWithResidentTargets(*pref, &maxResListLen);
    if (overlap) {
        // one slot per query, ids = ascending keys (the order of a DBReader's index)
        DBReader<unsigned int> qr(queryDb.c_str(), queryDbIndex.c_str(), 1, DBReader<unsigned int>::USE_INDEX);
        qr.open(DBReader<unsigned int>::NOSORT);
        store.keys.resize(qr.getSize());
        for (size_t i = 0; i < qr.getSize(); i++) store.keys[i] = qr.getDbKey(i);
        qr.close();
        if (!std::is_sorted(store.keys.begin(), store.keys.end())) std::sort(store.keys.begin(), store.keys.end());
        // a line of QueryMatcher::prefilterHitToBuffer: key, score, diagonal as decimal numbers (10 + 11 + 11 characters at most), two tabs, newline
        store.slotBytes = maxResListLen * 36 + 1;
        store.mapped = std::max<size_t>(store.keys.size(), 1) * store.slotBytes;
        overlap = store.mapped <= Util::getTotalSystemMemory() / 4;
        if (overlap) {
            void *m = mmap(NULL, store.mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            overlap = m != MAP_FAILED;
            if (overlap) store.slots = static_cast<char *>(m);
        }
    }
    if (!overlap) {
        std::vector<unsigned int>().swap(store.keys);
        pref->runAllSplits(prefDb, prefDbIndex);
        Debug(Debug::INFO) << "Time for processing: " << prefTimer.lap() << "\n";
        // The object is NOT taken apart while the alignment module starts (round 6): unmapping its readers and freeing its tables on a
        // helper thread held the address-space lock for ~0.15 s in pieces, and the alignment module's first steps - its writer's 32
        // buffers of 32 MB, its readers' mappings - waited on exactly that lock (0.17 s between "Calculation of alignments" and the
        // hook's first lap at 1 M targets).  The search process ends through _exit (run(), below): the operating system takes the
        // object back with everything else.  MMGPU_FUSED_UNWIND=1 (leak checkers) deletes it, in line.
        {
            const char *unwind = getenv("MMGPU_FUSED_UNWIND");
            if (unwind != NULL && unwind[0] == '1') delete pref;
        }
        return EXIT_SUCCESS;
    }
    Debug(Debug::INFO) << "MMGPU: the alignment module starts while the prefilter module runs (MMGPU_FUSED_OVERLAP=1)\n";
    store.produced.assign(store.keys.size(), 0);
    store.producedCount = 0;
    store.producerDone = false;
    store.progressive = true;
    resident.decided = false;
    overlapped = true;
    {   // the alignment module's parameter check wants to see its input database: an empty one until the prefilter module's
        // DBWriter replaces it (the entries themselves never reach it)
        DBWriter placeholder(prefDb.c_str(), prefDbIndex.c_str(), 1, 0, Parameters::DBTYPE_PREFILTER_RES);
        placeholder.open();
        placeholder.close();
    }
    Timer alnTimer;
    freshParameters(par, *ac);
    std::vector<const char *> alnArgv;
    for (size_t i = 0; i < alnArgs.size(); i++) alnArgv.push_back(alnArgs[i].c_str());
    par.overrideParameterDescription(par.PARAM_ALIGNMENT_MODE, "How to compute the alignment:\n0: automatic\n1: only score and end_pos\n2: also start_pos and cov\n3: also seq.id", NULL, 0);
    par.parseParameters((int)alnArgv.size(), alnArgv.data(), *ac, true, 0, MMseqsParameter::COMMAND_ALIGN);
    std::thread aligner([&par, &alnTimer]() {
        {
            Alignment aln(par.db1, par.db2, par.db3, par.db3Index, par.db4, par.db4Index, par, false);
            Debug(Debug::INFO) << "Calculation of alignments\n";
            aln.run();
        }
        Debug(Debug::INFO) << "Time for processing: " << alnTimer.lap() << "\n";
    });
    pref->runAllSplits(prefDb, prefDbIndex);
    {
        std::lock_guard<std::mutex> guard(fusedLock);
        store.producerDone = true;
    }
    fusedChanged.notify_all();
    delete pref;
    Debug(Debug::INFO) << "Time for processing: " << prefTimer.lap() << "\n";
    aligner.join();
    *aligned = true;
    return EXIT_SUCCESS;
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
    // ... and the kernels' code objects loaded
    std::thread opener([]() { mmgpu_warmup(MMGpuRun::context()); });
    const bool onDisk = getenv("MMGPU_FUSED_PREF_ON_DISK") != NULL && getenv("MMGPU_FUSED_PREF_ON_DISK")[0] == '1';
    // Entries kept in memory: the module's DBWriter still creates its (then empty) database, and the alignment module's parameter
    // check wants to see it.  It gets a name blastp.sh does not know, so that a run that died half-way never leaves an empty
    // "pref_0" behind for a later search to resume from (blastp.sh:60), and is removed when the search is done.
    const std::string pref = tmpDir + (onDisk ? "/pref_0" : "/pref_0_mmgpu_in_memory");
    if (!onDisk && FileUtil::fileExists((pref + ".dbtype").c_str())) DBReader<unsigned int>::removeDb(pref);
    int status = EXIT_SUCCESS;
    bool aligned = false;
    if (!FileUtil::fileExists((pref + ".dbtype").c_str())) {      // blastp.sh:60 (a re-run after an interrupted search)
        store.on = !onDisk;
        store.db = pref;
        store.perThread.clear();
        store.perThreadKeys.clear();
        store.entries.clear();
        store.reader = NULL;
        std::vector<std::string> a;
        a.push_back(query); a.push_back(target); a.push_back(pref);
        const std::vector<std::string> p = words(prefilterPar);
        a.insert(a.end(), p.begin(), p.end());
        a.push_back("-s"); a.push_back(sens);
        // the prefilter module as prefilter() runs it (src/prefiltering/Main.cpp:13-62), the object built and released here
        // (bothModules).  MMGPU_FUSED_OVERLAP=1 also starts the alignment module beside it.  That is off by default: at 10 000 x 1 M
        // the alignment hook's per-bucket costs and the contention of the two modules for the host threads outweigh the overlap
        // (1.86 - 2.11 s for buckets of 5120 ... 1024 queries against 1.77 s one after the other,
        // profiles/r04_fused_overlap_variants.json).
        const char *e = getenv("MMGPU_FUSED_OVERLAP");
        if (!onDisk) {
            std::vector<std::string> b;
            b.push_back(query); b.push_back(target); b.push_back(pref); b.push_back(result);
            const std::vector<std::string> q = words(alignPar);
            b.insert(b.end(), q.begin(), q.end());
            status = bothModules(par, a, b, e != NULL && e[0] == '1', &aligned);
        } else {
            status = module(par, "prefilter", a);
        }
    }
    opener.join();
    if (status != EXIT_SUCCESS) {
        Debug(Debug::ERROR) << (aligned ? "Alignment died\n" : "Prefilter died\n");
        EXIT(EXIT_FAILURE);
    }
    if (!aligned) {
        std::vector<std::string> a;
        a.push_back(query); a.push_back(target); a.push_back(pref); a.push_back(result);
        const std::vector<std::string> p = words(alignPar);
        a.insert(a.end(), p.begin(), p.end());
        status = module(par, "align", a);
    }
    closeProgressive();
    store.on = false;
    delete resident.lookup;
    resident.lookup = NULL;
    resident.deviceOnly = false;
    if (status != EXIT_SUCCESS) {
        Debug(Debug::ERROR) << "Alignment died\n";
        EXIT(EXIT_FAILURE);
    }
    if (removeTmp || !onDisk) {      // blastp.sh:143-158; the in-memory run's placeholder database always goes
        if (FileUtil::fileExists((pref + ".dbtype").c_str())) DBReader<unsigned int>::removeDb(pref);
    }
    // The stock `search` ends in execProgram(blastp.sh) and never returns to its caller (Search.cpp:618-621).  Returning from here
    // would unwind a process that holds ~70 GB of device mappings and the host copies of both databases: 0.2 s of runtime and
    // allocator teardown after the last result is on disk (profiles/r04_search_timeline.txt).  All writers are closed; the operating
    // system takes the rest back.  MMGPU_FUSED_UNWIND=1 returns normally (leak checkers).
    const char *unwind = getenv("MMGPU_FUSED_UNWIND");
    if (!(unwind != NULL && unwind[0] == '1')) {
        std::cout.flush();
        std::cerr.flush();
        fflush(NULL);
        _exit(EXIT_SUCCESS);
    }
    return EXIT_SUCCESS;
}

bool MMGpuFusedSearch::capturing(const std::string &db) { return store.on && db == store.db; }

bool MMGpuFusedSearch::keepsTargets() { return store.on; }

void MMGpuFusedSearch::keepResidentTargets(SequenceLookup *lookup, DBReader<unsigned int> *tdbr, void *gpu) {
    if (lookup != NULL) {
        const size_t n = lookup->getSequenceCount();
        std::vector<unsigned int> keys(n);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) keys[i] = tdbr->getDbKey(i);
        std::lock_guard<std::mutex> guard(fusedLock);
        delete resident.lookup;
        resident.lookup = lookup;
        resident.gpu = gpu;
        resident.keys.swap(keys);
        resident.decided = true;
    } else {      // (overlapped run: nothing will be handed over - the alignment module maps and uploads its targets itself)
        std::lock_guard<std::mutex> guard(fusedLock);
        resident.decided = true;
    }
    fusedChanged.notify_all();
}

void MMGpuFusedSearch::keepResidentTargetsOnDevice(DBReader<unsigned int> *tdbr, void *gpu) {
    const size_t n = tdbr->getSize();
    std::vector<unsigned int> keys(n);
    std::vector<uint64_t> offsets(n + 1, 0);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        keys[i] = tdbr->getDbKey(i);
        offsets[i + 1] = tdbr->getSeqLen(i);
    }
    for (size_t i = 0; i < n; i++) offsets[i + 1] += offsets[i];
    {
        std::lock_guard<std::mutex> guard(fusedLock);
        delete resident.lookup;
        resident.lookup = NULL;
        resident.deviceOnly = true;
        resident.gpu = gpu;
        resident.keys.swap(keys);
        resident.offsets.swap(offsets);
        resident.decided = true;
    }
    fusedChanged.notify_all();
}

bool MMGpuFusedSearch::overlappedRun() { return overlapped; }

bool MMGpuFusedSearch::holdsLookup(const SequenceLookup *lookup) { return lookup != NULL && lookup == resident.lookup; }

void MMGpuFusedSearch::publish() {
    if (!store.progressive) return;
    {
        std::lock_guard<std::mutex> guard(fusedLock);
        store.producedCount++;
    }
    fusedChanged.notify_all();
}

void MMGpuFusedSearch::waitForEntries(size_t firstId, size_t count) {
    if (!store.progressive) return;
    const size_t end = std::min(firstId + count, store.produced.size());
    std::unique_lock<std::mutex> guard(fusedLock);
    size_t next = firstId;
    for (;;) {
        while (next < end && store.produced[next]) next++;
        if (next >= end) return;
        if (store.producerDone) {
            // the prefilter module writes an entry for every query (Prefiltering.cpp:876-917): a missing one is an error of the hook
            Debug(Debug::ERROR) << "MMGPU: the prefilter module ended without an entry for query id " << next << "\n";
            EXIT(EXIT_FAILURE);
        }
        fusedChanged.wait(guard);
    }
}

bool MMGpuFusedSearch::residentTargets(DBReader<unsigned int> *tdbr, void *gpu, const unsigned char **data, const uint64_t **offsets) {
    if (overlapped) {      // the prefilter hook offers the lookup right after its hand-over to the device
        std::unique_lock<std::mutex> guard(fusedLock);
        fusedChanged.wait(guard, []() { return resident.decided || store.producerDone; });
    }
    if ((resident.lookup == NULL && !resident.deviceOnly) || resident.gpu != gpu) return false;
    // (device only: single sequences are mapped on demand by the alignment hook, through the reader's first thread slot - an
    // uncompressed database has no per-thread buffers behind getData)
    if (resident.deviceOnly && tdbr->isCompressed()) return false;
    const size_t n = resident.deviceOnly ? resident.keys.size() : resident.lookup->getSequenceCount();
    if (tdbr->getSize() != n) return false;
    static_assert(sizeof(size_t) == sizeof(uint64_t), "SequenceLookup::getOffsets() is handed over as it is");
    const uint64_t *off = resident.deviceOnly ? resident.offsets.data() : reinterpret_cast<const uint64_t *>(resident.lookup->getOffsets());
    bool same = true;
#pragma omp parallel for schedule(static) reduction(&& : same)
    for (size_t i = 0; i < n; i++) same = same && tdbr->getDbKey(i) == resident.keys[i] && tdbr->getSeqLen(i) == off[i + 1] - off[i];
    if (!same) return false;
    *data = resident.deviceOnly ? NULL : reinterpret_cast<const unsigned char *>(resident.lookup->getData());
    *offsets = off;
    return true;
}

void MMGpuFusedSearch::capture(unsigned int queryKey, const char *data, size_t len, unsigned int thread, const unsigned int *targetKeys,
                               size_t nTargets) {
    if (store.progressive) {      // straight into the query's slot; MMGpuFusedSearch::publish() makes it visible to the reader
        const size_t id = std::lower_bound(store.keys.begin(), store.keys.end(), queryKey) - store.keys.begin();
        if (id >= store.keys.size() || store.keys[id] != queryKey || len + 1 > store.slotBytes) {
            Debug(Debug::ERROR) << "MMGPU: prefilter entry of query " << queryKey << " (" << len << " bytes) has no slot\n";
            EXIT(EXIT_FAILURE);
        }
        char *slot = store.slots + id * store.slotBytes;
        memcpy(slot, data, len);
        slot[len] = '\0';
        store.produced[id] = 1;
        return;
    }
    // (called from inside the hook's parallel region: one slot per thread, sized by the hook's first call outside of it)
    std::vector<char> &buf = store.perThread[thread];
    PrefStore::Entry e;
    e.key = queryKey;
    e.thread = thread;
    e.offset = buf.size();
    e.length = len;
    buf.insert(buf.end(), data, data + len);
    buf.push_back('\0');
    std::vector<unsigned int> &kb = store.perThreadKeys[thread];
    e.keyOffset = kb.size();
    e.keyCount = nTargets;
    kb.insert(kb.end(), targetKeys, targetKeys + nTargets);
    store.entries[thread].push_back(e);
}

bool MMGpuFusedSearch::capturedKeys(const DBReader<unsigned int> *reader, size_t id, const unsigned int **keys, size_t *n) {
    if (reader == NULL || reader != store.reader || store.progressive || id + 1 >= store.targetKeyOffsets.size()) return false;
    *keys = store.targetKeys.data() + store.targetKeyOffsets[id];
    *n = store.targetKeyOffsets[id + 1] - store.targetKeyOffsets[id];
    return true;
}

void mmgpuFusedPrepareCapture(size_t threads) {
    if (store.perThread.size() < threads) {
        store.perThread.resize(threads);
        store.perThreadKeys.resize(threads);
        store.entries.resize(threads);
    }
}

DBReader<unsigned int> *MMGpuFusedSearch::openCaptured(const std::string &db, int threads) {
    if (store.progressive && db == store.db) {
        const size_t n = store.keys.size();
        DBReader<unsigned int>::Index *index = new DBReader<unsigned int>::Index[std::max<size_t>(n, 1)];
        for (size_t i = 0; i < n; i++) {
            index[i].id = store.keys[i];
            index[i].offset = i * store.slotBytes;
            index[i].length = (unsigned int)store.slotBytes;
        }
        DBReader<unsigned int> *r = new DBReader<unsigned int>(index, n, store.mapped, n ? store.keys[n - 1] : 0u, Parameters::DBTYPE_PREFILTER_RES,
                                                               (unsigned int)store.slotBytes, threads);
        r->open(DBReader<unsigned int>::NOSORT);
        r->setData(store.slots, store.mapped);
        r->setMode(DBReader<unsigned int>::USE_DATA);
        return r;
    }
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
    // the same entries as lists of target keys (MMGpuAlignRun::plan builds its target lists from them instead of parsing the text)
    store.targetKeyOffsets.assign(all.size() + 1, 0);
    for (size_t i = 0; i < all.size(); i++) store.targetKeyOffsets[i + 1] = store.targetKeyOffsets[i] + all[i].keyCount;
    store.targetKeys.resize(store.targetKeyOffsets[all.size()]);
    for (size_t i = 0; i < all.size(); i++)
        if (all[i].keyCount)
            memcpy(store.targetKeys.data() + store.targetKeyOffsets[i], store.perThreadKeys[all[i].thread].data() + all[i].keyOffset,
                   all[i].keyCount * sizeof(unsigned int));
    for (size_t t = 0; t < store.perThread.size(); t++) {
        std::vector<char>().swap(store.perThread[t]);
        std::vector<unsigned int>().swap(store.perThreadKeys[t]);
    }
    const unsigned int lastKey = all.empty() ? 0u : all.back().key;
    DBReader<unsigned int> *r = new DBReader<unsigned int>(index, all.size(), total, lastKey, Parameters::DBTYPE_PREFILTER_RES, maxLen, threads);
    r->open(DBReader<unsigned int>::NOSORT);
    r->setData(store.data.data(), total);
    r->setMode(DBReader<unsigned int>::USE_DATA);
    store.reader = r;
    return r;
}
