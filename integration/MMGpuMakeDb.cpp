// `mmseqs makemmgpudb <i:targetDB> <o:layoutFile> [prefilter options]`: the persisted device layout (SURVEY.md section 8 row f1;
// include/mmgpu.h mmgpu_db_save) of a target database as a command of its own - the counterpart of the reference's makepaddedseqdb
// (src/util/makepaddedseqdb.cpp:14: sequences in the GPU server's layout) and createindex (PrefilteringIndexReader.cpp:52: index table
// + masked lookup) in ONE file holding what the device holds: targets, tantan-masked view, k-mer index.  A later
// `MMGPU_DB_FILE=<layoutFile> mmseqs search ...` with the same prefilter options (sensitivity / k-mer size / masking: they decide the
// index's k-mer threshold and masking, and are part of the file's index fingerprint) loads it before a single sequence is mapped.
// The options are the prefilter module's own, parsed by its own parser: what they mean for the index is decided by the reference's
// Prefiltering constructor, exactly as in the search that will use the file.
#include "Prefiltering.h"
#include "Parameters.h"
#include "FileUtil.h"
#include "Debug.h"
#include "Timer.h"
#include "Util.h"

#include "MMGpuRun.h"

#include <cstdlib>

int makemmgpudb(int argc, const char **argv, const Command &command) {
    Parameters &par = Parameters::getInstance();
    par.parseParameters(argc, argv, command, true, 0, MMseqsParameter::COMMAND_PREFILTER);
    Timer timer;
    if (!MMGpuRun::enabled()) {
        Debug(Debug::ERROR) << "MMGPU: no device path in this process (MMGPU_DISABLE is set)\n";
        return EXIT_FAILURE;
    }
    const int targetDbType = FileUtil::parseDbType(par.db1.c_str());
    if (targetDbType == -1) {
        Debug(Debug::ERROR) << "Please recreate your database or add a .dbtype file to your sequence/profile database.\n";
        return EXIT_FAILURE;
    }
    if (!Parameters::isEqualDbtype(targetDbType, Parameters::DBTYPE_AMINO_ACIDS)) {
        // nucleotide databases: the 4^15 offsets of their index are 4.3 GB - reading them back costs what the device needs to build
        // them (0.35 s at 111 M nucleotides); profile databases: the index is the host's (similar k-mers of the profiles)
        Debug(Debug::ERROR) << "MMGPU: makemmgpudb persists amino-acid sequence databases\n";
        return EXIT_FAILURE;
    }
    setenv("MMGPU_DB_FILE", par.db2.c_str(), 1);
    // the database on both sides: the k-mer threshold of the index follows the sensitivity for sequence queries
    Prefiltering pref(par.db1, par.db1Index, par.db1, par.db1Index, targetDbType, targetDbType, par);
    const bool ok = MMGpuPrefilterRun::buildAndSave(pref);
    Debug(Debug::INFO) << "Time for processing: " << timer.lap() << "\n";
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
