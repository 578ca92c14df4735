"""Option sweep for PROFILE query databases (result2profile of a small family-structured set): prefilter / align of the patched
binary against the stock one.  Usage: python scripts/dropin_profile_sweep.py [emu|device]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_mmseqs_dropin import *
EMULATE = (sys.argv[1] if len(sys.argv) > 1 else "emu") == "emu"
w = tempfile.mkdtemp()
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=50, members=20, n_queries=30, seed=21)
wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)
run(STOCK, ["search", "q", "t", "res0", "tmp0", "-s", "5.7", "-a", "--threads", THREADS, "-v", "1"], w)
run(STOCK, ["result2profile", "q", "t", "res0", "prof", "--threads", THREADS, "-v", "1"], w)
run(STOCK, ["prefilter", "prof", "t", "pref_p", "-s", "5.7", "--threads", THREADS, "-v", "1"], w)
bad_total = 0
def one(module, args, in_db, tag):
    global bad_total
    base = [module, "prof", "t"] + in_db
    th = ["--threads", THREADS]
    try:
        run(STOCK, base + ["%s_s" % tag] + args + th + ["-v", "2"], w)
    except AssertionError as e:
        print("%-9s %-50s stock rejects" % (module, " ".join(args))); return
    try:
        log = run(MMGPU, base + ["%s_g" % tag] + args + th + ["-v", "3"], w, EMULATE)
    except AssertionError as e:
        print("%-9s %-50s PATCHED FAILED: %s" % (module, " ".join(args), str(e)[-300:].replace("\n", " | "))); bad_total += 1; return
    why = [l for l in log.split("\n") if "not covered by the device path" in l]
    n, bad, msgs = dbio.diff_dbs(os.path.join(w, "%s_s" % tag), os.path.join(w, "%s_g" % tag))
    print("%-9s %-50s %-40s %s" % (module, " ".join(args), ("CPU path: " + why[0].split("(")[-1][:40]) if why else "device", "identical" if bad == 0 else "%d of %d DIFFER" % (bad, n)))
    bad_total += bad != 0
for i, a in enumerate(([], ["-s", "7.5"], ["-k", "7", "-s", "4"], ["--diag-score", "0"], ["--split", "2", "--split-mode", "0"], ["--comp-bias-corr", "0"], ["--mask", "0"],
                       ["--max-seqs", "5"], ["--exact-kmer-matching", "1"], ["--min-ungapped-score", "40"], ["--spaced-kmer-mode", "0"], ["--pca", "substitution:1.5,context:1.4"])):
    one("prefilter", (["-s", "5.7"] if "-s" not in a else []) + a, [], "pp%d" % i)
for i, a in enumerate((["-a"], ["-a", "--alt-ali", "1"], ["-a", "--realign", "1"], ["--alignment-mode", "3", "--gap-open", "aa:9,nucl:5", "--gap-extend", "aa:2,nucl:2"],
                       ["-a", "-e", "1e-5", "--min-seq-id", "0.2"], ["-a", "--max-accept", "3"], ["-a", "--add-self-matches", "1"], ["-a", "--score-bias", "1"])):
    one("align", a, ["pref_p"], "pa%d" % i)
print("differences / failures:", bad_total)
