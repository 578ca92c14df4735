"""Sweep of prefilter / align options through the patched binary against the stock one (example proteins).  For every
option set: did the hook run (or say why it kept the CPU path), and do the result DBs agree byte for byte.
Usage: python scripts/dropin_option_sweep.py [emu|device]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import dbio                                                   # noqa: E402
from tests.test_mmseqs_dropin import STOCK, MMGPU, EXAMPLES, THREADS, run, copy_db, reference_for   # noqa: E402

emulate = (sys.argv[1] if len(sys.argv) > 1 else "emu") == "emu"
PREF = [
    ["-s", "7.5"], ["-s", "1"], ["-s", "4", "-k", "7"], ["-k", "5"], ["--max-seqs", "1000"], ["--max-seqs", "5"],
    ["--comp-bias-corr", "0"], ["--comp-bias-corr-scale", "0.5"], ["--mask", "0"], ["--mask-lower-case", "1"],
    ["--mask-prob", "0.5"], ["--spaced-kmer-mode", "0"], ["--exact-kmer-matching", "1"], ["-c", "0.8", "--cov-mode", "0"],
    ["-c", "0.5", "--cov-mode", "1"], ["--min-ungapped-score", "30"], ["--k-score", "seq:100,prof:80"],
    ["--seed-sub-mat", "aa:VTML40.out,nucl:nucleotide.out"], ["--alph-size", "aa:13,nucl:5"], ["--threads", "1"],
    ["--spaced-kmer-pattern", "1101011"], ["--max-seq-len", "300"], ["--split-memory-limit", "1G"], ["--diag-score", "0"],
    ["--min-ungapped-score", "0"], ["--target-search-mode", "1"],
]
ALIGN = [
    ["-a"], ["--alignment-mode", "3", "-e", "1e-5"], ["-a", "--alt-ali", "2"], ["-a", "--realign", "1"],
    ["--alignment-mode", "3", "--gap-open", "aa:9,nucl:5", "--gap-extend", "aa:2,nucl:2"], ["-a", "--seq-id-mode", "1"],
    ["-a", "--seq-id-mode", "2"], ["-a", "--min-aln-len", "50"], ["-a", "--corr-score-weight", "0.5"],
    ["--alignment-mode", "3", "--score-bias", "2"], ["-a", "--sub-mat", "aa:blosum45.out,nucl:nucleotide.out"],
    ["-a", "--comp-bias-corr-scale", "0.3"], ["-a", "--max-seq-len", "300"], ["-a", "--alignment-output-mode", "1"],
    ["-a", "--wrapped-scoring", "1"], ["-a", "-c", "0.7", "--cov-mode", "1"], ["-a", "--min-seq-id", "0.3", "--alignment-mode", "3"],
    ["--alignment-mode", "4"], ["-a", "--threads", "1"], ["-a", "--max-accept", "5"], ["-a", "--max-rejected", "3"],
]
w = tempfile.mkdtemp()
copy_db(EXAMPLES, os.path.join(w, "q"))
bad_total = 0


def one(module, args, in_db, tag):
    global bad_total
    base = [module, "q", "q"] + in_db
    th = [] if "--threads" in args else ["--threads", THREADS]
    # second alignments of accepted hits (--realign, --alt-ali) are the reference loop's own calls into the block-aligner crate: the
    # patched test binary links do-nothing stubs there (nothing of oracle/ on the product side), so its partner for those options is
    # the stock tree with the same stubs, and first-pass int16-range pairs take the reference's fallback too (tests/test_mmseqs_dropin.py)
    ref, env = reference_for(args)
    try:
        run(ref, base + ["%s_s" % tag] + args + th + ["-v", "2"], w)
    except AssertionError as e:
        print("%-9s %-60s stock binary rejects the options" % (module, " ".join(args)))
        return
    try:
        log = run(MMGPU, base + ["%s_g" % tag] + args + th + ["-v", "3"], w, emulate, extra_env=env)
    except AssertionError as e:
        print("%-9s %-60s PATCHED BINARY FAILED: %s" % (module, " ".join(args), str(e)[-300:].replace("\n", " | ")))
        bad_total += 1
        return
    why = [l for l in log.split("\n") if "not covered by the device path" in l]
    n, bad, msgs = dbio.diff_dbs(os.path.join(w, "%s_s" % tag), os.path.join(w, "%s_g" % tag))
    path = "CPU path (%s)" % why[0].split("(")[-1].split(")")[0] if why else ("device" if "MMGPU: device" in log else "CPU path (silent)")
    print("%-9s %-60s %-55s %s" % (module, " ".join(args), path, "identical" if bad == 0 else "%d of %d entries DIFFER" % (bad, n)))
    bad_total += bad != 0


for i, a in enumerate(PREF):
    one("prefilter", ["-s", "5.7"] + a if "-s" not in a else a, [], "p%d" % i)
run(STOCK, ["prefilter", "q", "q", "pref", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
for i, a in enumerate(ALIGN):
    one("align", a, ["pref"], "a%d" % i)
print("option sets with differences or failures:", bad_total)
sys.exit(1 if bad_total else 0)
