"""Sweep of whole workflows (search / cluster / rbh / map with options) through the patched binary against the stock one on the
example proteins: the workflow scripts call prefilter / align of the same binary, so every module in between sees the
device's output.  Usage: python scripts/dropin_workflow_sweep.py [emu|device]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import dbio                                                   # noqa: E402
from tests.test_mmseqs_dropin import STOCK, MMGPU, EXAMPLES, THREADS, run, copy_db, reference_for   # noqa: E402

emulate = (sys.argv[1] if len(sys.argv) > 1 else "emu") == "emu"
FLOWS = [
    ("search", ["-s", "5.7", "-a"]),
    ("search", ["--num-iterations", "2", "-s", "4"]),
    ("search", ["--start-sens", "2", "-s", "6", "--sens-steps", "3"]),
    ("search", ["-s", "4", "--alignment-mode", "3", "--max-seqs", "20", "--max-accept", "5"]),
    ("search", ["-s", "4", "-a", "--alt-ali", "2"]),
    ("search", ["-s", "4", "-e", "1e-10", "--min-seq-id", "0.3", "-c", "0.8"]),
    ("search", ["-s", "4", "--exhaustive-search", "1"]),
    ("search", ["-s", "4", "--realign", "1", "-a"]),
    ("search", ["-s", "4", "--diag-score", "0"]),
    ("rbh", ["-s", "4"]),
    ("map", []),
    ("cluster", ["--min-seq-id", "0.3", "-s", "4"]),
    ("cluster", ["--cluster-reassign", "1", "-s", "4"]),
]
w = tempfile.mkdtemp()
copy_db(EXAMPLES, os.path.join(w, "q"))
bad_total = 0
for i, (flow, args) in enumerate(FLOWS):
    th = ["--threads", THREADS]
    if flow == "cluster":
        base_s, base_g = ["cluster", "q", "res_s%d" % i, "tmp_s%d" % i], ["cluster", "q", "res_g%d" % i, "tmp_g%d" % i]
    else:
        base_s, base_g = [flow, "q", "q", "res_s%d" % i, "tmp_s%d" % i], [flow, "q", "q", "res_g%d" % i, "tmp_g%d" % i]
    # (workflows with second alignments of accepted hits - iterations, --realign, --alt-ali, cluster - are compared with the stock tree
    # that carries the same do-nothing block-aligner stubs as the patched test binary: tests/test_mmseqs_dropin.py::reference_for)
    ref, env = reference_for([flow] + args)
    try:
        run(ref, base_s + args + th + ["-v", "2"], w)
    except AssertionError as e:
        print("%-8s %-55s stock binary fails: %s" % (flow, " ".join(args), str(e)[-160:].replace("\n", " | ")))
        continue
    try:
        log = run(MMGPU, base_g + args + th + ["-v", "3"], w, emulate, extra_env=env)
    except AssertionError as e:
        print("%-8s %-55s PATCHED BINARY FAILED: %s" % (flow, " ".join(args), str(e)[-400:].replace("\n", " | ")))
        bad_total += 1
        continue
    n, bad, msgs = dbio.diff_dbs(os.path.join(w, "res_s%d" % i), os.path.join(w, "res_g%d" % i))
    print("%-8s %-55s device calls %2d, CPU-path notes %2d   %s" % (flow, " ".join(args), log.count("MMGPU: device"),
                                                                   log.count("using the CPU path"), "identical (%d entries)" % n if bad == 0 else "%d of %d entries DIFFER %s" % (bad, n, msgs[:1])))
    bad_total += bad != 0
print("workflows with differences or failures:", bad_total)
sys.exit(1 if bad_total else 0)
