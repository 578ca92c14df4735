"""GPU box helper: the long-sequence drop-in case step by step with the library / integration trace on."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_mmseqs_dropin import _long_case, run, STOCK, MMGPU, THREADS
w = tempfile.mkdtemp()
_long_case(w)
run(STOCK, ["prefilter", "q", "t", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
env = dict(os.environ, MMGPU_TRACE="1")
for extra in ([], ["--threads", "1"]):
    r = subprocess.run([MMGPU, "align", "q", "t", "pref_s", "aln_g%d" % len(extra), "-a", "-v", "3"] + (extra or ["--threads", THREADS]), cwd=w, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print("rc", r.returncode, "\n".join(r.stdout.splitlines()[-25:]))
