"""GPU box: wall time of `mmseqs prefilter` and `mmseqs align -a` through the stock and the patched binary on
BASELINE.json configs[2] at 1/10 scale (1000 queries x 100 000 targets), all host threads.  Prints one JSON object."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl, dbio
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
threads = str(os.cpu_count() or 1)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = tempfile.mkdtemp(prefix="mmgpu_timing_")
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=2000 * scale, members=50, n_queries=1000 * scale, seed=10)
wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
def run(b, args):
    t0 = time.perf_counter()
    r = subprocess.run([b] + args, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    return time.perf_counter() - t0, r.stdout
run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"]); run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"])
out = {"workload": "%d queries x %d targets, -s 5.7, --max-seqs 300, %s threads" % (1000 * scale, 100000 * scale, threads)}
for name, b in (("stock", STOCK), ("patched", MMGPU)):
    tp, log = run(b, ["prefilter", "q", "t", "pref_" + name, "-s", "5.7", "--threads", threads, "-v", "3"])
    ta, _ = run(b, ["align", "q", "t", "pref_" + name, "aln_" + name, "-a", "--threads", threads, "-v", "3"])
    proc = [l for l in log.splitlines() if "Time for processing" in l or "Index table: fill" in l or "Time for index table init" in l]
    out[name] = {"prefilter_wall_s": round(tp, 2), "align_wall_s": round(ta, 2), "prefilter_log": proc[-3:]}
n, bad, _ = dbio.diff_dbs(os.path.join(w, "aln_stock"), os.path.join(w, "aln_patched"))
out["alignment_dbs_identical"] = bad == 0
out["entries"] = n
print(json.dumps(out, indent=1))
