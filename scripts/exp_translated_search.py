"""GPU box: BASELINE.json configs[4] as a translated search through the binaries - `mmseqs search reads contigs res tmp
--search-type 2` (extractorfs -> translatenucs -> prefilter -> align -> offsetalignment, data/workflow/translated_search.sh):
wall time of the patched binary at the given sizes, of the stock binary where asked for, and whether the two result databases
are the same.   python scripts/exp_translated_search.py <contigs> <reads> [--stock]"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl, dbio
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
n_contigs, n_reads = int(sys.argv[1]), int(sys.argv[2])
threads = os.environ.get("MMGPU_BENCH_THREADS", "32")
w = tempfile.mkdtemp(prefix="mmgpu_translated_")
t0 = time.perf_counter()
queries, (tres, toff), _ = wl.config5_nucleotide(n_contigs, n_reads, 10000, seed=20)
wl.write_nucl_fasta(os.path.join(w, "contigs.fasta"), wl.split(tres, toff), "c")
wl.write_nucl_fasta(os.path.join(w, "reads.fasta"), queries, "r")
out = {"contigs": n_contigs, "reads": n_reads, "contig_nt": int(toff[-1]), "generate_s": round(time.perf_counter() - t0, 1)}


def run(b, a, env_extra=None):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "mmseqs2_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    if env_extra:
        env.update(env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([b] + a, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    return time.perf_counter() - t0, r.stdout


t, _ = run(STOCK, ["createdb", "contigs.fasta", "contigs", "-v", "1"])
out["createdb_contigs_s"] = round(t, 1)
run(STOCK, ["createdb", "reads.fasta", "reads", "-v", "1"])
args = ["--search-type", "2", "--threads", threads, "-v", "3"]
t, log = run(MMGPU, ["search", "reads", "contigs", "res_g", "tmp_g"] + args, {"MMGPU_TRACE": "1"})
out["patched_wall_s"] = round(t, 2)
sys.stderr.write("\n".join(l for l in log.splitlines() if "[mmgpu" in l or "Time for" in l or "MMGPU" in l or l.startswith(("extractorfs", "translatenucs", "prefilter", "align", "offsetalignment", "search", "swapresults"))) + "\n")
out["modules_on_device"] = log.count("MMGPU: device")
out["cpu_path_messages"] = [l for l in log.splitlines() if "using the CPU path" in l][:5]
if "--stock" in sys.argv:
    t, _ = run(STOCK, ["search", "reads", "contigs", "res_s", "tmp_s"] + args[:-1] + ["1"])
    out["stock_wall_s"] = round(t, 2)
    n, bad, _ = dbio.diff_dbs(os.path.join(w, "res_s"), os.path.join(w, "res_g"))
    out["entries"] = n
    out["entries_differing"] = bad
print(json.dumps(out))
