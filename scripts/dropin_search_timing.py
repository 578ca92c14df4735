"""GPU box: wall time of `mmseqs search` through the patched binary on BASELINE.json configs[2] (scale 10 = 10 000 queries x
1 000 000 targets), fused (prefilter + align inside the search process) and as the workflow script (MMGPU_FUSED=0), with the
MMGPU_TRACE lines; `--stock` adds the stock binary (slow).  Prints one JSON object, traces to stderr."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl, dbio
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
STUB = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock_stub")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
scale = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
threads = os.environ.get("MMGPU_BENCH_THREADS", "32")
extra = [a for a in sys.argv[2:] if a not in ("--stock", "--stub")]
w = tempfile.mkdtemp(prefix="mmgpu_search_timing_")
t0 = time.perf_counter()
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=2000 * scale, members=50, n_queries=1000 * scale, seed=10)
wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
t_gen = time.perf_counter() - t0


def run(b, args, env_extra=None, trace=False):
    env = dict(os.environ)
    if trace:
        env["MMGPU_TRACE"] = "1"
    if env_extra:
        env.update(env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([b] + args, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-3000:]
    return dt, r.stdout


run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"])
run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"])
out = {"workload": "%d queries x %d targets, mmseqs search -s 5.7 (defaults: --mask 1, --max-seqs 300, -e 1e-3, alignment-mode 2), %s threads"
                   % (1000 * scale, 100000 * scale, threads), "generate_s": round(t_gen, 1)}
base = ["-s", "5.7", "--threads", threads, "-v", "3"] + extra
for rep in range(2):
    dt, log = run(MMGPU, ["search", "q", "t", "res_fused%d" % rep, "tmp_fused%d" % rep] + base, trace=True)
    out["fused_wall_s_%d" % rep] = round(dt, 3)
    sys.stderr.write("==== fused search, run %d: %.3f s ====\n" % (rep, dt))
    sys.stderr.write("\n".join(l for l in log.splitlines() if "[mmgpu" in l or "Time for" in l or "MMGPU" in l) + "\n")
dt, log = run(MMGPU, ["search", "q", "t", "res_script", "tmp_script"] + base, env_extra={"MMGPU_FUSED": "0"}, trace=True)
out["script_wall_s"] = round(dt, 3)
sys.stderr.write("==== workflow script (two child processes): %.3f s ====\n" % dt)
sys.stderr.write("\n".join(l for l in log.splitlines() if "[mmgpu" in l or "Time for" in l) + "\n")
n, bad, _ = dbio.diff_dbs(os.path.join(w, "res_fused0"), os.path.join(w, "res_script"))
out["fused_equals_script"] = bad == 0
out["entries"] = n
# environment variants of the fused run: MMGPU_TIMING_VARIANTS="NAME=VAL,NAME2=VAL2;NAME=VAL"
for vi, spec in enumerate([x for x in os.environ.get("MMGPU_TIMING_VARIANTS", "").split(";") if x]):
    env_extra = dict(kv.split("=", 1) for kv in spec.split(","))
    best = None
    for rep in range(3 if "MMGPU_DB_FILE" in env_extra else 2):      # (the first run with a layout file builds and saves it)
        dt, log = run(MMGPU, ["search", "q", "t", "res_var%d_%d" % (vi, rep), "tmp_var%d_%d" % (vi, rep)] + base, env_extra=env_extra, trace=True)
        if rep == 0 and "MMGPU_DB_FILE" in env_extra:
            out["variant_" + spec + "_first_run_builds_and_saves"] = round(dt, 3)
            continue
        best = dt if best is None else min(best, dt)
    out["variant_" + spec] = round(best, 3)
    n, bad, _ = dbio.diff_dbs(os.path.join(w, "res_fused0"), os.path.join(w, "res_var%d_1" % vi))
    out["variant_%d_equals_fused" % vi] = bad == 0
    sys.stderr.write("==== variant %s: best %.3f s ====\n" % (spec, best))
    sys.stderr.write("\n".join(l for l in log.splitlines() if "[mmgpu" in l or "Time for" in l or "MMGPU" in l) + "\n")
for flag, b, name in (("--stock", STOCK, "stock"), ("--stub", STUB, "stock_stub")):
    if flag in sys.argv:
        dt, _ = run(b, ["search", "q", "t", "res_" + name, "tmp_" + name] + base[:-2] + ["-v", "1"])
        out[name + "_wall_s"] = round(dt, 2)
        n, bad, _ = dbio.diff_dbs(os.path.join(w, "res_fused0"), os.path.join(w, "res_" + name))
        out["fused_equals_" + name] = bad == 0
print(json.dumps(out))
