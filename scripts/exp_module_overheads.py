"""GPU box: fixed costs of the module processes - `mmseqs prefilter` / `align` / `search` (patched) on a 500-sequence database,
where everything that scales with the data is negligible: what remains is process start, parameter handling, the constructors
(score tables), device opening and first-launch costs."""
import json, os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
EX = os.path.join(ROOT, "oracle", "_ref", "dropin_data", "examples")
w = tempfile.mkdtemp(prefix="mmgpu_ovh_")
for f in os.listdir(os.path.dirname(EX)):
    if f.startswith("examples"):
        shutil.copy(os.path.join(os.path.dirname(EX), f), os.path.join(w, "q" + f[len("examples"):]))
def run(b, args, env=None):
    e = dict(os.environ); e.update(env or {})
    t0 = time.perf_counter()
    r = subprocess.run([b] + args, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e)
    assert r.returncode == 0, r.stdout[-2000:]
    return round(time.perf_counter() - t0, 3), r.stdout
out = {}
th = os.environ.get("MMGPU_BENCH_THREADS", "32")
out["version"] = run(MMGPU, ["version"])[0]
for rep in range(2):
    out["prefilter_%d" % rep], log = run(MMGPU, ["prefilter", "q", "q", "p%d" % rep, "-s", "5.7", "--threads", th, "-v", "3"], {"MMGPU_TRACE": "1"})
    out["align_%d" % rep], log2 = run(MMGPU, ["align", "q", "q", "p%d" % rep, "a%d" % rep, "--alignment-mode", "2", "--threads", th, "-v", "3"], {"MMGPU_TRACE": "1"})
    out["search_%d" % rep], log3 = run(MMGPU, ["search", "q", "q", "r%d" % rep, "t%d" % rep, "-s", "5.7", "--threads", th, "-v", "3"], {"MMGPU_TRACE": "1"})
out["stock_prefilter"] = run(STOCK, ["prefilter", "q", "q", "ps", "-s", "5.7", "--threads", th, "-v", "3"])[0]
out["stock_align"] = run(STOCK, ["align", "q", "q", "ps", "as", "--alignment-mode", "2", "--threads", th, "-v", "3"])[0]
out["prefilter_cpu_path"] = run(MMGPU, ["prefilter", "q", "q", "pc", "-s", "5.7", "--threads", th, "-v", "3"], {"MMGPU_DISABLE": "1"})[0]
print(json.dumps(out))
sys.stderr.write("\n".join(l for l in (log + log2 + log3).splitlines() if "[mmgpu" in l or "Time for" in l) + "\n")
