"""GPU box: the fused `mmseqs search` of configs[2] (scale 10 = 10 000 x 1 M) attached to a resident mmgpu_server (the counterpart
of the reference's gpuserver: targets, masked copy and k-mer index stay on the device between searches) through
LD_PRELOAD=libmmgpu_client.so, against the same search opening the device itself."""
import json, os, signal, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl, dbio
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
SERVER = os.path.join(ROOT, "mmseqs2_amd", "lib", "mmgpu_server")
CLIENT = os.path.join(ROOT, "mmseqs2_amd", "lib", "libmmgpu_client.so")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 10
threads = os.environ.get("MMGPU_BENCH_THREADS", "32")
w = tempfile.mkdtemp(prefix="mmgpu_srv_")
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=2000 * scale, members=50, n_queries=1000 * scale, seed=10)
wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
for n in ("q", "t"):
    subprocess.run([STOCK, "createdb", n + ".fasta", n, "-v", "1"], cwd=w, check=True)


def search(tag, env_extra):
    env = dict(os.environ, MMGPU_TRACE="1")
    env.update(env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([MMGPU, "search", "q", "t", "res_" + tag, "tmp_" + tag, "-s", "5.7", "--threads", threads, "-v", "3"], cwd=w, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-3000:]
    sys.stderr.write("==== %s: %.3f s ====\n" % (tag, dt))
    sys.stderr.write("\n".join(l for l in r.stdout.splitlines() if "[mmgpu prefilter]" in l or "[mmgpu align]" in l or "Time for" in l or "MMGPU" in l) + "\n")
    return dt


out = {"direct_s": [round(search("direct%d" % i, {}), 3) for i in range(2)]}
sock = os.path.join(w, "mmgpu.sock")
srv = subprocess.Popen([SERVER, "--socket", sock], stderr=subprocess.PIPE, text=True)
t0 = time.time()
while not os.path.exists(sock):
    assert srv.poll() is None and time.time() - t0 < 120
    time.sleep(0.05)
env = {"LD_PRELOAD": CLIENT, "MMGPU_SERVER_SOCKET": sock}
out["via_server_s"] = [round(search("srv%d" % i, env), 3) for i in range(3)]
srv.send_signal(signal.SIGTERM)
try:
    srv.wait(timeout=30)
except subprocess.TimeoutExpired:
    srv.kill()
sys.stderr.write(srv.stderr.read()[-1500:])
n, bad, _ = dbio.diff_dbs(os.path.join(w, "res_direct0"), os.path.join(w, "res_srv2"))
out["entries"] = n
out["entries_differing"] = bad
print(json.dumps(out))
