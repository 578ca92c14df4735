"""GPU box: wall-clock timeline of one fused `mmseqs search` (configs[2], scale 10 = 10 000 x 1 000 000): every log line of the
patched binary with the time at which it appeared (stdbuf: unbuffered), so that the gaps between the MMGPU_TRACE laps show."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
translated = "--translated" in sys.argv      # configs[4] as `search --search-type 2` (1 000 reads x 5 000 contigs) instead of configs[2]
argv = [a for a in sys.argv[1:] if a != "--translated"]
scale = int(argv[0]) if argv else 10
threads = os.environ.get("MMGPU_BENCH_THREADS", "32")
w = tempfile.mkdtemp(prefix="mmgpu_timeline_")
if translated:
    queries, (tres, toff), _ = wl.config5_nucleotide(500 * scale, 100 * scale, 10000, seed=20)
    wl.write_nucl_fasta(os.path.join(w, "t.fasta"), wl.split(tres, toff), "c")
    wl.write_nucl_fasta(os.path.join(w, "q.fasta"), queries, "r")
else:
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=2000 * scale, members=50, n_queries=1000 * scale, seed=10)
    wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
    wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
for n in ("q", "t"):
    subprocess.run([STOCK, "createdb", n + ".fasta", n, "-v", "1"], cwd=w, check=True)
for rep in range(2):
    t0 = time.time()
    env = dict(os.environ, MMGPU_TRACE="1")
    opts = ["--search-type", "2"] if translated else ["-s", "5.7"]
    p = subprocess.Popen(["stdbuf", "-o0", "-e0", MMGPU, "search", "q", "t", "res%d" % rep, "tmp%d" % rep] + opts + ["--threads", threads, "-v", "3"],
                         cwd=w, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, bufsize=0)
    lines = []
    buf = b""
    while True:
        d = os.read(p.stdout.fileno(), 65536)
        if not d:
            break
        t = time.time() - t0
        buf += d
        while b"\n" in buf:
            line, buf = buf.split(b"\n", 1)
            line = line.strip().lstrip(b"[=").strip()
            if line and not line.startswith(b"=") and (b"\t" not in line):
                lines.append("%.3f %s" % (t, line.decode(errors="replace")[:140]))
    p.wait()
    print("==== run %d: %.3f s ====" % (rep, time.time() - t0))
    if rep == 1:
        # the workflow's modules: from the echo of a module's command line to its "Time for processing"
        mods, cur = [], None
        names = ("extractorfs", "translatenucs", "prefilter", "align", "offsetalignment", "swapresults", "splitsequence", "extractframes",
                 "mvdb", "rmdb", "lndb", "createsubdb", "filterdb", "result2stats", "mergedbs")
        for l in lines:
            t, _, rest = l.partition(" ")
            if rest.split(" ")[0] in names:
                cur = [rest.split(" ")[0], float(t), None]
                mods.append(cur)
            elif rest.startswith("Time for processing") and cur is not None and cur[2] is None:
                cur[2] = float(t)
        print("---- modules (wall seconds between a module's command line and its 'Time for processing') ----")
        for name, a, b in mods:
            print("%-16s %7.3f s  (from %.3f)" % (name, (b if b is not None else a) - a, a))
        print("---- log ----")
        print("\n".join(lines))
