"""GPU box: `mmseqs search --search-type 3` (the blastn workflow: extractframes / splitsequence -> prefilter -> align ->
offsetalignment) through the stock and the patched binary on the nucleotide workload of bench.py (configs[4]-like: reads of
10 kb against contigs), wall seconds of the whole workflow and of its prefilter / align modules, result databases compared
entry by entry.  Stock runs with --threads 1 as well: its alignment output depends on what a thread mapped before (one residue
past the end of the per-thread buffers is read), the patched binary replays the one-thread history.
    python scripts/dropin_nucl_timing.py [contigs reads read_len]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl, dbio

contigs, reads, read_len = (int(x) for x in (sys.argv[1:4] + ["4000", "1000", "10000"][len(sys.argv) - 1:]))
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
hw = os.cpu_count() or 1
try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    threads = hw if q == "max" else int(max(1, min(hw, round(2 * float(q) / float(per)))))
except (OSError, ValueError):
    threads = hw
w = tempfile.mkdtemp(prefix="mmgpu_nucl_")
queries, (tres, toff), _ = wl.config5_nucleotide(contigs, reads, read_len, seed=20)
letters = np.frombuffer(b"ACTGN", np.uint8)      # NucleotideMatrix codes: A C T G X
with open(os.path.join(w, "q.fasta"), "wb") as fh:
    for i, s in enumerate(queries):
        fh.write(b">q%d\n" % i + letters[s].tobytes() + b"\n")
with open(os.path.join(w, "t.fasta"), "wb") as fh:
    for i in range(len(toff) - 1):
        fh.write(b">t%d\n" % i + letters[tres[int(toff[i]):int(toff[i + 1])]].tobytes() + b"\n")
env = dict(os.environ)
env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "mmseqs2_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
if os.environ.get("MMGPU_EMU") == "1":      # no GPU at hand: the CPU stand-in of the C-ABI (tests only; slow)
    env["LD_PRELOAD"] = os.path.join(ROOT, "oracle", "_build", "emu", "libmmgpu.so")
extra = os.environ.get("MMGPU_NUCL_EXTRA", "").split()


def run(b, a):
    t0 = time.perf_counter()
    r = subprocess.run([b] + a, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError("%s %s failed: %s" % (os.path.basename(b), a[0], r.stdout[-600:]))
    return time.perf_counter() - t0, r.stdout


def module_times(log):      # "Time for processing" of the prefilter and align steps inside the workflow's log
    out = {}
    for name in ("prefilter", "align"):
        m = re.search(r"^%s .*?Time for processing: (\d+)h (\d+)m (\d+)s (\d+)ms" % name, log, re.S | re.M)
        if m:
            h, mi, s, ms = (int(x) for x in m.groups())
            out[name + "_s"] = round(3600 * h + 60 * mi + s + ms / 1000.0, 3)
    return out


run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"])
run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"])
res = {"workload": "%d reads of %d nt vs %d contigs (median 20 kb), `mmseqs search --search-type 3 -a`" % (reads, read_len, contigs)}
for name, b, th in (("stock_threads_%d" % threads, STOCK, threads), ("stock_threads_1", STOCK, 1), ("patched_threads_%d" % threads, MMGPU, threads)):
    t, log = run(b, ["search", "q", "t", "res_" + name, "tmp_" + name, "--search-type", "3", "-a", "--threads", str(th), "-v", "3", "--remove-tmp-files", "0"] + extra)
    res[name] = dict(workflow_wall_s=round(t, 2), **module_times(log))
    if b == MMGPU:
        res[name]["nucleotide_alignment_on_the_device"] = "MMGPU: nucleotide alignment on the device" in log
        if env.get("MMGPU_TRACE"):
            res[name]["trace"] = [l.strip()[:160] for l in log.replace("\r", "\n").split("\n") if l.strip().startswith(("[mmgpu", "[nucl_align", "Time for", "Index table", "MMGPU:"))]


def same(a, b):
    da, db = dbio.read_db(os.path.join(w, a)), dbio.read_db(os.path.join(w, b))
    return sum(1 for k in da if da[k] != db.get(k)) + sum(1 for k in db if k not in da), len(da)


def find(tmp, name):
    for root, _, files in os.walk(os.path.join(w, tmp)):
        if name + ".index" in files:      # (the data may be in per-thread files name.0, name.1, ...)
            return os.path.relpath(os.path.join(root, name), w)
    return None


# the prefilter databases of the workflows (strand-queries x split targets), entry by entry
pa, pb = find("tmp_patched_threads_%d" % threads, "pref_0"), find("tmp_stock_threads_1", "pref_0")
if pa and pb:
    res["prefilter_entries_differing_patched_vs_stock"], res["prefilter_entries"] = same(pa, pb)
res["entries_differing_patched_vs_stock_one_thread"], res["entries"] = same("res_patched_threads_%d" % threads, "res_stock_threads_1")
res["entries_differing_stock_%d_threads_vs_stock_one_thread" % threads] = same("res_stock_threads_%d" % threads, "res_stock_threads_1")[0]
print(json.dumps(res))
