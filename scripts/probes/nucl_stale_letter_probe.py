"""Does the stock binary's nucleotide alignment depend on what its per-thread buffers held before?  (It reads one residue past the
end of Sequence::numSequence / queryRevCompSeq, BandedNucleotideAligner.cpp:61,68,93.)  The same reads are searched in two
database orders with --threads 1; per read the result lines are compared.  Usage: python nucl_stale_letter_probe.py [workdir]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
w = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="mmgpu_stale_")
os.makedirs(w, exist_ok=True)
rng = np.random.default_rng(5)
L = "ACGT"
contigs = ["".join(L[i] for i in rng.integers(0, 4, 4000)) for _ in range(6)]
reads = []
for i in range(80):
    c = contigs[int(rng.integers(0, 6))]
    ln = int(rng.integers(120, 900))
    a = int(rng.integers(0, len(c) - ln - 5))
    s = list(c[a:a + ln])
    for p in np.nonzero(rng.random(ln) < 0.03)[0]:
        if 25 < p < ln - 25:          # both ends match: the ungapped seed reaches the ends of the read
            s[p] = L[int(rng.integers(0, 4))]
    reads.append("".join(s))


def write(name, order):
    with open(os.path.join(w, name), "w") as f:
        for i in order:
            f.write(">q%d\n%s\n" % (i, reads[i]))


def run(a):
    r = subprocess.run([S] + a, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-800:]


write("qa.fasta", list(range(len(reads))))
write("qb.fasta", sorted(range(len(reads)), key=lambda i: -len(reads[i])))      # longest first: every later read sees stale letters
with open(os.path.join(w, "t.fasta"), "w") as f:
    for i, c in enumerate(contigs):
        f.write(">t%d\n%s\n" % (i, c))
run(["createdb", "t.fasta", "t", "-v", "1"])
res = {}
for n in "ab":
    run(["createdb", "q%s.fasta" % n, "q" + n, "-v", "1"])
    run(["search", "q" + n, "t", "res_" + n, "tmp_" + n, "--search-type", "3", "-a", "--threads", "1", "-v", "1"])
    run(["convertalis", "q" + n, "t", "res_" + n, "res_%s.m8" % n, "--format-output", "query,target,qstart,qend,tstart,tend,cigar,bits", "-v", "1"])
    d = {}
    for l in open(os.path.join(w, "res_%s.m8" % n)):
        f = l.rstrip("\n").split("\t")
        d.setdefault(f[0], []).append(tuple(f[1:]))
    res[n] = d
diff = [k for k in res["a"] if res["a"][k] != res["b"].get(k)]
print("reads with hits: %d / %d; reads whose result lines differ between the two database orders: %d" % (len(res["a"]), len(res["b"]), len(diff)))
for k in diff[:6]:
    print(k, res["a"][k][:1], res["b"][k][:1])
