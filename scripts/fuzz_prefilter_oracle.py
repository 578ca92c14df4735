"""TEST INFRASTRUCTURE. Fuzz the plain-C prefilter restatement (oracle/prefilter_oracle.c) against the real
reference classes (oracle/_ref/libmmref.so).  Needs /root/reference (matrices).  Usage:
    python scripts/fuzz_prefilter_oracle.py [n_targets] [n_queries] [sens]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle.pyoracle import PfOracle, RefPrefilter, Oracle, kmer_threshold
from mmseqs2_amd import workloads as W

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sens = float(sys.argv[3]) if len(sys.argv) > 3 else 5.7
k = 6
ref = RefPrefilter(k)
km8, um8, km16, pback = ref.matrices()
t0 = time.time()
orc = PfOracle(km16, um8, k=k, spaced=True)
print("oracle score matrices %.1fs" % (time.time() - t0))
for which, s, i in ((3, orc.s3, orc.i3), (2, orc.s2, orc.i2)):
    rs, ri = ref.score_matrix(which)
    assert np.array_equal(rs, s), which
    assert np.array_equal(ri, i), which
print("score matrices identical")

rng = np.random.default_rng(7)
# similar k-mer lists
for _ in range(300):
    kmer = rng.integers(0, 20, k).astype(np.uint8)
    thr = int(rng.integers(60, 140))
    a, na = orc.kmer_list(kmer, thr)
    b, nb = ref.kmer_list(kmer, thr)
    assert na == nb and np.array_equal(a, b), (kmer, thr, na, nb)
print("kmer lists identical")

# targets with planted homologs + some X
(qres, qoff), (tres, toff) = W.config2_align_only(nq, nt, planted_frac=0.3, seed=11)
tres = tres.copy(); qres = qres.copy()
xs = rng.choice(len(tres), len(tres) // 500, replace=False); tres[xs] = 20
xs = rng.choice(len(qres), len(qres) // 300, replace=False); qres[xs] = 20
kthr = kmer_threshold(sens, k)
t0 = time.time(); ref.build_index(tres, toff, kthr); t1 = time.time()
orc.build_index(tres, toff, kthr); t2 = time.time()
ro, ri, rp = ref.index_dump()
assert np.array_equal(ro, orc.offsets)
assert np.array_equal(ri, orc.ids[:orc.n_entries]) and np.array_equal(rp, orc.pos[:orc.n_entries])
print("index identical: %d entries (ref %.1fs, oracle %.1fs)" % (orc.n_entries, t1 - t0, t2 - t1))

swo = Oracle()   # composition bias (restatement already pinned against the reference)
for max_hits, fb in ((300, 0), (20, 0), (20, 16), (50, 2048), (300, 128)):
    bins = ref.make_matcher(max_hits=max_hits, force_bins=fb)
    print("reference bins =", bins, "max_hits", max_hits)
    qs = W.split(qres, qoff)
    nh = 0
    for qi, q in enumerate(qs):
        cb = swo.comp_bias(km16, pback, q)
        ident = None if qi % 3 else int(rng.integers(0, nt))
        r = ref.match(q, ident)
        o = orc.match(q, cb, bins, max_hits=max_hits, identity_id=ident)
        assert o["stats"]["rc"] == 0
        ok = (np.array_equal(r["id"], o["id"]) and np.array_equal(r["score"], o["score"])
              and np.array_equal(r["diagonal"], o["diagonal"]) and r["db_matches"] == o["stats"]["db_matches"])
        if not ok:
            print("MISMATCH query", qi, len(q), r["db_matches"], o["stats"])
            print(r["id"][:20], r["score"][:20], r["diagonal"][:20])
            print(o["id"][:20], o["score"][:20], o["diagonal"][:20])
            sys.exit(1)
        nh += len(r["id"])
    print("matchQuery identical for %d queries, %d hits, last stats %s" % (len(qs), nh, o["stats"]))
