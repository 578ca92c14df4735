"""Shared helpers of the prefilter tests (test infrastructure)."""
import os

import numpy as np

from mmseqs2_amd import workloads as wl

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden():
    if "g" not in _cache:
        _cache["g"] = dict(np.load(os.path.join(GOLDEN, "prefilter_vectors.npz")))
    return _cache["g"]


def pf_oracle():
    """PfOracle over the golden matrices (the 3-mer table takes a few seconds: built once per session)."""
    if "o" not in _cache:
        from oracle.pyoracle import PfOracle
        g = golden()
        _cache["o"] = PfOracle(g["vtml80_kmer16"], g["blosum62_ungapped"], k=int(g["k"]), spaced=bool(g["spaced"]))
    return _cache["o"]


def golden_queries(g):
    qs = wl.split(g["qres"], g["qoff"])
    cbs = wl.split(g["comp_bias"].view(np.uint32), g["qoff"])
    out = []
    for i, q in enumerate(qs):
        ident = None if g["identity"][i] == 0xFFFFFFFF else int(g["identity"][i])
        out.append(dict(q=q, comp_bias=cbs[i].view(np.float32), identity_id=ident))
    return out


def expected_hits(g, si):
    off = np.concatenate([[0], np.cumsum(g["hit_count_%d" % si].astype(np.int64))])
    return [(g["hit_id_%d" % si][off[i]:off[i + 1]], g["hit_score_%d" % si][off[i]:off[i + 1]],
             g["hit_diag_%d" % si][off[i]:off[i + 1]]) for i in range(len(off) - 1)]


def keepmax_reference(dd_id, dd_diag, dd_count, min_score):
    """keepMaxElement semantics on the oracle's post-findDuplicates dump: per target the first element (dump order)
    holding the target's maximum count; only count >= min_score."""
    best = {}
    for i, d, c in zip(dd_id.tolist(), dd_diag.tolist(), dd_count.tolist()):
        if i not in best or c > best[i][1]:
            best[i] = (d, c)
    return {i: v for i, v in best.items() if v[1] >= min_score}


def synthetic_case(n_queries, n_targets, seed, planted=0.2, x_every=400):
    rng = np.random.default_rng(seed)
    (qres, qoff), (tres, toff) = wl.config2_align_only(n_queries, n_targets, planted_frac=planted, seed=seed)
    tres, qres = tres.copy(), qres.copy()
    tres[rng.choice(len(tres), max(1, len(tres) // x_every), replace=False)] = 20
    qres[rng.choice(len(qres), max(1, len(qres) // x_every), replace=False)] = 20
    return (qres, qoff), (tres, toff)


def shards(g, world):
    """Contiguous target shards (residues, offsets) and their sizes."""
    n = len(g["toff"]) - 1
    cut = [n * r // world for r in range(world + 1)]
    out = []
    for r in range(world):
        o = g["toff"][cut[r]:cut[r + 1] + 1].astype(np.int64)
        out.append((g["tres"][o[0]:o[-1]], (o - o[0]).astype(np.uint64)))
    return out, [cut[r + 1] - cut[r] for r in range(world)]


def long_case(seed, long_query=False, device=False):
    """Targets of 32768 residues or more among ordinary ones (UngappedAlignment::computeLongScore and the batches of
    scoreDiagonalAndUpdateHits, UngappedAlignment.cpp:187-312).  Query 0 has a homolog inside a 40 000-residue target at 1 000 and
    query 1 at 34 000 (a diagonal beyond the 16-bit range); query 2 has one at 66 000 of a 70 000-residue target (index positions wrap
    at 65 536); query 3 meets MANY targets on diagonal 0 - substitution-only copies of itself, three of them at the start of long
    targets and one at position 65 536 of a long target (diagonal -65 536 == 0 in 16 bits) - so that full batches of eight elements
    of one diagonal hold long targets.  long_query: a 33 000-residue query that carries target 5 at 5 000 and a piece of a long
    target.  device: no sequence beyond 65 535 residues (the reference's default --max-seq-len, Parameters.h:271, which is what the
    library accepts): the 70 000-residue target becomes 65 535 with the homolog at 62 000, the copy at 65 536 one at 0.
    Returns (queries, targets) as lists of uint8 arrays; the long targets are dealt among the others by the seed."""
    rng = np.random.default_rng(seed)
    (qres, qoff), (tres, toff) = wl.config2_align_only(6, 300, planted_frac=0.4, seed=seed)
    qs, tl = wl.split(qres, qoff), wl.split(tres, toff)
    bg = lambda n: rng.choice(20, size=n, p=wl.BACKGROUND).astype(np.uint8)
    extra = []
    big = bg(40000)
    for k, at in ((0, 1000), (1, 34000)):
        h = wl.mutate(rng, qs[k], 0.8)
        big[at:at + len(h)] = h
    extra.append(big)
    big2 = bg(65535 if device else 70000)
    h = wl.mutate(rng, qs[2], 0.85)
    at2 = 62000 if device else 66000
    big2[at2:at2 + len(h)] = h
    extra.append(big2)
    q3 = qs[3]
    for _ in range(int(rng.integers(9, 20))):                       # ordinary targets on diagonal 0
        extra.append(wl.mutate(rng, q3, 0.9, max_indels=0))
    for n in (33000, 36000, 50000):                                 # long targets on diagonal 0
        b = bg(n)
        b[:len(q3)] = wl.mutate(rng, q3, 0.9, max_indels=0)
        extra.append(b)
    b = bg(65000 if device else 67000)
    at3 = 0 if device else 65536
    b[at3:at3 + len(q3)] = wl.mutate(rng, q3, 0.9, max_indels=0)
    extra.append(b)
    if long_query:
        lq = bg(33000)
        lq[5000:5000 + len(tl[5])] = tl[5]
        lq[20000:20600] = big[10000:10600]
        qs.append(lq)
    tl = tl + extra
    perm = rng.permutation(len(tl))
    return qs, [tl[i] for i in perm]
