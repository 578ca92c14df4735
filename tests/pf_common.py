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
