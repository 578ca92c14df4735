#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ from the REAL reference (oracle/_ref/libmmref.so, built
by `make -C oracle ref` from /root/reference).  Runs only where /root/reference exists; the outputs are small
.npz files that travel to the GPU box.

    python tests/golden/make_golden.py

matrices.npz : the integer substitution matrices the reference derives from data/*.out
               (BaseMatrix::generateSubMatrix, BaseMatrix.cpp:141-154) and its background pBack
sw_vectors.npz : (query, target) pairs with the reference's s_align fields in modes 0/1/2
nucl_vectors.npz : nucleotide (query, target, diagonal, strand) cases with what BandedNucleotideAligner::align returns
               (real reference classes through oracle/ref_shim_nucl.cpp): score, positions, identities, backtrace
prefilter_vectors.npz : queries + targets with the hit_t lists QueryMatcher::matchQuery returns (real reference
               classes through oracle/ref_shim_pref.cpp) for several (max_hits, bin count) settings
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import RefLib  # noqa: E402
from mmseqs2_amd import workloads as wl  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def matrices():
    d = {}
    sw = RefLib("blosum62.out", 2.0, 0.0)              # Alignment.cpp:152
    d["blosum62_sw"] = sw.matrix()
    d["num2aa"] = np.frombuffer(sw.num2aa().encode(), np.uint8)
    ung = RefLib("blosum62.out", 2.0, -0.2, gap_open=0)            # Prefiltering.cpp:69
    d["blosum62_ungapped"] = ung.matrix()
    km = RefLib("VTML80.out", 8.0, -0.2, gap_open=0)               # Prefiltering.cpp:68
    d["vtml80_kmer"] = km.matrix()
    pb = np.zeros(21, np.float64)
    import ctypes
    sw.L.mmref_get_pback.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    sw.L.mmref_get_pback(sw.c, pb.ctypes.data)
    d["blosum62_pback"] = pb
    # lets the prebuilt oracle/_ref library rebuild its matrix where /root/reference is absent
    d["blosum62_serialized"] = np.frombuffer(sw.serialized_matrix(), np.uint8)
    pb2 = np.zeros(21, np.float64)
    km.L.mmref_get_pback.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    km.L.mmref_get_pback(km.c, pb2.ctypes.data)
    d["vtml80_pback"] = pb2
    d["vtml80_serialized"] = np.frombuffer(km.serialized_matrix(), np.uint8)
    from oracle.pyoracle import RefNucl
    nu = RefNucl()
    d["nucleotide"] = nu.matrix()
    d["nucleotide_reverse"] = nu.reverse_lookup()
    d["nucleotide_serialized"] = np.frombuffer(nu.serialized_matrix(), np.uint8)
    nu.close()
    np.savez_compressed(os.path.join(OUT, "matrices.npz"), **d)
    print("matrices.npz", {k: v.shape for k, v in d.items()})
    return sw


def sw_vectors(ref):
    rng = np.random.default_rng(20260922)
    qs, ts, rows, bts, cbs = [], [], [], [], []
    lens = [1, 2, 7, 16, 17, 31, 64, 100, 127, 128, 129, 200, 255, 256, 257, 350, 383, 384, 385, 500, 511, 512,
            513, 700, 1025, 1600]
    n = 0
    for it in range(260):
        Lq = int(lens[it % len(lens)])
        Lt = int(rng.choice(lens)) if it % 3 else int(rng.integers(1, 900))
        q = rng.choice(21, size=Lq, p=np.append(wl.BACKGROUND * 0.99, 0.01)).astype(np.uint8)
        kind = it % 5
        if kind == 0 and Lq > 10:
            t = wl.mutate(rng, q, float(rng.uniform(0.25, 0.95)))
            pre = rng.choice(20, size=int(rng.integers(0, 60)), p=wl.BACKGROUND).astype(np.uint8)
            t = np.concatenate([pre, t, pre[::-1]])
        elif kind == 1:
            t = q.copy()
        else:
            t = rng.choice(20, size=Lt, p=wl.BACKGROUND).astype(np.uint8)
        ref.sw_set_query(q)
        cbf = ref.comp_bias(q)
        cb = np.array([int(b - 0.5) if b < 0 else int(b + 0.5) for b in cbf.astype(np.float64)], np.int8)
        a0 = ref.sw_align(t, 0)
        a1 = ref.sw_align(t, 1)
        a2 = ref.sw_align(t, 2)
        assert a0["score"] == a1["score"] == a2["score"]
        qs.append(q); ts.append(t); cbs.append(cb)
        # score 0: the reference returns before touching identicalAACnt (uninitialised, ssw_align_private :850-852)
        ident = a2["ident"] if a0["score"] > 0 else 0
        rows.append([a0["score"], a0["q_end"], a0["t_end"], a1["q_start"], a1["t_start"], a0["word"], ident])
        bts.append(a2["bt"])
        n += 1
    qres, qoff = wl.seqs_from_list(qs)
    tres, toff = wl.seqs_from_list(ts)
    cbres = np.concatenate(cbs)
    bt_all = "\n".join(bts)
    np.savez_compressed(os.path.join(OUT, "sw_vectors.npz"), qres=qres, qoff=qoff, tres=tres, toff=toff, cb=cbres,
                        expect=np.array(rows, np.int32), bt=np.frombuffer(bt_all.encode(), np.uint8),
                        gap_open=11, gap_extend=1)
    print("sw_vectors.npz", n, "pairs; word-mode:", int(np.array(rows)[:, 5].sum()))


def prefilter_vectors():
    from oracle.pyoracle import RefPrefilter, Oracle, kmer_threshold
    rng = np.random.default_rng(20260923)
    k, sens = 6, 5.7
    ref = RefPrefilter(k)
    km8, um8, km16, pback = ref.matrices()
    (qres, qoff), (tres, toff) = wl.config2_align_only(24, 1500, planted_frac=0.4, seed=31)
    tres, qres = tres.copy(), qres.copy()
    tres[rng.choice(len(tres), len(tres) // 400, replace=False)] = 20     # some X
    qres[rng.choice(len(qres), len(qres) // 250, replace=False)] = 20
    qs = wl.split(qres, qoff)
    qs[3] = qs[3][:9]            # shorter than the spaced pattern: no k-mer window at all
    qs[4] = qs[4][:10]           # exactly one window
    qs[5] = np.tile(np.array([9, 9, 9, 0, 9, 9, 15, 9], np.uint8), 40)   # low complexity: strong composition bias
    qres, qoff = wl.seqs_from_list(qs)
    thr = kmer_threshold(sens, k)
    ref.build_index(tres, toff, thr)
    swo = Oracle()
    settings = [(300, 2), (300, 64), (20, 2), (7, 16), (3, 2048), (28, 2), (28, 8), (33, 4)]   # (max_hits, CacheFriendlyOperations bins)
    d = dict(qres=qres, qoff=qoff, tres=tres, toff=toff, kmer_thr=thr, k=k, spaced=1, min_diag_score=15,
             settings=np.array(settings, np.int32), vtml80_kmer16=km16, vtml80_pback=pback, blosum62_ungapped=um8)
    ident = np.array([0xFFFFFFFF if i % 3 else int(rng.integers(0, 1500)) for i in range(len(qs))], np.uint32)
    d["identity"] = ident
    cbs = [swo.comp_bias(km16, pback, q) for q in qs]
    d["comp_bias"] = np.concatenate(cbs).astype(np.float32)
    for si, (mh, bins) in enumerate(settings):
        got = ref.make_matcher(max_hits=mh, force_bins=bins)
        assert got == bins
        ids, scs, dgs, cnt, dbm = [], [], [], [], []
        for qi, q in enumerate(qs):
            r = ref.match(q, None if ident[qi] == 0xFFFFFFFF else int(ident[qi]))
            ids.append(r["id"]); scs.append(r["score"]); dgs.append(r["diagonal"]); cnt.append(len(r["id"]))
            dbm.append(r["db_matches"])
        d["hit_id_%d" % si] = np.concatenate(ids).astype(np.uint32)
        d["hit_score_%d" % si] = np.concatenate(scs).astype(np.int32)
        d["hit_diag_%d" % si] = np.concatenate(dgs).astype(np.uint16)
        d["hit_count_%d" % si] = np.array(cnt, np.uint32)
        d["db_matches"] = np.array(dbm, np.uint64)
        print("prefilter setting", (mh, bins), "hits", int(np.sum(cnt)), "saturated", int((np.concatenate(scs) > 255).sum()))
    np.savez_compressed(os.path.join(OUT, "prefilter_vectors.npz"), **d)


def nucl_vectors():
    """Reads with substitutions / indels against their source, both strands, right and wrong prefilter diagonals, short
    and long (z-drop, redone backward pass) cases, N letters, all five past-the-end letters."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(2024)
    ref = po.RefNucl()
    mat, rl = ref.matrix(), ref.reverse_lookup()

    def mutate(s, sub, indel):
        out, i = [], 0
        while i < len(s):
            r = rng.random()
            if r < indel / 2:
                out.append(int(rng.integers(0, 4)))
                continue
            if r < indel:
                i += int(rng.integers(1, 4))
                continue
            out.append(int(rng.integers(0, 4)) if rng.random() < sub else int(s[i]))
            i += 1
        return np.array(out if out else [0], np.uint8)

    def letters(a):
        return "".join(po.NUCL_LETTERS[int(x)] for x in a)

    qs, ts, meta, exp, bts = [], [], [], [], []
    lens = [1, 2, 7, 15, 16, 17, 33, 64, 65, 100, 150, 257, 400, 800, 1500, 3000, 6000]
    for L in lens:
        for rep in range(4):
            base = rng.integers(0, 4, size=L).astype(np.uint8)
            if rep == 3 and L > 10:
                base[rng.integers(0, L, size=max(1, L // 25))] = 4
            q = mutate(base, [0.0, 0.03, 0.1, 0.25][rep], [0.0, 0.01, 0.04, 0.08][rep])
            t = mutate(base, [0.0, 0.02, 0.05, 0.1][rep], [0.0, 0.01, 0.02, 0.05][rep])
            if rep >= 2:
                t = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 120))).astype(np.uint8), t,
                                    rng.integers(0, 4, size=int(rng.integers(0, 120))).astype(np.uint8)])
            pq, pt = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            ref.set_query(letters(q), pq)
            for reverse in (0, 1):
                tt = np.array([rl[x] for x in t[::-1]], np.uint8) if reverse else t
                for diag in (0, int(rng.integers(-len(tt), len(q) + 1)), int(rng.integers(0, 65536))):
                    r, bt = ref.align(letters(tt), diag & 0xFFFF, reverse, pt)
                    qs.append(q)
                    ts.append(tt)
                    meta.append((diag & 0xFFFF, reverse, pq, pt))
                    exp.append(r)
                    bts.append(bt)
    qoff = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.uint64)
    toff = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.uint64)
    boff = np.concatenate([[0], np.cumsum([len(x) for x in bts])]).astype(np.uint64)
    np.savez_compressed(os.path.join(OUT, "nucl_vectors.npz"), qres=np.concatenate(qs), qoff=qoff, tres=np.concatenate(ts),
                        toff=toff, meta=np.array(meta, np.int32), expected=np.array(exp, np.int64),
                        bt=np.frombuffer("".join(bts).encode(), np.uint8), boff=boff, mat=mat, reverse=rl)
    ref.close()
    print("nucl_vectors:", len(qs), "cases,", int(boff[-1]), "backtrace letters")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nucl":
        nucl_vectors()
        sys.exit(0)
    ref = matrices()
    sw_vectors(ref)
    prefilter_vectors()
    nucl_vectors()
