"""Shared helpers of the nucleotide alignment tests (row a18)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "nucl_vectors.npz")))


def golden_cases(g):
    """-> list of (q, t, diagonal, reverse, past_q, past_t, expected tuple, backtrace)"""
    out = []
    qoff, toff, boff = g["qoff"].astype(np.int64), g["toff"].astype(np.int64), g["boff"].astype(np.int64)
    bt = g["bt"].tobytes().decode()
    for i in range(len(qoff) - 1):
        m = g["meta"][i]
        out.append((g["qres"][qoff[i]:qoff[i + 1]], g["tres"][toff[i]:toff[i + 1]], int(m[0]), int(m[1]), int(m[2]), int(m[3]),
                    tuple(int(x) for x in g["expected"][i]), bt[boff[i]:boff[i + 1]]))
    return out


def mutate(rng, s, sub, indel):
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
