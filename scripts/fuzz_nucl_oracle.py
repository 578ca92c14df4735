"""Pins oracle/nucl_oracle.c against the real reference (oracle/_ref): ksw_extz2_sse alone and the whole
BandedNucleotideAligner::align, on random reads with substitutions and indels.  usage: fuzz_nucl_oracle.py [n] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po


def mutate(rng, s, sub, indel):
    out = []
    i = 0
    while i < len(s):
        r = rng.random()
        if r < indel / 2:
            out.append(int(rng.integers(0, 4)))             # insertion
            continue
        if r < indel:
            i += int(rng.integers(1, 4))                    # deletion
            continue
        out.append(int(rng.integers(0, 4)) if rng.random() < sub else int(s[i]))
        i += 1
    return np.array(out, np.uint8) if out else np.zeros(1, np.uint8)


def letters(a):
    return "".join(po.NUCL_LETTERS[int(x)] for x in a)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ref, orc = po.RefNucl(), po.NuclOracle()
    mat, rl = ref.matrix(), ref.reverse_lookup()
    print("matrix", mat.tolist(), "reverse", rl.tolist())
    bad = 0
    for it in range(n):
        L = int(rng.choice([1, 2, 5, 15, 16, 17, 31, 32, 33, 60, 64, 65, 100, 129, 200, 400, 1000, 3000]))
        base = rng.integers(0, 4, size=L).astype(np.uint8)
        if rng.random() < 0.2:
            base[rng.integers(0, L, size=max(1, L // 20))] = 4        # N
        q = mutate(rng, base, rng.choice([0.0, 0.02, 0.1, 0.3]), rng.choice([0.0, 0.01, 0.03, 0.1]))
        t = mutate(rng, base, rng.choice([0.0, 0.02, 0.1]), rng.choice([0.0, 0.01, 0.05]))
        if rng.random() < 0.3:
            t = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 80))).astype(np.uint8), t,
                                rng.integers(0, 4, size=int(rng.integers(0, 80))).astype(np.uint8)])
        # --- the extension kernel alone ---
        for flag in (po.KSW_SCORE_ONLY | po.KSW_EXTZ_ONLY, po.KSW_EXTZ_ONLY):
            for zdrop in (40, 100, -1):
                a = ref.ksw_extz2(q, t, mat.reshape(-1), 5, 2, 64, zdrop, flag)
                b = orc.ksw_extz2(q, t, mat.reshape(-1), 5, 2, 64, zdrop, flag)
                if a[0] != b[0] or not np.array_equal(a[1], b[1]):
                    bad += 1
                    print("KSW MISMATCH it", it, "flag", flag, "zdrop", zdrop, len(q), len(t), a[0], b[0], len(a[1]), len(b[1]))
        # --- the whole alignment step, both strands, the true diagonal and wrong ones ---
        pq, pt = int(rng.integers(0, 5)), int(rng.integers(0, 5))     # the letters "found" past the ends
        ref.set_query(letters(q), pq)
        for reverse in (0, 1):
            tt = t
            if reverse:                                   # make the reverse strand the matching one half of the time
                tt = np.array([rl[x] for x in t[::-1]], np.uint8) if rng.random() < 0.7 else t
            for diag in (0, int(rng.integers(-len(tt), len(q) + 1)), int(rng.integers(0, 65536))):
                a = ref.align(letters(tt), diag & 0xFFFF, reverse, pt)
                b = orc.align(q, tt, mat.reshape(-1), rl, 5, 2, 40, diag & 0xFFFF, reverse, pq, pt)
                if a != b:
                    bad += 1
                    print("ALIGN MISMATCH it", it, "rev", reverse, "diag", diag, len(q), len(tt), a[0], b[0], a[1][:60], b[1][:60])
    print("done", n, "iterations, mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
