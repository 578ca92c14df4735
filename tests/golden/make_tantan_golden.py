"""Records tests/golden/tantan_vectors.npz from the REAL reference (oracle/_ref/libmmref.so = lib/tantan/tantan.cpp compiled with
the reference's AVX2 + FMA flags): a few hundred family-structured protein sequences, a third of them with planted tandem repeats
and low-complexity stretches, masked by tantan::maskSequences exactly as IndexBuilder::fillDatabase does (Masker::maskSequence,
maskTantan only, --mask-prob 0.9f); plus the likelihood-ratio table of the k-mer matrix (ProbabilityMatrix over VTML80) and the
per-letter repeat probabilities of every sequence.  Run in the build container:  python tests/golden/make_tantan_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mmseqs2_amd import workloads as wl                             # noqa: E402
from oracle import pyoracle                                        # noqa: E402


def sequences(seed=3):
    (_, _), (tres, toff), _, _ = wl.config3_prefilter(120, 4, 4, seed=seed)
    rng = np.random.default_rng(seed)
    ts = wl.split(tres, toff)
    for i in range(0, len(ts), 3):
        t = ts[i].copy()
        n_l = len(t)
        unit = rng.integers(0, 20, size=rng.integers(1, 9)).astype(np.uint8)
        a = int(rng.integers(0, max(1, n_l - 60)))
        n = min(n_l - a, int(rng.integers(20, 120)))
        rep = np.resize(unit, n)
        mut = rng.random(n) < 0.1
        rep[mut] = rng.integers(0, 20, size=int(mut.sum()))
        t[a:a + n] = rep
        if i % 9 == 0:      # an X inside, a very short and an empty sequence among them
            t[min(5, n_l - 1)] = 20
        ts[i] = t
    ts[1] = ts[1][:3]
    ts[4] = ts[4][:0]
    ts[7] = ts[7][:51]
    return ts


def main():
    ref = pyoracle.RefPrefilter()
    ts = sequences()
    tres, toff = wl.seqs_from_list(ts)
    mask_prob = float(np.float32(0.9))      # Parameters::maskProb is a float; IndexBuilder hands it on as one
    masked_res, n_masked, lr, _ = ref.tantan_mask(tres, toff, mask_prob)
    probs = np.zeros(len(tres), np.float32)
    for i, t in enumerate(ts):
        if len(t) == 0:
            continue
        r1, o1 = wl.seqs_from_list([t])
        probs[int(toff[i]):int(toff[i + 1])] = ref.tantan_mask(r1, o1, mask_prob)[3]
    np.savez_compressed(os.path.join(HERE, "tantan_vectors.npz"), tres=tres, toff=toff, masked=masked_res, n_masked=n_masked,
                        vtml80_likelihood_ratios=lr, probs=probs, mask_prob=mask_prob)
    print("wrote tantan_vectors.npz: %d sequences, %d residues, %d masked" % (len(ts), len(tres), n_masked))


if __name__ == "__main__":
    main()
