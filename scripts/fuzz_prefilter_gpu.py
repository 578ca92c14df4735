"""TEST INFRASTRUCTURE (GPU box): random prefilter cases on the device against the oracle, every stage and the final lists -
database sizes across several device-bin counts, planted homolog fractions, list lengths / reference bin counts, with the
emitter table of the replay kernel at its size and cut to 2 entries (redo path) and the non-empty-k-mer bit table on and off.
    python scripts/fuzz_prefilter_gpu.py [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmseqs2_amd
from mmseqs2_amd import workloads as wl
from oracle.pyoracle import Oracle
from tests import pf_common as pc
from tests import pf_gpu_check as chk

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
gpu = mmseqs2_amd.MMGpu(0)
g = pc.golden()
thr = int(g["kmer_thr"])
swo = Oracle()
rng = np.random.default_rng(2026)
bad = 0
t0 = time.time()
for r in range(rounds):
    nt = int(rng.choice([3000, 9000, 30000, 70000]))
    nq = int(rng.integers(8, 28))
    planted = float(rng.choice([0.02, 0.1, 0.3, 0.6]))
    seed = int(rng.integers(1, 1 << 30))
    (qres, qoff), (tres, toff) = pc.synthetic_case(nq, nt, seed=seed, planted=planted)
    qs = []
    for i, q in enumerate(wl.split(qres, qoff)):
        if i % 5 == 2:
            q = q[: 25 + 3 * i]
        qs.append(dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q),
                       identity_id=int(rng.integers(0, nt)) if i % 3 == 0 else None))
    orc = pc.pf_oracle()
    orc.build_index(tres, toff, thr)
    for cap, bitmap in ((None, "auto"),):
        chk.load_case(gpu, g, tres, toff, thr)
        for mh, rb in ((300, 2), (int(rng.integers(5, 60)), int(rng.choice([2, 8, 64])))):
            ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=True, label="fuzz %d nt=%d planted=%.2f cap=%s bitmap=%s" % (r, nt, planted, cap, bitmap))
            print(rep[0], "OK" if ok else "MISMATCH")
            if not ok:
                bad += 1
                print("\n".join(rep[1:20]))
print("fuzz_prefilter_gpu: %d rounds, %d mismatching configurations, %.0f s" % (rounds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
