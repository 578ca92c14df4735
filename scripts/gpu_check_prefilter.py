"""TEST INFRASTRUCTURE (GPU box): stage-by-stage report of the device prefilter against the oracle.
    python scripts/gpu_check_prefilter.py [n_targets] [n_queries]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import mmseqs2_amd
from mmseqs2_amd import workloads as wl
from tests import pf_common as pc
from tests import pf_gpu_check as chk

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
gpu = mmseqs2_amd.MMGpu(0)
g = pc.golden()
orc = pc.pf_oracle()
thr = int(g["kmer_thr"])
if nt == 0:
    tres, toff = g["tres"], g["toff"]
    qs = pc.golden_queries(g)
else:
    (qres, qoff), (tres, toff) = pc.synthetic_case(nq, nt, seed=5, planted=0.1)
    from oracle.pyoracle import Oracle
    swo = Oracle()
    qs = [dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q), identity_id=None)
          for q in wl.split(qres, qoff)]
orc.build_index(tres, toff, thr)
chk.load_case(gpu, g, tres, toff, thr)
for mh, rb in ((300, 2), (20, 16)):
    ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=True, label="check nt=%d" % len(toff))
    print("\n".join(rep[:60]))
