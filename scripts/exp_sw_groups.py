"""GPU box experiment: alignment stage of the 10k x 1M search per kernel group, score+end vs score+end+start.
Usage: MMGPU_LIB=... python scripts/exp_sw_groups.py [families] [queries]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

torch.cuda.init()
import mmseqs2_amd
from mmseqs2_amd import capi, evalue, workloads as wl

m = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
gpu = mmseqs2_amd.MMGpu(0)
km16 = m["vtml80_kmer"].astype(np.int16)
mat = m["blosum62_sw"]
sub16 = mat.astype(np.int16)
nfam = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(nfam, 50, nq, seed=10)
qs = wl.split(qres, qoff)
s3, i3 = capi.host_score_matrix(km16, 3)
gpu.load_targets(tres, toff, 21)
gpu.pf_build_index(6, 21, True, s3, i3, km16, 112, m["blosum62_ungapped"])
queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, m["vtml80_pback"], q)[0], identity_id=None) for q in qs]
pfb = gpu.pf_prepare(queries, 112, max_hits=300, min_diag_score=15, ref_bins=2)
pfb.run()
thr = {}
swq = []
for q in qs:
    L = len(q)
    if L not in thr:
        thr[L] = evalue.min_score_for_evalue(1e-3, L, float(toff[-1]))
    swq.append(dict(q=q, comp_bias=capi.host_comp_bias(sub16, m["blosum62_pback"], q)[1], min_start_score=thr[L]))
msh = gpu.sw_marshal_queries(mat, 11, 1, swq)
chk = None
for mode, name in ((1, "score+end+start"), (0, "score+end")):
    os.environ.pop("MMGPU_TRACE", None)
    best = 1e9
    for rep in range(4):
        fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=mode, marshalled=msh)
        fb.run()
        ms = fb.kernel_ms()
        best = min(best, ms)
        cells = fb.cells
        if mode == 1 and rep == 0:
            r = fb.fetch()
            chk = (int(r["score"].astype(np.int64).sum()), int(r["q_start"].astype(np.int64).sum()), int(r["t_start"].astype(np.int64).sum()),
                   int(r["q_end"].astype(np.int64).sum()), int(r["t_end"].astype(np.int64).sum()))
        fb.free()
    print("%-18s align kernels %.2f ms  %.1f GCUPS" % (name, best, cells / best / 1e6), flush=True)
    os.environ["MMGPU_TRACE"] = "1"
    fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=mode, marshalled=msh)
    fb.run()
    gpu.synchronize()
    fb.free()
print("checksums (score, q_start, t_start, q_end, t_end):", chk)
