"""Experiment: alignment time of one prefilter batch's lists, two-call path vs fused hand-over, per batch size."""
import os, sys, time, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from mmseqs2_amd import capi, workloads as wl
gpu = capi.MMGpu(0)
M = np.load(os.path.join(ROOT, "tests/golden/matrices.npz"))
km16 = M["vtml80_kmer"].astype(np.int16)
thr = int(163.2 - 8.917 * 5.7)
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(20000, 50, 4096, seed=10, target_seed=11)
qs = wl.split(qres, qoff)
s3, i3 = capi.host_score_matrix(km16, 3)
gpu.load_targets(tres, toff, 21)
gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, M["blosum62_ungapped"])
mat = M["blosum62_sw"]; sub16 = mat.astype(np.int16)
db_res = float(toff[-1])
def T(f, n=3):
    f(); gpu.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    gpu.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for bsz in [int(x) for x in (sys.argv[1:] or ["1024", "4096"])]:
    queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, M["vtml80_pback"], q)[0], identity_id=None) for q in qs[:bsz]]
    pfb = gpu.pf_prepare(queries, thr, max_hits=300, min_diag_score=15, ref_bins=2)
    pfb.run()
    h, c, st, _ = pfb.fetch()
    swq = [dict(q=q, comp_bias=capi.host_comp_bias(sub16, M["blosum62_pback"], q)[1], targets=h[i]["id"][:c[i]].copy(),
                min_start_score=max(1, int(math.ceil(math.log(0.041 * len(q) * db_res / 1e-3) / 0.267)))) for i, q in enumerate(qs[:bsz])]
    sep = gpu.sw_prepare(mat, 11, 1, swq, mode=1)
    fus = gpu.sw_prepare_from_pf(mat, 11, 1, swq, pfb, mode=1)
    print(bsz, "two-call run ms", round(T(sep.run), 2), "fused run ms", round(T(fus.run), 2), "cells", sep.cells, fus.cells, flush=True)
    sep.free(); fus.free(); pfb.free()
