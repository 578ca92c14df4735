"""GPU box experiment: where the alignment stage of the 10k x 1M search spends its time.  Builds the hit lists with the
prefilter once, then times the fused alignment (a) with start positions, (b) score + end only, and prints the library's
per-group trace (MMGPU_TRACE=1: the three kernel groups one at a time)."""
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
for mode, name in ((1, "score+end+start"), (0, "score+end")):
    for rep in range(3):
        fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=mode, marshalled=msh)
        fb.run()
        ms = fb.kernel_ms()
        cells = fb.cells
        fb.free()
    print("%-18s align kernels %.2f ms  %.1f GCUPS (forward cells %d)" % (name, ms, cells / ms / 1e6, cells))
os.environ["MMGPU_TRACE"] = "1"
fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=1, marshalled=msh)
t0 = time.perf_counter()
fb.run()
gpu.synchronize()
print("traced (groups serialised) total %.2f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(2):
    t0 = time.perf_counter()
    info, _ = fb.traceback(np.arange(300 * min(1000, nq), dtype=np.uint32))
    print("traceback of the first 1000 lists: %.1f ms incl. python, %.1f ms in the C-ABI calls, %d cigars" %
          ((time.perf_counter() - t0) * 1e3, fb.last_traceback_call_s * 1e3, int((info["status"] == 0).sum())))
fb.free()
