import os, sys, json
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT); sys.path.insert(0, ROOT)
import mmseqs2_amd
from mmseqs2_amd import capi, workloads as wl
import ctypes
pair = int(sys.argv[1])
mats = dict(np.load("tests/golden/matrices.npz")); mat = mats["blosum62_sw"]; sub16 = mat.astype(np.int16)
(qres, qoff), (tres, toff), fam_t, fam_q = wl.config3_prefilter(400, 50, 200, seed=10)
qs = wl.split(qres, qoff)
gpu = mmseqs2_amd.MMGpu(0); gpu.load_targets(tres, toff, 21)
order = np.argsort(fam_t, kind="stable"); starts = np.searchsorted(fam_t[order], np.arange(401))
qi, k = divmod(pair, 50)
q = qs[qi]; f = int(fam_q[qi]); ids = order[starts[f]:starts[f + 1]].astype(np.uint32)
queries = [dict(q=q, comp_bias=capi.host_comp_bias(sub16, mats["blosum62_pback"], q)[1], targets=ids, min_start_score=0)]
b = gpu.sw_prepare(mat, 11, 1, queries, mode=1); b.run(); res = b.fetch()
pi = np.array([k, k], np.uint32)
os.environ["MMGPU_B2_DBG"] = str((0 + 1) << 8)
cap = 1024
out = np.zeros(2, capi.SW_BLOCK_DTYPE); g = np.zeros((2, 1 + 4 * cap), np.uint32)
gpu._check(gpu.L.mmgpu_sw_block_growth(gpu.ctx, b.handle, pi.ctypes.data_as(ctypes.c_void_p), 2, out.ctypes.data_as(ctypes.c_void_p), g.ctypes.data_as(ctypes.c_void_p), cap))
print("blocks", int(g[0, 0]), "first", g[0, 1:13].reshape(3, 4).tolist())
d = g[0, 1 + 2048:1 + 2048 + 12 * 14].astype(np.int32).reshape(14, 12)
print("dir bs off_max best_max mx Dmm growmax rmax dmax off st_i st_j")
for r in d: print(r.tolist())
