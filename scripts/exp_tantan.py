"""GPU box: device tantan masking of the 1 M-target database of BASELINE.json configs[2] - time, masked residues."""
import json, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, ".")
import mmseqs2_amd
from mmseqs2_amd import workloads as wl
vec = dict(np.load("tests/golden/tantan_vectors.npz"))
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 10
(_, _), (tres, toff), _, _ = wl.config3_prefilter(2000 * scale, 50, 10, seed=10)
gpu = mmseqs2_amd.MMGpu(0)
gpu.load_targets(tres, toff, 21)
out = {"targets": len(toff) - 1, "residues": int(toff[-1])}
for rep in range(3):
    t0 = time.perf_counter()
    n = gpu.pf_mask_targets(vec["vtml80_likelihood_ratios"], float(vec["mask_prob"]), 20)
    out["mask_s_%d" % rep] = round(time.perf_counter() - t0, 4)
out["masked"] = int(n)
gpu.close()
print(json.dumps(out))
