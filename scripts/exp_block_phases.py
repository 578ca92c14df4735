"""Block aligner on the device at 1/10 of configs[2]'s family structure: call time, statuses, tiers, and (from the kernel's
profiling aid in mmgpu_sw_block::reserved) the share of each pair's time spent in the serial walk back."""
import json
import sys
import time

import numpy as np
import torch  # noqa: F401  (HIP runtime first)

sys.path.insert(0, ".")
import mmseqs2_amd
from mmseqs2_amd import capi
from mmseqs2_amd import workloads as wl


def main():
    mats = dict(np.load("tests/golden/matrices.npz"))
    mat = mats["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    (qres, qoff), (tres, toff), fam_t, fam_q = wl.config3_prefilter(2000, 50, 1000, seed=10)
    qs = wl.split(qres, qoff)
    gpu = mmseqs2_amd.MMGpu(0)
    gpu.load_targets(tres, toff, 21)
    order = np.argsort(fam_t, kind="stable")
    starts = np.searchsorted(fam_t[order], np.arange(2001))
    queries = []
    for qi, q in enumerate(qs):
        f = int(fam_q[qi])
        ids = order[starts[f]:starts[f + 1]].astype(np.uint32)
        queries.append(dict(q=q, comp_bias=capi.host_comp_bias(sub16, mats["blosum62_pback"], q)[1], targets=ids, min_start_score=0))
    b = gpu.sw_prepare(mat, 11, 1, queries, mode=1)
    b.run()
    res = b.fetch()
    idx = np.nonzero(res["word"] == 1)[0].astype(np.uint32)
    out = {"pairs": int(len(res)), "word1": int(len(idx))}
    for rep in range(3):
        t0 = time.perf_counter()
        blk, strs = b.block_backtrace(idx)
        out["call_s_%d" % rep] = round(time.perf_counter() - t0, 4)
        out["c_abi_s_%d" % rep] = round(b.last_block_call_s, 4)
    out["tiers"] = b.block_tiers()
    out["status"] = {int(k): int(v) for k, v in zip(*np.unique(blk["status"], return_counts=True))}
    share = blk["reserved"][blk["status"] == 0] & 0xFFFF
    tries = (blk["reserved"][blk["status"] == 0] >> 16) & 0xFF
    out["min_block_sizes_tried"] = {int(k): int(v) for k, v in zip(*np.unique(tries, return_counts=True))}
    out["walk_share_permille"] = {"p10": int(np.percentile(share, 10)), "p50": int(np.percentile(share, 50)), "p90": int(np.percentile(share, 90)),
                                  "mean": float(share.mean())}
    out["bt_len_mean"] = float(blk["bt_len"][blk["status"] == 0].mean())
    b.free()
    gpu.close()
    print(json.dumps(out))
    open("gpurun_out/r04_exp_block_phases.json", "w").write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
