"""Block aligner, two pairs per wavefront (block2_kernel.hip) against the one-pair-per-wavefront kernel (block_kernel.hip, forced with
MMGPU_BLOCK_FIRST_TIER=0) on the word == 1 pairs of a configs[2]-shaped workload: every field and every string equal, the
block lists equal, call times of both, and the starts-only mode.  usage: exp_block2.py [families] [queries]"""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (HIP runtime first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
import mmseqs2_amd
from mmseqs2_amd import capi
from mmseqs2_amd import workloads as wl


def main():
    fams = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    only_new = len(sys.argv) > 3 and sys.argv[3] == "new"      # timing runs: skip the comparison with block_kernel.hip
    mats = dict(np.load("tests/golden/matrices.npz"))
    mat = mats["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    (qres, qoff), (tres, toff), fam_t, fam_q = wl.config3_prefilter(fams, 50, nq, seed=10)
    qs = wl.split(qres, qoff)
    gpu = mmseqs2_amd.MMGpu(0)
    gpu.load_targets(tres, toff, 21)
    order = np.argsort(fam_t, kind="stable")
    starts = np.searchsorted(fam_t[order], np.arange(fams + 1))
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
    os.environ.pop("MMGPU_BLOCK_FIRST_TIER", None)
    for rep in range(3):
        new, new_s = b.block_backtrace(idx)
        out["new_call_s_%d" % rep] = round(b.last_block_call_s, 4)
    out["new_tiers"] = b.block_tiers()
    for rep in range(2):
        st, _ = b.block_backtrace(idx, mode="starts")
        out["starts_only_call_s_%d" % rep] = round(b.last_block_call_s, 4)
    for rep in range(2):
        ns, _ = b.block_backtrace(idx, mode="no_strings")
        out["no_strings_call_s_%d" % rep] = round(b.last_block_call_s, 4)
    if only_new:
        out["status_new"] = {int(k): int(v) for k, v in zip(*np.unique(new["status"], return_counts=True))}
        print(json.dumps(out))
        return
    _, new_lists = b.block_growth(idx[:20000], cap=512)
    # rows of the shift blocks (every eight residues of progress is one), and blocks per pair
    hist, nblk = {}, []
    for a in new_lists:
        if not len(a):
            continue
        h, w, right = a[:, 2], a[:, 3], a[:, 4]
        rows = np.where(right == 1, h, w)[((right == 1) & (w == 8)) | ((right == 0) & (h == 8))]
        for r, c in zip(*np.unique(rows, return_counts=True)):
            hist[int(r)] = hist.get(int(r), 0) + int(c)
        nblk.append(len(a))
    out["shift_block_rows_histogram"] = hist
    out["blocks_per_pair_mean"] = float(np.mean(nblk)) if nblk else 0.0
    os.environ["MMGPU_BLOCK_FIRST_TIER"] = "0"
    for rep in range(2):
        old, old_s = b.block_backtrace(idx)
        out["old_call_s_%d" % rep] = round(b.last_block_call_s, 4)
    _, old_lists = b.block_growth(idx[:20000], cap=512)
    os.environ.pop("MMGPU_BLOCK_FIRST_TIER", None)
    out["status_new"] = {int(k): int(v) for k, v in zip(*np.unique(new["status"], return_counts=True))}
    out["status_old"] = {int(k): int(v) for k, v in zip(*np.unique(old["status"], return_counts=True))}
    bad = {}
    for f in ("status", "q_start", "t_start", "ident", "bt_len"):
        bad[f] = int((new[f] != old[f]).sum())
    bad["strings"] = int(sum(1 for a, c in zip(new_s, old_s) if a != c))
    bad["starts_only_status"] = int((st["status"] != old["status"]).sum())
    bad["starts_only_q_start"] = int((st["q_start"] != old["q_start"]).sum())
    bad["starts_only_t_start"] = int((st["t_start"] != old["t_start"]).sum())
    bad["no_strings_ident"] = int((ns["ident"] != old["ident"]).sum())
    bad["no_strings_bt_len"] = int((ns["bt_len"] != old["bt_len"]).sum())
    bad["block_lists"] = int(sum(1 for a, c in zip(new_lists, old_lists) if a.shape != c.shape or not np.array_equal(a, c)))
    out["differing"] = bad
    ex = []
    for k, (a, c) in enumerate(zip(new_lists, old_lists)):
        if a.shape != c.shape or not np.array_equal(a, c):
            m = min(len(a), len(c))
            d = [int(x) for x in np.nonzero((a[:m] != c[:m]).any(axis=1))[0][:2]]
            ex.append(dict(k=k, n_new=len(a), n_old=len(c), rows=d, new=[a[x].tolist() for x in d], old=[c[x].tolist() for x in d],
                           max_new=int(a[:, 2:4].max()) if len(a) else 0, max_old=int(c[:, 2:4].max()) if len(c) else 0))
    out["differing_lists"] = ex[:3] + ex[len(ex) // 2:len(ex) // 2 + 3]
    out["differing_lists_same_length"] = sum(1 for e in ex if e["n_new"] == e["n_old"])
    out["dbg"] = [dict(k=e["k"], target=int(res[idx[e["k"]]]["score"]), first=int(st[e["k"]]["reserved"]) & 0xFFFF, steps=int(st[e["k"]]["reserved"]) >> 16, d0=int(st[e["k"]]["ident"]), d1=int(st[e["k"]]["bt_len"]), pair=int(idx[e["k"]]), q_end=int(res[idx[e["k"]]]["q_end"]), t_end=int(res[idx[e["k"]]]["t_end"])) for e in out["differing_lists"]]
    first = [int(k) for k in np.nonzero((new["q_start"] != old["q_start"]) | (new["t_start"] != old["t_start"]) | (new["status"] != old["status"]))[0][:5]]
    out["first_differing"] = [dict(k=k, pair=int(idx[k]), new=[int(new[k][f]) for f in ("status", "q_start", "t_start", "ident", "bt_len")],
                                   old=[int(old[k][f]) for f in ("status", "q_start", "t_start", "ident", "bt_len")],
                                   q_end=int(res[idx[k]]["q_end"]), t_end=int(res[idx[k]]["t_end"]), score=int(res[idx[k]]["score"])) for k in first]
    b.free()
    gpu.close()
    print(json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/r06_exp_block2.json", "w").write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
