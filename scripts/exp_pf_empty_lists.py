"""GPU box experiment: what fraction of the similar k-mers of the headline workload have an EMPTY index list (1M targets, and one
of 8 length-bucket shards)?  They cost a random 8-byte read of the offsets table each (pf_kmers_kernel<true>)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mmseqs2_amd
from mmseqs2_amd import capi, workloads as wl

m = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
gpu = mmseqs2_amd.MMGpu(0)
km16 = m["vtml80_kmer"].astype(np.int16)
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(20000, 50, 10000, seed=10)
qs = wl.split(qres, qoff)[:150]
s3, i3 = capi.host_score_matrix(km16, 3)
queries = [dict(q=q, comp_bias=capi.host_comp_bias(km16, m["vtml80_pback"], q)[0], identity_id=None) for q in qs]
for shards in (1, 8):
    if shards == 1:
        gpu.load_targets(tres, toff, 21)
    else:
        shard_of, local_id, sizes, _ = capi.partition_targets(toff, shards)
        sres, soff, gids = capi.shard_sequences(tres, toff, shard_of, 0)
        gpu.load_targets(sres, soff, 21)
    gpu.pf_build_index(6, 21, True, s3, i3, km16, 112, m["blosum62_ungapped"])
    b = gpu.pf_prepare(queries, 112, max_hits=300, min_diag_score=15, ref_bins=2)
    b.run()
    lists = b.debug("lists")
    n = len(lists)
    empty = int((lists["len"] == 0).sum())
    print("shards %d: %d similar k-mers of %d queries, %d with an empty list (%.1f %%), mean length of the others %.2f"
          % (shards, n, len(qs), empty, 100.0 * empty / max(n, 1), float(lists["len"][lists["len"] > 0].mean())), flush=True)
    b.free()
