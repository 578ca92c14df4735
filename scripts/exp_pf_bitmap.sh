#!/bin/bash
# GPU box: the non-empty-k-mer bit table (MMGPU_PF_BITMAP) on the full configs[2] index and on one shard of eight (1/8 of the targets,
# the same 10 000 queries): stage times [similar k-mers, split, replay, ...] and the digest of all hit lists
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
for fam in 2500; do
  for bm in 0 1; do
    echo "== families $fam, MMGPU_PF_BITMAP=$bm"
    MMGPU_PF_BITMAP=$bm python $R/scripts/bench_prefilter.py --families $fam --members 50 --queries 2500 --batch 1024 --steps 2 > /tmp/pf_bm.log 2>&1
    grep '^{' /tmp/pf_bm.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'similar_kmers', 'db_matches', 'hits')})" || tail -5 /tmp/pf_bm.log
  done
done
