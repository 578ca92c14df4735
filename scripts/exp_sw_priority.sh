#!/bin/bash
# GPU box: stream priorities of the three alignment kernel groups (MMGPU_SW_PRIORITY = one of h / m / l per group S, M, L; default "lmh")
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in lmh lhm lhh mhh hhh mhl; do
  echo "== MMGPU_SW_PRIORITY=$p"
  MMGPU_SW_PRIORITY=$p python $R/scripts/exp_sw_groups.py 2>&1 | grep -E "align kernels|checksums|group . done"
done
