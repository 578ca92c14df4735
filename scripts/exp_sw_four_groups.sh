#!/bin/bash
# GPU box: three kernel groups (rounds 1-2: single tiles of R >= 26 and the multi-tile queries in one kernel) against four
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "== three groups"; MMGPU_LIB=$R/variants/sw_g3/libmmgpu.so python $R/scripts/exp_sw_groups.py 2>&1 | grep -E "align kernels|checksums|group . done"
for p in default lmmh lmhh llmh lhmh; do
  echo "== four groups, MMGPU_SW_PRIORITY=$p"
  if [ $p = default ]; then unset MMGPU_SW_PRIORITY; else export MMGPU_SW_PRIORITY=$p; fi
  python $R/scripts/exp_sw_groups.py 2>&1 | grep -E "align kernels|checksums|group . done"
done
