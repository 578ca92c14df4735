#!/bin/bash
# GPU box: prefilter stage times of library variants (scripts/build_variant.sh) on configs[2]; usage: exp_pf_variants.sh base NAME...
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
for v in "$@"; do
  if [ "$v" = base ]; then unset MMGPU_LIB; else export MMGPU_LIB=$R/variants/$v/libmmgpu.so; fi
  echo "== $v"
  python $R/scripts/bench_prefilter.py --families ${FAMILIES:-20000} --members 50 --queries ${QUERIES:-10000} --batch 1024 --steps 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits')})"
done
