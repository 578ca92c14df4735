R=${GRAFT_REPO_ROOT:-/root/repo}
for v in base pf_allold; do
  echo "== $v"
  if [ "$v" = base ]; then unset MMGPU_LIB; else export MMGPU_LIB=$R/variants/$v/libmmgpu.so; fi
  timeout 300 python -m pytest $R/tests/test_prefilter_gpu.py -x -q -s -m gpu -k "larger" 2>&1 | grep -v "^  File\|Extension modules" | tail -12 | cut -c1-300
done
