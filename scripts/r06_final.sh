# round 6, final collection on one box: GPU suite, smoke, the profile passes of this build (copied into profiles/ so that the bench line of the same
# run reports their counter traffic), the default bench line, 2 and 4 ranks on the one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu --durations=10 > $O/gpu_tests_full.log 2>&1; tail -4 $O/gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash scripts/collect_profiles.sh > $O/collect.log 2>&1; tail -3 $O/collect.log
cd $R
for f in bench_kernel_trace_stats bench_sw_timeline search_pmc_fetch_size search_pmc_write_size search_pmc_sq_insts_valu search_pmc_sq_active_inst_valu search_pmc_sq_busy_cycles search_pmc_sq_wave_cycles; do
  cp gpurun_out/profiles_new/$f.txt profiles/r06_$f.txt
done
python bench.py > $O/bench_n1.json 2> $O/bench_err.txt; tail -c 600 $O/bench_n1.json; echo
for N in 2 4; do
  MMGPU_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $N --headline-only --no-cpu-baseline > $O/bench_${N}ranks_one_gpu_gloo_full.json 2> $O/bench_${N}ranks_err.txt
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_${N}ranks_one_gpu_gloo_full.json').read().strip().splitlines()[-1])
    print($N, d['value'], d['ms_per_step'], d.get('parity_vs_unsplit', {}).get('equal'), d.get('parity_vs_unsplit', {}).get('queries_compared'))
except Exception as e:
    print('N=$N failed', e)
PY
done
