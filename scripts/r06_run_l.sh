# round 6: replay kernel with its rounds really in flight (named slots, unconditional loads: vmcnt(3) instead of vmcnt(0)) - parity, stage times,
# then the profile passes of this build
O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py -q -m gpu -x > $O/gpu_tests_pf.log 2>&1; tail -3 $O/gpu_tests_pf.log | head -2
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 10000 --sort 0 --steps 3 --check 12 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})" > $O/pf_stage.txt 2>&1
cat $O/pf_stage.txt
bash scripts/collect_profiles.sh > $O/collect.log 2>&1; tail -12 $O/collect.log
