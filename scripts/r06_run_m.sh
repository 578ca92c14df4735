# round 6: ungapped scoring with passes of 256 or 384 cells (score_chunk) - parity, stage times
O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py tests/test_profile_query.py tests/test_nucl_prefilter.py -q -m gpu -x > $O/gpu_tests_pf.log 2>&1; tail -3 $O/gpu_tests_pf.log | head -2
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 10000 --sort 0 --steps 3 --check 12 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})" > $O/pf_stage.txt 2>&1
cat $O/pf_stage.txt
