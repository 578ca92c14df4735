# round 6: batches beyond 2^32 entries / slots, persisted-layout checks, translated-search timeline
O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py tests/test_db_file.py tests/test_sharded_gpu.py -q -m gpu -x --durations=5 > $O/gpu_tests_a.log 2>&1; tail -12 $O/gpu_tests_a.log
python -m pytest tests/test_mmseqs_dropin.py tests/test_server_mode.py -q -m gpu -x > $O/gpu_tests_b.log 2>&1; tail -5 $O/gpu_tests_b.log
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
python scripts/bench_prefilter.py --families 20000 --members 50 --queries 13000 --batch 13000 --sort 0 --steps 2 --check 12 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches', 'entries', 'tiles')})" > $O/pf_13000.txt 2>&1
cat $O/pf_13000.txt; tail -3 $O/err.txt
python scripts/search_timeline.py 10 --translated > $O/translated_search_timeline.txt 2>&1; head -30 $O/translated_search_timeline.txt
