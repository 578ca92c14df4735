# work list of the larger buckets: prefilter GPU suite, headline bench
O=gpurun_out/r06w; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py tests/test_sharded_gpu.py tests/test_profile_query.py -q -m gpu -x > $O/pf_tests.log 2>&1; grep -n "passed\|failed" $O/pf_tests.log
python bench.py --no-cpu-baseline --headline-only --steps 10 --warmup 2 > $O/bench.json 2> $O/bench_err.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06w/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_stages'], d['prefilter']['stage_ms'])
PY
