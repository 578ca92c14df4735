# round 6: block aligner launch 1 cut in two (hand-ons of the longest eighth start early); option sweep on the device
O=gpurun_out/r06o; mkdir -p $O
python -m pytest tests/test_sw_gpu.py -q -m gpu -x -k "block" > $O/gpu_tests_block.log 2>&1; tail -3 $O/gpu_tests_block.log
MMGPU_TRACE=1 python bench.py --no-cpu-baseline --headline-only > $O/bench_headline.json 2> $O/bench_err.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06o/bench_headline.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_search_semantics'), d.get('search_semantics', {}).get('stages_ms'), d.get('search_semantics', {}).get('parity'))
PY
grep "block aligner\]" $O/bench_err.txt | tail -12
python scripts/dropin_option_sweep.py device > $O/dropin_option_sweep_device.txt 2> $O/sweep_err.txt; grep -v "device  *identical" $O/dropin_option_sweep_device.txt | head -20
