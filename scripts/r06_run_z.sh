O=gpurun_out/r06z; mkdir -p $O
python bench.py --no-cpu-baseline --headline-only --steps 10 --warmup 2 > $O/bench.json 2> $O/bench_err.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06z/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_stages'], d['prefilter']['stage_ms'], d.get('ms_per_step_search_semantics'))
PY
