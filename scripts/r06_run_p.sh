O=gpurun_out/r06p; mkdir -p $O
for i in 1 2; do
python bench.py --no-cpu-baseline --headline-only > $O/bench_headline_$i.json 2> $O/bench_err_$i.txt
python - <<PY
import json
d = json.loads(open('gpurun_out/r06p/bench_headline_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_search_semantics'), d.get('search_semantics', {}).get('stages_ms'), d.get('search_semantics', {}).get('parity', {}).get('queries_with_a_differing_record'))
PY
done
