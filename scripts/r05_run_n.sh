O=gpurun_out/r05n; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --steps 3 --batch 12000 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('batches', 's_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== 8-byte list records; 8 queries checked"
run --check 8
echo "== batches of 1024 queries"
run --batch 1024
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
python scripts/fuzz_prefilter_gpu.py 8 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
python -m pytest tests/test_prefilter_gpu.py tests/test_profile_query.py tests/test_nucl_prefilter.py tests/test_sharded_gpu.py -x -q -m gpu > $O/test_pf.log 2>&1; tail -3 $O/test_pf.log
