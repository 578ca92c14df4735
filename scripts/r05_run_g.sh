O=gpurun_out/r05g; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --steps 2 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('batches', 's_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== compact split words, batches of 1024 queries; 12 queries checked"
run --batch 1024 --check 12
echo "== compact split words, ONE batch of 10 000 queries (what bench.py's headline runs)"
run --batch 12000
echo "== the same without order / compact offsets"
MMGPU_PF_NO_ORDER=1 MMGPU_PF_COFS=0 run --batch 12000
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
python -m pytest tests/test_prefilter_gpu.py -x -q -m gpu > $O/test_pf.log 2>&1; tail -3 $O/test_pf.log
python scripts/fuzz_prefilter_gpu.py 6 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
