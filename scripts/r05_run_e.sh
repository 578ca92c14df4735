O=gpurun_out/r05e; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 1024 --steps 2 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== default (contiguous order, compact offsets, dup-key loop, segment-free round, direct ballots); 12 queries checked"
run --check 12
echo "== MMGPU_PF_NO_ORDER=1 MMGPU_PF_COFS=0"
MMGPU_PF_NO_ORDER=1 MMGPU_PF_COFS=0 run
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
python scripts/fuzz_prefilter_gpu.py 8 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
python -m pytest tests/test_prefilter_gpu.py tests/test_profile_query.py tests/test_nucl_prefilter.py -x -q -m gpu > $O/test_pf.log 2>&1; tail -3 $O/test_pf.log
