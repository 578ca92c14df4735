O=gpurun_out/r05o; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --steps 3 --batch 12000 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('batches', 's_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== two rounds of similar k-mers per trip (default); 8 queries checked"
run --check 8
echo "== one round per trip (as before)"
MMGPU_LIB=$PWD/variants/emit1/libmmgpu.so run
echo "== three rounds per trip"
MMGPU_LIB=$PWD/variants/emit3/libmmgpu.so run
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
