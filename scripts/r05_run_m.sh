O=gpurun_out/r05m; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --steps 3 --batch 12000 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('batches', 's_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== compact offset table in 16-byte blocks (12 k-mers, one load per look-up); 8 queries checked"
run --check 8
echo "== 32-byte blocks (28 k-mers, two loads)"
MMGPU_LIB=$PWD/variants/cofs32/libmmgpu.so run
echo "== 16-byte blocks again"
run
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
python -m pytest tests/test_prefilter_gpu.py -x -q -m gpu -k "compact or k7 or stages or golden" > $O/test_pf.log 2>&1; tail -3 $O/test_pf.log
