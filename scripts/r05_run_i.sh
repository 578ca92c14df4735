O=gpurun_out/r05i; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --steps 2 --batch 12000 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('batches', 's_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits')})"; }
{
echo "== one batch, MMGPU_PF_STAGE_GB=16 (7 stage chunks)"
run
echo "== MMGPU_PF_STAGE_GB=40"
MMGPU_PF_STAGE_GB=40 run
echo "== MMGPU_PF_STAGE_GB=120 (one chunk)"
MMGPU_PF_STAGE_GB=120 run
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
