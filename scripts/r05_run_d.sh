O=gpurun_out/r05d; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 1024 --steps 2 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== default: order mode 1 (units of 20 rows dealt to the XCDs, sorted by last then first 3-mer) + compact offsets + dup-key loop in the replay; 12 queries checked"
run --check 12
echo "== MMGPU_PF_ORDER_MODE=2 (contiguous rows per XCD, sorted by both 3-mers)"
MMGPU_PF_ORDER_MODE=2 run
echo "== MMGPU_PF_ORDER_MODE=3 (units dealt, no sort by the first 3-mer)"
MMGPU_PF_ORDER_MODE=3 run
echo "== replay with the 12-bit lane matching (round 4)"
MMGPU_LIB=$PWD/variants/match12/libmmgpu.so run
echo "== replay statistics"
MMGPU_LIB=$PWD/variants/replay_stats/libmmgpu.so run
grep "replay stats" $O/err.txt | tail -1
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
