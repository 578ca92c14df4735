O=gpurun_out/r06e; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 10000 --sort 0 --steps 3 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== default build"
run --check 12
echo "== replay statistics"
MMGPU_LIB=$PWD/variants/replay_stats/libmmgpu.so run
grep "replay stats" $O/err.txt | tail -1
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
