O=gpurun_out/r05f; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 1024 --steps 2 "$@" 2>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
{
echo "== default (list records stored non-temporal)"
run
echo "== list records stored plain"
MMGPU_LIB=$PWD/variants/lists_plain/libmmgpu.so run
echo "== split tiles stored non-temporal"
MMGPU_LIB=$PWD/variants/split_nt/libmmgpu.so run
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
