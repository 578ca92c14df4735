O=gpurun_out/r05c; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 1024 --steps 2 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'overflow_queries', 'hits', 'checked_vs_oracle', 'mismatches')})"; }
echo "== order (groups dealt to XCDs) + compact offsets (new default), 12 queries checked against the oracle" > $O/pf_variants.txt
run --check 12 >> $O/pf_variants.txt
echo "== MMGPU_PF_COFS=0" >> $O/pf_variants.txt
MMGPU_PF_COFS=0 run >> $O/pf_variants.txt
echo "== MMGPU_PF_NO_ORDER=1 (compact offsets alone)" >> $O/pf_variants.txt
MMGPU_PF_NO_ORDER=1 run >> $O/pf_variants.txt
cat $O/pf_variants.txt
python -m pytest tests/test_prefilter_gpu.py -x -q -m gpu -k "compact or k7 or stages" > $O/test_pf.log 2>&1; tail -5 $O/test_pf.log
