# round 6: prefetch depth of the replay kernel (rounds in flight per wavefront)
O=gpurun_out/r06h; mkdir -p $O
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
run() { python scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 10000 --sort 0 --steps 3 "$@" 2>>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('s_per_pass', 'stage_ms', 'lists_crc32', 'hits')})"; }
{
echo "== default (4 rounds in flight)"; run
for v in ${VARIANTS:-replay_pd1 replay_pd2 replay_pd8}; do echo "== $v"; MMGPU_LIB=$PWD/variants/$v/libmmgpu.so run; done
} > $O/pf_variants.txt 2>&1
cat $O/pf_variants.txt
