O=gpurun_out/r06q; mkdir -p $O
MMGPU_TRACE=1 python bench.py --no-cpu-baseline --headline-only --steps 2 --warmup 1 > $O/bench.json 2> $O/bench_err.txt
grep "handed on\|the longest eighth\|beside the skewed\|skewed form" $O/bench_err.txt | tail -8
