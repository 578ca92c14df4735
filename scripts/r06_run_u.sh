# how much of the step overlaps when two half-batches share the device: 2 ranks on the one GPU, each with half the queries and the whole database
O=gpurun_out/r06u; mkdir -p $O
MMGPU_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --query-groups 2 --headline-only --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_2groups.json 2> $O/bench_2groups_err.txt
tail -c 1500 $O/bench_2groups.json; echo; tail -5 $O/bench_2groups_err.txt
