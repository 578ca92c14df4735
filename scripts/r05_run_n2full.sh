# BASELINE configs[2] at its full size as TWO ranks on the one GPU (gloo), with the in-run check against the unsplit run
O=gpurun_out/r05n2; mkdir -p $O
export MMGPU_PF_STAGE_GB=8
MMGPU_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-modules --no-nucl --no-align-only > $O/bench_2ranks_one_gpu_gloo_full.json 2> $O/bench_2ranks.err
tail -c 1500 $O/bench_2ranks_one_gpu_gloo_full.json; tail -5 $O/bench_2ranks.err
