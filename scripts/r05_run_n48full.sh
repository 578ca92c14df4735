# BASELINE configs[2] at its full size as FOUR (2 x 2) and EIGHT (4 x 2) ranks on the one GPU (gloo), with the in-run check against the unsplit run
O=gpurun_out/r05n48; mkdir -p $O
export MMGPU_PF_STAGE_GB=4
for N in 4 8; do
MMGPU_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2954$N bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline --no-modules --no-nucl --no-align-only > $O/bench_${N}ranks_one_gpu_gloo_full.json 2> $O/bench_${N}ranks.err
python -c "
import json
d = json.load(open('$O/bench_${N}ranks_one_gpu_gloo_full.json'))
print($N, 'ranks:', d['value'], d['ms_per_step'], d['config']['parallelism'][:60], d.get('parity_vs_unsplit'))" || tail -5 $O/bench_${N}ranks.err
done
