# the bench line again (bench.py's stage traffic now counts every dispatch of a run; no device source changed) and the N-rank runs of
# the final build on the one GPU
O=gpurun_out/r05final2; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json; echo
for N in 2 4; do
MMGPU_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 2 --warmup 1 --pf-families 2000 --pf-queries 1000 --no-cpu-baseline --no-modules --no-nucl --no-align-only > $O/bench_${N}ranks_one_gpu_gloo_tenth.json 2> $O/bench_${N}ranks.err
python -c "
import json
d = json.load(open('$O/bench_${N}ranks_one_gpu_gloo_tenth.json'))
print($N, 'ranks:', d['value'], d['ms_per_step'], d['config']['parallelism'][:60], d.get('parity_vs_unsplit', {}).get('equal'))"
done
