O=gpurun_out/r05p; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests_full.log 2>&1; tail -3 $O/gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for N in 2 4; do
MMGPU_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 2 --warmup 1 --pf-families 2000 --pf-queries 1000 --no-cpu-baseline --no-modules --no-nucl --no-align-only > $O/bench_${N}ranks_one_gpu_gloo_tenth.json 2> $O/bench_${N}ranks.err
python -c "
import json
d = json.load(open('$O/bench_${N}ranks_one_gpu_gloo_tenth.json'))
print($N, 'ranks:', d['value'], d['ms_per_step'], d['config']['parallelism'][:60], d.get('parity_vs_unsplit'))"
done
