# round 6: the unsplit re-run of flagged queries (library, one rank, the binary), then the 2-rank bench on one GPU over gloo at a tenth of
# the size
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_sharded_gpu.py -q -m gpu -x > $O/gpu_tests_sharded.log 2>&1; tail -4 $O/gpu_tests_sharded.log
python -m pytest tests/test_mmseqs_dropin.py -q -m gpu -x -k "sharded_prefilter_reruns or several_device or query_groups" > $O/gpu_tests_dropin.log 2>&1; tail -4 $O/gpu_tests_dropin.log
MMGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --headline-only --steps 3 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; tail -c 1200 $O/bench_2ranks.json; echo; tail -3 $O/bench_2ranks.err
