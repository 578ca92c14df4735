# round 6: after the prune - the whole GPU suite, then the default bench line
O=gpurun_out/r06j; mkdir -p $O
python -m pytest tests -q -m gpu -x --durations=15 > $O/gpu_tests.log 2>&1; tail -25 $O/gpu_tests.log
python bench.py > $O/bench_n1.json 2> $O/bench_err.txt; tail -c 1500 $O/bench_n1.json
