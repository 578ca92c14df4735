# round-6 first package: the GPU suite and the bench line of the build with the four-pairs-per-wavefront block aligner
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json; echo; tail -5 $O/bench_n1.err
