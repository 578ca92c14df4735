# round 6: the new alignment modes (MMGPU_SW_START_NOT_WORD, mmgpu_sw_reverse_pairs, mmgpu_sw_block_starts) on the device, then the
# headline with the search-semantics step
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_sw_gpu.py -q -m gpu -x -k "not_word or block_starts or block_aligner or min_start" > $O/gpu_tests_sw.log 2>&1; tail -4 $O/gpu_tests_sw.log
MMGPU_TRACE=1 python bench.py --no-cpu-baseline --no-modules --steps 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo; grep "block aligner" $O/bench.err | tail -24
