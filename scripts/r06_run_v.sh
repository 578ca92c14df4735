# long sequences on the device: the GPU test (pool of 4096), drop-in test with a 40 000-residue target and a 33 000-residue query
O=gpurun_out/r06v; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py -q -m gpu -x -k "32768" > $O/long_test.log 2>&1; tail -15 $O/long_test.log
python -m pytest tests/test_mmseqs_dropin.py -q -m gpu -x -k "long" > $O/dropin_long.log 2>&1; tail -5 $O/dropin_long.log
