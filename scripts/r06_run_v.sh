# long targets on the device: the new GPU test, the prefilter GPU suite, the drop-in test with a 40 000-residue target
O=gpurun_out/r06v; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py -q -m gpu -x -k "32768" > $O/long_test.log 2>&1; tail -15 $O/long_test.log
python -m pytest tests/test_prefilter_gpu.py tests/test_sharded_gpu.py -q -m gpu -x > $O/pf_tests.log 2>&1; tail -3 $O/pf_tests.log
python -m pytest tests/test_mmseqs_dropin.py -q -m gpu -x -k "long" > $O/dropin_long.log 2>&1; tail -5 $O/dropin_long.log
