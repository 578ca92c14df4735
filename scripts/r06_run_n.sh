# round 6: similar k-mers with k = 5 on the device; makemmgpudb; the option sweep of the binaries
O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests/test_prefilter_gpu.py -q -m gpu -x -k "k5 or k7 or golden or stages" > $O/gpu_tests_k5.log 2>&1; tail -3 $O/gpu_tests_k5.log
python -m pytest tests/test_mmseqs_dropin.py -q -m gpu -x -k "persisted or examples" > $O/gpu_tests_dropin.log 2>&1; tail -3 $O/gpu_tests_dropin.log
python scripts/dropin_option_sweep.py > $O/dropin_option_sweep_device.txt 2> $O/sweep_err.txt; grep -v "device  *identical" $O/dropin_option_sweep_device.txt | head -20; tail -3 $O/sweep_err.txt
