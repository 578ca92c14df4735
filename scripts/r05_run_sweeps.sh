# regression evidence for the final build: GPU fuzz of every prefilter stage against the oracle, option and workflow sweeps of the patched
# binary against the stock one on the device
O=gpurun_out/r05sweeps; mkdir -p $O
timeout 300 python scripts/fuzz_prefilter_gpu.py 8 > $O/fuzz_prefilter_gpu.log 2>&1; tail -2 $O/fuzz_prefilter_gpu.log
timeout 420 python scripts/dropin_option_sweep.py device > $O/dropin_option_sweep_device.txt 2>&1; tail -3 $O/dropin_option_sweep_device.txt
timeout 420 python scripts/dropin_workflow_sweep.py device > $O/dropin_workflow_sweep_device.txt 2>&1; tail -3 $O/dropin_workflow_sweep_device.txt
