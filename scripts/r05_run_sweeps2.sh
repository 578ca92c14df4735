O=gpurun_out/r05sweeps; mkdir -p $O
timeout 420 python scripts/dropin_option_sweep.py device > $O/dropin_option_sweep_device.txt 2>&1; tail -2 $O/dropin_option_sweep_device.txt
timeout 420 python scripts/dropin_workflow_sweep.py device > $O/dropin_workflow_sweep_device.txt 2>&1; tail -2 $O/dropin_workflow_sweep_device.txt
