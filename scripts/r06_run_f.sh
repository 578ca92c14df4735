O=gpurun_out/r06f; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/lds_order scripts/probes/lds_atomic_order.hip 2>/dev/null
for i in 1 2 3; do /tmp/lds_order; done > $O/lds_atomic_order_probe.txt 2>&1
cat $O/lds_atomic_order_probe.txt
