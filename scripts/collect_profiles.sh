#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh'): the rocprofv3 passes behind profiles/.
# Kernel trace and every PMC counter in its own run (counters are never combined with other trace domains).
# Summaries land in gpurun_out/profiles_new/ ; copy the ones to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprof args..., -- , command
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout 900 rocprofv3 "$@" > $OUT/$name.log 2>&1
}
# 1. kernel trace of the default bench command
run trace --kernel-trace --stats -d /tmp/prof_trace -o trace -- python $R/bench.py --no-cpu-baseline
python $R/scripts/rocprof_summary.py /tmp/prof_trace/trace_results.db $OUT/bench_kernel_trace_stats.txt
python $R/scripts/rocprof_timeline.py /tmp/prof_trace/trace_results.db sw_kernel $OUT/bench_sw_timeline.txt
# 2. HBM traffic of the alignment kernels, configs[1] workload, one step
for C in FETCH_SIZE WRITE_SIZE; do
    c=$(echo $C | tr 'A-Z' 'a-z')
    run sw_$c --pmc $C --kernel-trace --stats -d /tmp/prof_sw_$c -o pmc -- python $R/bench.py --no-cpu-baseline --no-search --steps 1 --warmup 0
    python $R/scripts/rocprof_summary.py /tmp/prof_sw_$c/pmc_results.db $OUT/sw_config2_pmc_$c.txt
done
# 3. HBM traffic of the prefilter kernels, configs[2] workload (alignment section reduced to a token size)
for C in FETCH_SIZE WRITE_SIZE; do
    c=$(echo $C | tr 'A-Z' 'a-z')
    run pf_$c --pmc $C --kernel-trace --stats -d /tmp/prof_pf_$c -o pmc -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --queries 20 --targets 2000 --pf-steps 1 --prefilter-only
    python $R/scripts/rocprof_summary.py /tmp/prof_pf_$c/pmc_results.db $OUT/prefilter_config3_pmc_$c.txt
done
ls -la $OUT
