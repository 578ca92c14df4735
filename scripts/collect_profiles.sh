#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh'): the rocprofv3 passes behind profiles/.
# Kernel trace and every PMC counter in its own run (counters are never combined with other trace domains).
# Summaries land in gpurun_out/profiles_new/ ; copy the ones to be judged into profiles/ (prefix r04_).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprof args..., -- , command
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout 900 rocprofv3 "$@" > $OUT/$name.log 2>&1
}
# 1. kernel trace of the default bench command (the headline line's own run: 10 steps of the 10k x 1M search + sections)
run trace --kernel-trace --stats -d /tmp/prof_trace -o trace -- python $R/bench.py --no-cpu-baseline --no-modules
python $R/scripts/rocprof_summary.py /tmp/prof_trace/trace_results.db $OUT/bench_kernel_trace_stats.txt
python $R/scripts/rocprof_timeline.py /tmp/prof_trace/trace_results.db sw_kernel $OUT/bench_sw_timeline.txt
# 2. HBM traffic of the headline's kernels (prefilter + alignment of the hit lists), one timed step after the warm-up
for C in FETCH_SIZE WRITE_SIZE; do
    c=$(echo $C | tr 'A-Z' 'a-z')
    run search_$c --pmc $C --kernel-trace --stats -d /tmp/prof_search_$c -o pmc -- python $R/bench.py --no-cpu-baseline --headline-only --steps 1 --warmup 1
    python $R/scripts/rocprof_summary.py /tmp/prof_search_$c/pmc_results.db $OUT/search_pmc_$c.txt
done
# 3. VALU issue counters of the alignment kernels on the same workload (each counter its own pass)
for C in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do
    c=$(echo $C | tr 'A-Z' 'a-z')
    run sq_$c --pmc $C --kernel-trace --stats -d /tmp/prof_sq_$c -o pmc -- python $R/bench.py --no-cpu-baseline --headline-only --steps 1 --warmup 1
    python $R/scripts/rocprof_summary.py /tmp/prof_sq_$c/pmc_results.db $OUT/search_pmc_$c.txt
done
ls -la $OUT
