R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trace
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o trace -- python $R/bench.py --no-cpu-baseline --no-modules > $OUT/trace.log 2>&1
python $R/scripts/rocprof_summary.py /tmp/prof_trace/trace_results.db $OUT/bench_kernel_trace_stats.txt
python $R/scripts/rocprof_timeline.py /tmp/prof_trace/trace_results.db sw_kernel $OUT/bench_sw_timeline.txt
