# kernel durations of the block aligner kernels (new: sw_block2_*, old: sw_block_kernel) for scripts/exp_block2.py <families> <queries>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b2 -o trace -- python $R/scripts/exp_block2.py ${1:-400} ${2:-200} $3 > $OUT/r06_exp_block2_trace.log 2>&1
timeout 120 python $R/scripts/rocprof_summary.py /tmp/prof_b2/trace_results.db $OUT/r06_exp_block2_kernel_stats.txt
grep -i "block\|Kernel\|name" $OUT/r06_exp_block2_kernel_stats.txt | head -20
tail -2 $OUT/r06_exp_block2_trace.log | cut -c1-1500
