# instruction counters of the block aligner kernels for scripts/exp_block2.py <families> <queries>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in ${PMC_SETS:-"SQ_INSTS_VALU SQ_INSTS_SALU"}; do
    c=$(echo $C | tr 'A-Z ' 'a-z_')
    rm -rf /tmp/prof_b2p
    timeout 600 rocprofv3 --pmc $C --kernel-trace --stats -d /tmp/prof_b2p -o pmc -- python $R/scripts/exp_block2.py ${1:-2000} ${2:-1000} $3 > $OUT/r06_exp_block2_pmc.log 2>&1
    timeout 120 python $R/scripts/rocprof_summary.py /tmp/prof_b2p/pmc_results.db $OUT/r06_exp_block2_pmc_$c.txt
    grep -i "block\|kernel  " $OUT/r06_exp_block2_pmc_$c.txt | cut -c1-260 | head -8
done
