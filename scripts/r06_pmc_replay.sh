# counters of the prefilter kernels (the replay kernel first of all) for one batch of 10 000 queries through scripts/bench_prefilter.py;
# every counter set in its own pass
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PMC_TAG:-r06_pmc_replay}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MMGPU_WL_CACHE=/tmp/mmgpu_wl
for C in ${PMC_SETS:-"SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"}; do
    c=$(echo $C | tr 'A-Z ' 'a-z_')
    rm -rf /tmp/prof_rp
    timeout 600 rocprofv3 --pmc $C --kernel-trace --stats -d /tmp/prof_rp -o pmc -- python $R/scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 10000 --sort 0 --steps 1 > $OUT/run.log 2>&1
    timeout 120 python $R/scripts/rocprof_summary.py /tmp/prof_rp/pmc_results.db $OUT/pmc_$c.txt
    grep -i "pf_replay\|pf_split\|kernel  " $OUT/pmc_$c.txt | cut -c1-200 | head -8
done
