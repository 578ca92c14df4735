#!/bin/bash
# GPU box: where the wall time of the patched `mmseqs prefilter` / `align` goes on configs[2] (10k x 1M), MMGPU_TRACE=1
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$(mktemp -d /tmp/mmgpu_trace_XXXX)
cd $W
python - <<PY
import sys; sys.path.insert(0, "$R")
from mmseqs2_amd import workloads as wl
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=${1:-20000}, members=50, n_queries=${2:-10000}, seed=10)
wl.write_fasta("q.fasta", qres, qoff, "q"); wl.write_fasta("t.fasta", tres, toff, "t")
PY
S=$R/oracle/_ref/mmseqs_stock; G=$R/oracle/_ref/mmseqs_mmgpu; T=$(nproc)
# two threads per core of the cgroup's CPU quota when there is one (bench.py's module_seconds does the same)
if [ -r /sys/fs/cgroup/cpu.max ]; then read Q P < /sys/fs/cgroup/cpu.max; if [ "$Q" != max ]; then T2=$(( 2 * Q / P )); [ $T2 -ge 1 ] && [ $T2 -lt $T ] && T=$T2; fi; fi
T=${THREADS:-$T}; echo "threads $T"
$S createdb q.fasta q -v 1; $S createdb t.fasta t -v 1
( time MMGPU_TRACE=1 $G prefilter q t pref -s 5.7 --threads $T -v 3 ) 2>&1 | grep -E "mmgpu|MMGPU|real|Time for|pf_run|stage" | grep -v "matcher\]" | head -40
( time MMGPU_TRACE=1 $G align q t pref aln -a --threads $T -v 3 ) 2>&1 | grep -E "^\[mmgpu|real|Time for" | head -60
