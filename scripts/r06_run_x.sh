R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_x -o trace -- python $R/scripts/bench_prefilter.py --families 20000 --members 50 --queries 10000 --batch 10000 --sort 0 --steps 2 > $O/trace.log 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect('/tmp/prof_x/trace_results.db')
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
# the last pass: from the last pf_split kernel on
last_split = max(i for i, r in enumerate(rows) if 'pf_split' in r[0])
t0 = rows[last_split][1]
out = open('$O/pass_timeline.txt', 'w')
prev_end = None
for r in rows[last_split - 3:]:
    nm = r[0].split('::')[-1].split('(')[0]
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0.0
    out.write("%-40s start %10.3f ms  dur %9.3f ms  gap before %8.1f us  grid %d\n" % (nm, (r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, gap, r[3] // max(r[4], 1)))
    prev_end = r[2]
out.close()
print(open('$O/pass_timeline.txt').read())
PY
