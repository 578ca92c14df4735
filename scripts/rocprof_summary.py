#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (gpurun_out/<dir>/*_results.db) into the plain-text per-kernel summary that
is committed under profiles/ (rocprofv3 --kernel-trace --stats writes only the .db in this ROCm build).

usage: scripts/rocprof_summary.py <results.db> [<out.txt>]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out.write("# rocprofv3 --kernel-trace --stats summary (from %s)\n" % db)
    try:      # which kernels this was taken with (scripts/csrc_digest.py)
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from csrc_digest import csrc_digest
        out.write("# csrc_digest: %s\n" % csrc_digest())
    except Exception:
        pass
    out.write("%-90s %6s %14s %14s %14s %14s %7s %5s %5s %5s %7s %10s %5s\n" % (
        "kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "agpr", "sgpr", "lds_B", "grid_x", "wg_x"))
    for r in rows:
        out.write("%-90s %6d %14d %14.0f %14d %14d %7.2f %5d %5d %5d %7d %10d %5d\n" % (
            r[0][:90], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[11], r[12]))
    try:
        pmc = c.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in pmc.description]
        rows = c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                         "group by kernel_name, counter_name").fetchall() if "counter_name" in cols and "kernel_name" in cols else []
        if rows:
            out.write("\n# PMC counters (sum / mean per dispatch)\n")
            for r in rows:
                out.write("%-80s %-28s n=%d sum=%.6g mean=%.6g\n" % (r[0][:80], r[1], r[2], r[3], r[4]))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main()
