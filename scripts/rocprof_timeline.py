#!/usr/bin/env python3
"""Concurrency timeline of the kernels matching a substring in a rocprofv3 rocpd database: bursts of overlapping /
back-to-back dispatches (gap < 0.2 ms), per burst the wall span, the summed kernel time and the time during which
fewer than N kernels were resident (the tail).

usage: scripts/rocprof_timeline.py <results.db> <kernel substring> [<out.txt>]
"""
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], sys.argv[2]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels where name like ? order by start",
                     ("%" + pat + "%",)).fetchall()
    bursts, cur, cur_end = [], [], 0
    for r in rows:
        if cur and r[1] - cur_end > 200000:
            bursts.append(cur)
            cur, cur_end = [], 0
        cur.append(r)
        cur_end = max(cur_end, r[2])
    if cur:
        bursts.append(cur)
    out.write("# bursts of kernels matching '%s' in %s\n" % (pat, db))
    out.write("%5s %8s %10s %12s %10s %10s %10s  longest kernels (ms, grid in workgroups)\n" % (
        "burst", "kernels", "span_ms", "sum_kern_ms", "<=1 act", "<=2 act", "<=4 act"))
    for bi, b in enumerate(bursts):
        t0, t1 = min(r[1] for r in b), max(r[2] for r in b)
        ev = sorted([(r[1], 1) for r in b] + [(r[2], -1) for r in b])
        act, last, low = 0, t0, {1: 0, 2: 0, 4: 0}
        for t, d in ev:
            for k in low:
                if act <= k:
                    low[k] += t - last
            last = t
            act += d
        longest = sorted(b, key=lambda r: r[1] - r[2])[:3]
        desc = "; ".join("%s +%.2f..%.2f (%d)" % (r[0].split("sw_kernel")[-1].split("(")[0][:20] if "sw_kernel" in r[0] else r[0][-40:],
                                                  (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, r[3] // max(r[4], 1)) for r in sorted(b, key=lambda r: r[1])[:8])
        out.write("%5d %8d %10.3f %12.3f %10.3f %10.3f %10.3f  %s\n" % (
            bi, len(b), (t1 - t0) / 1e6, sum(r[2] - r[1] for r in b) / 1e6, low[1] / 1e6, low[2] / 1e6, low[4] / 1e6, desc))


if __name__ == "__main__":
    main()
