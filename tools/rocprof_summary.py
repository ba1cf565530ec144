#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / min / max / total,
plus the launch gaps on the busiest stream.  Usage: rocprof_summary.py results.db [> out.txt]"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
        "from kernels group by name order by sum(end-start) desc"))
    tot = sum(r[5] for r in rows) or 1
    print('# rocprofv3 --kernel-trace summary of %s' % path)
    print('%-72s %8s %10s %10s %10s %11s %6s' % ('kernel', 'calls', 'avg_us', 'min_us', 'max_us',
                                                 'total_ms', '%'))
    for name, n, avg, mn, mx, sm in rows:
        short = name.replace('n2nmn::(anonymous namespace)::', '').split('(')[0]
        print('%-72s %8d %10.2f %10.2f %10.2f %11.3f %6.1f' % (short[:72], n, avg / 1e3, mn / 1e3,
                                                               mx / 1e3, sm / 1e6, 100.0 * sm / tot))
    ev = list(cur.execute("select start, end from kernels order by start"))
    if len(ev) > 1:
        gaps = sorted(max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1))
        span = ev[-1][1] - ev[0][0]
        print('# %d dispatches; kernel time %.3f ms of %.3f ms span; median gap %.2f us, p90 %.2f us'
              % (len(ev), tot / 1e6, span / 1e6, gaps[len(gaps) // 2] / 1e3,
                 gaps[int(len(gaps) * 0.9)] / 1e3))


def by_grid(path, pattern):
    """per (kernel, grid size) rows for kernels whose name contains `pattern`"""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gcol = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    q = ("select name, %s, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
         "where name like ? group by name, %s order by name, %s" % (gcol, gcol, gcol))
    print('# per grid size, kernels matching %r (grid column %s)' % (pattern, gcol))
    for name, g, n, avg, mn, mx in cur.execute(q, ('%' + pattern + '%',)):
        short = name.replace('n2nmn::(anonymous namespace)::', '').split('(')[0]
        print('%-40s grid_x %8s calls %6d avg %9.2f min %9.2f max %9.2f us' % (
            short[:40], g, n, avg / 1e3, mn / 1e3, mx / 1e3))


if __name__ == '__main__':
    main(sys.argv[1])
    for pat in sys.argv[2:]:
        by_grid(sys.argv[1], pat)
