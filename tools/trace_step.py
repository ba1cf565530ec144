#!/usr/bin/env python3
"""List the kernels of ONE step (the last complete one) of a rocprofv3 --kernel-trace run in start
order with their grid and duration.  Usage: trace_step.py results.db marker_kernel [filter]"""
import sqlite3
import sys


def main(db, marker, filt=None):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    g = [c for c in ('grid_x', 'grid_y', 'grid_z') if c in cols] or \
        [c for c in ('grid_size_x', 'grid_size_y', 'grid_size_z') if c in cols]
    w = [c for c in ('workgroup_x', 'workgroup_size_x') if c in cols]
    rows = list(cur.execute("select name, start, end, %s, %s, stream_id from kernels order by start"
                            % (', '.join(g), w[0] if w else '0')))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print('marker not found twice'); return
    lo, hi = marks[-2], marks[-1]
    t0 = rows[lo][1]
    for r in rows[lo:hi]:
        name = r[0].replace('n2nmn::(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if filt and filt not in name:
            continue
        wg = r[6] or 1
        print('%9.1f us  %-34s grid %5d x %3d x %3d  %8.2f us  stream %s' % (
            (r[1] - t0) / 1e3, name[:34], r[3] // wg, r[4], r[5], (r[2] - r[1]) / 1e3, r[7]))
    print('step span %.1f us' % ((rows[hi][1] - t0) / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:4])
