#!/usr/bin/env python3
"""Durations of the recurrent-step launches of ONE pass in launch order, from a rocprofv3
--kernel-trace database (rocpd sqlite).  Usage: lstm_step_trace.py results.db [pass index]"""
import sqlite3
import sys


def main(path, which=2):
    cur = sqlite3.connect(path).cursor()
    ev = list(cur.execute("select name, start, end from kernels order by start"))
    short = lambda n: n.replace('n2nmn::(anonymous namespace)::', '').split('(')[0].replace('void ', '')   # noqa: E731
    # a pass starts at enc_prepare_kernel
    starts = [i for i, e in enumerate(ev) if 'enc_prepare' in e[0]]
    if len(starts) <= which + 1:
        which = max(0, len(starts) - 2)
    lo, hi = starts[which], starts[which + 1]
    t0 = ev[lo][1]
    print('# pass %d: %d dispatches, span %.1f us' % (which, hi - lo, (ev[hi - 1][2] - t0) / 1e3))
    for i in range(lo, hi):
        n, s, e = ev[i]
        gap = (s - ev[i - 1][2]) / 1e3 if i > lo else 0.0
        print('%4d %-40s start %9.1f  dur %7.2f  gap %6.2f' % (i - lo, short(n)[:40], (s - t0) / 1e3,
                                                               (e - s) / 1e3, gap))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
