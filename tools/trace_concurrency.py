#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (rocpd sqlite): GPU busy fraction and average kernel concurrency
in the busiest contiguous window of `--window` ms (default: the multi-stream timed region)."""
import sqlite3
import sys


def main(path, window_ms=100.0):
    cur = sqlite3.connect(path).cursor()
    ev = sorted(cur.execute("select start, end from kernels"))
    t0, t1 = ev[0][0], ev[-1][1]
    w = int(window_ms * 1e6)
    best = None
    # slide in steps of w/4 and keep the window with most kernel time
    import bisect
    starts = [e[0] for e in ev]
    s = t0
    while s + w <= t1:
        i = bisect.bisect_left(starts, s)
        j = bisect.bisect_left(starts, s + w)
        ktime = sum(min(e[1], s + w) - e[0] for e in ev[i:j])
        if best is None or ktime > best[0]:
            best = (ktime, s, i, j)
        s += w // 4
    ktime, s, i, j = best
    # busy time = union of intervals
    busy, cur_end = 0, s
    for a, b in ev[i:j]:
        b = min(b, s + w)
        if a > cur_end:
            busy += b - a; cur_end = b
        elif b > cur_end:
            busy += b - cur_end; cur_end = b
    print('window %.1f ms: %d kernels, busy %.1f %%, average concurrency while busy %.2f, '
          'launch rate %.0f k/s' % (w / 1e6, j - i, 100.0 * busy / w, ktime / max(busy, 1),
                                    (j - i) / (w / 1e9) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0)
