#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs:
they do not fit one pass -- MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    pmc_traffic.py fetch.db write.db out.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced stream (MI355X_MICROARCH.md section HBM), so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is uncalibrated and reported as is."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, avg, n in cur.execute(
            "select kernel_name, avg(value), count(*) from counters_collection "
            "where counter_name = ? group by kernel_name", (counter,)):
        short = name.replace('n2nmn::(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        out[short] = (avg, n)
    return out


def main(fetch_db, write_db, out_path):
    f = per_kernel(fetch_db, 'FETCH_SIZE')
    w = per_kernel(write_db, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, fn = f.get(k, (0.0, 0))
        wk, wn = w.get(k, (0.0, 0))
        res[k] = {'launches': max(fn, wn), 'FETCH_SIZE_KiB_avg': round(fk, 2),
                  'WRITE_SIZE_KiB_avg': round(wk, 2),
                  'read_bytes_per_launch': int(2 * fk * 1024),
                  'write_bytes_per_launch': int(wk * 1024),
                  'hbm_bytes_per_launch': int(2 * fk * 1024 + wk * 1024)}
    json.dump({'note': 'read bytes = 2 x FETCH_SIZE (gfx950 correction), WRITE_SIZE uncalibrated; '
                       'averages per launch over the traced run', 'kernels': res},
              open(out_path, 'w'), indent=1)
    for k, v in res.items():
        print('%-28s n=%6d read=%10.3f MB write=%9.3f MB' % (k, v['launches'],
              v['read_bytes_per_launch'] / 1e6, v['write_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main(*sys.argv[1:4])
