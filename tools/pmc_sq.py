#!/usr/bin/env python3
"""Per-kernel averages of SQ / GRBM counters from a rocprofv3 --pmc run (rocpd sqlite).
Usage: pmc_sq.py results.db [out.json]"""
import json
import sqlite3
import sys


def main(db, out=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    res = {}
    for name, cn, avg, n in rows:
        short = name.replace('n2nmn::(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        res.setdefault(short, {'launches': n})[cn] = avg
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0) * kv[1]['launches']):
        gui = v.get('GRBM_GUI_ACTIVE')
        mf = v.get('SQ_VALU_MFMA_BUSY_CYCLES')
        extra = ''
        if gui and mf is not None:
            # MFMA busy cycles are summed over the 1024 SIMDs (256 CUs x 4) of the device
            v['mfma_util'] = mf / (gui * 1024.0)
            extra = ' mfma_util %.3f' % v['mfma_util']
        print('%-34s n=%6d %s%s' % (k[:34], v['launches'],
              ' '.join('%s=%.3g' % (c, x) for c, x in sorted(v.items()) if c not in ('launches', 'mfma_util')), extra))
    if out:
        json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:3])
