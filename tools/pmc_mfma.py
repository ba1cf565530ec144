#!/usr/bin/env python3
"""Hardware-counter view of the MFMA-bound kernels: per kernel, from rocprofv3 --pmc passes
(rocpd sqlite), the matrix-pipe busy fraction and the flops the counters saw, next to the launch count.

    pmc_mfma.py busy.db mops.db [out.json]

busy.db: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
mops.db: --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 (either may be absent)

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the SQ counter sums busy cycles over the
256 CUs x 4 SIMDs of the device, the GRBM counter sums active cycles over the 8 XCDs.  Calibrated on this box with
tools/microbench/mfma_peak (nothing but fp32 MFMAs, 143 - 153 TFLOP/s = 0.91 - 0.97 of peak by its own clock): that
kernel reads 0.119 - 0.123 before the factor of 8, 0.95 - 0.98 with it (profiles/r06_pmc_mfma_calibration.txt).
counter flops = 512 x SQ_INSTS_VALU_MFMA_MOPS_* (one MOP = 512 flops)."""
import json
import sqlite3
import sys


def load(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    res = {}
    for name, cn, avg, n in rows:
        short = name.replace('n2nmn::(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        res.setdefault(short, {'launches': n})[cn] = avg
    return res


def main(busy_db, mops_db, out=None):
    busy, mops = load(busy_db), load(mops_db)
    res = {}
    for k, v in busy.items():
        gui, mf = v.get('GRBM_GUI_ACTIVE'), v.get('SQ_VALU_MFMA_BUSY_CYCLES')
        if not gui or not mf:
            continue
        r = dict(launches=v['launches'], gui_cycles=gui, mfma_busy_cycles=mf, mfma_busy=mf / (gui / 8.0 * 1024.0))
        if v.get('SQ_BUSY_CU_CYCLES'):
            r['cu_busy_cycles'] = v['SQ_BUSY_CU_CYCLES']
        m = mops.get(k, {})
        for key, tag in (('SQ_INSTS_VALU_MFMA_MOPS_F32', 'f32'), ('SQ_INSTS_VALU_MFMA_MOPS_BF16', 'bf16')):
            if m.get(key):
                r['counter_flops_' + tag] = 512.0 * m[key]
        res[k] = r
    for k, r in sorted(res.items(), key=lambda kv: -kv[1]['mfma_busy_cycles'] * kv[1]['launches']):
        print('%-36s n=%6d  mfma_busy %.3f  counter flops per launch: f32 %.4g  bf16 %.4g' % (
            k[:36], r['launches'], r['mfma_busy'], r.get('counter_flops_f32', 0.0), r.get('counter_flops_bf16', 0.0)))
    if out:
        json.dump(res, open(out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main(*sys.argv[1:4])
