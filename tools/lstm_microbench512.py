#!/usr/bin/env python3
"""Fixed cost of the recurrent step at super-bucket size (512 rows, 2048 workgroups): shipped /
loads only / neither / empty kernel, back to back (GPU only)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from n2nmn_amd import synth, _lib
from n2nmn_amd.engine import Engine
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

d = Dims(N=1024)
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
eng.load_weights(synth.make_weights(Dims(), seed=0))
names = {0: 'k-split 64x16', 2: 'loads only', 3: 'mfma only', 4: 'neither', 5: 'empty kernel',
         23: 'LDS tile, 3 stages', 25: 'LDS tile, 5 stages', 24: 'LDS tile, 4 stages', 26: 'LDS tile, 6 stages',
         34: 'LDS tile 4st, no DMA', 44: 'LDS tile 4st, no MFMA', 54: 'LDS tile 4st, DMA+barrier only',
         64: 'LDS tile 4st, no DMA waits', 74: 'LDS tile 4st, cache-hot DMA'}
VARIANTS = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 24, 23, 26, 2, 3, 4, 5]
for N in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (512, 1024, 256, 128, 64)):
    for rows in (64,):
        for v in VARIANTS:
            if v >= 20 and N < 128:
                continue
            us = C.c_double()
            _lib.check(eng._lib.n2nmn_debug_lstm_bench(eng._ctx, v, rows, 2, N, 200,
                                                       C.byref(us), eng.stream()))
            print('N=%d rows/wg=%d %-14s %7.2f us/launch' % (N, rows, names[v], us.value), flush=True)
