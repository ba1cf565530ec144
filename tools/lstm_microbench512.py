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

d = Dims(N=512)
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
eng.load_weights(synth.make_weights(Dims(), seed=0))
names = {0: 'shipped', 2: 'loads only', 3: 'mfma only', 4: 'neither', 5: 'empty kernel'}
for N in (512, 256, 64):
    for rows in (64, 32):
        for v in (0, 2, 3, 4, 5):
            us = C.c_double()
            _lib.check(eng._lib.n2nmn_debug_lstm_bench(eng._ctx, v, rows, 2, N, 200,
                                                       C.byref(us), eng.stream()))
            print('N=%d rows/wg=%d %-14s %7.2f us/launch' % (N, rows, names[v], us.value), flush=True)
