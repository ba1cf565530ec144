#!/usr/bin/env python3
"""Decompose the fused LSTM step: times kernel variants back-to-back (GPU only)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from n2nmn_amd import synth, _lib
from n2nmn_amd.engine import Engine
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

d = Dims()
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
eng.load_weights(synth.make_weights(d, seed=0))
names = {6: 'weight loads only', 7: 'state loads only', 0: 'shipped', 1: 'loads pinned first', 2: 'loads only', 3: 'mfma only', 4: 'neither',
         5: 'empty kernel'}
for njobs in (2, 1):
    for rows in (64, 32):
        for v in (0, 1, 2, 4, 5):
            us = C.c_double()
            _lib.check(eng._lib.n2nmn_debug_lstm_bench(eng._ctx, v, rows, njobs, 64, 400,
                                                       C.byref(us), eng.stream()))
            print('jobs=%d rows/wg=%d variant=%d %-20s %7.2f us/launch' % (njobs, rows, v, names[v % 10] + (' [row-major h]' if v >= 10 else ''), us.value))
