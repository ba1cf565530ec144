#!/usr/bin/env python3
"""One variant of the recurrent step at one size, for rocprofv3 --pmc / --kernel-trace (GPU only):
    python tools/lstm_tile_probe.py N variant [iters]      (variants: see lstm_microbench512.py)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from n2nmn_amd import synth, _lib
from n2nmn_amd.engine import Engine
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

N, v = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
eng = Engine(Dims(N=max(N, 64)), Assembler(list(CLEVR_MODULE_NAMES)))
eng.load_weights(synth.make_weights(Dims(), seed=0))
us = C.c_double()
_lib.check(eng._lib.n2nmn_debug_lstm_bench(eng._ctx, v, 64, 2, N, iters, C.byref(us), eng.stream()))
print('N=%d variant %d: %.2f us/launch' % (N, v, us.value), flush=True)
