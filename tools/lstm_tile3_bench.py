#!/usr/bin/env python3
"""Recurrent step at super-bucket sizes: the exact-fp32 LDS-DMA tile (lstm_tile_kernel) against the
split-operand bf16 tile (lstm_tile3_kernel) and its ablations, back-to-back launches of an encoder-style
step (layer-0 step + layer-1 step, every row active = the decoder's regime), HIP events.  GPU only.
Usage: python tools/lstm_tile3_bench.py [variants] [rows]"""
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
eng.set_mode('throughput_bf16x3')
names = {24: 'fp32 LDS tile, 4 stages', 34: 'fp32 tile, no DMA', 44: 'fp32 tile, no MFMA',
         1403: 'bf16x3 64 rows, 3 stages', 1404: 'bf16x3 64 rows, 4 stages', 1413: 'bf16x3 64 rows, no DMA',
         1423: 'bf16x3 64 rows, no MFMA', 1433: 'bf16x3 64 rows, DMA only',
         1803: 'bf16x3 128 rows, 3 stages', 1804: 'bf16x3 128 rows, 4 stages', 1814: 'bf16x3 128 rows, no DMA',
         1824: 'bf16x3 128 rows, no MFMA', 1834: 'bf16x3 128 rows, DMA only'}
VARIANTS = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else \
    [24, 1804, 1814, 1824, 1834, 1404, 24, 1804]
GF = {N: 2.0 * N * 4 * 512 * (512 + 1024) / 1e9 for N in (128, 256, 512, 1024)}
for N in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1024, 512, 256)):
    for v in VARIANTS:
        us = C.c_double()
        _lib.check(eng._lib.n2nmn_debug_lstm_bench(eng._ctx, v, 64, 2, N, 200, C.byref(us), eng.stream()))
        print('N=%4d %-28s %7.2f us/launch  %6.1f TFLOP/s fp32-equivalent' %
              (N, names.get(v, str(v)), us.value, GF[N] / us.value * 1e3), flush=True)
