#!/usr/bin/env python3
"""A/B of the recurrent-step modes over whole super-bucketed passes, one stream (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from n2nmn_amd import synth
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.superbucket import SuperBucket
d = Dims(); asm = Assembler(list(CLEVR_MODULE_NAMES))
for K in (4, 8, 16):
    sb = SuperBucket(d, asm, K)
    sb.load_weights(synth.make_weights(d, seed=0))
    for k in range(K):
        sb.fill(k, synth.make_inputs(d, seed=k), synth.template_layout_batch(d, offset=k))
    for mode in ('latency', 'throughput_ksplit', 'throughput'):
        sb.engine.set_mode(mode)
        for _ in range(5): sb.run(use_gt_layout=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): sb.run(use_gt_layout=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print('K=%d mode=%-17s %.3f ms per pass  %.0f q/s' % (K, mode, dt * 1e3, K * 64 / dt), flush=True)
