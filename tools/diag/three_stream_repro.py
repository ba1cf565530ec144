#!/usr/bin/env python3
"""Repro attempt for the intermittent 3-worker bf16x3 mismatch (profiles/r05_notes.md section 8): three workers
run the same-width pass concurrently; each worker's logits are compared with the same pass run alone.
Usage: python tools/diag/three_stream_repro.py [mode] [iterations] [streams]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get('N2NMN_DIAG_LIB'):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import build_diag
    print('library:', build_diag.use_diag_lib(os.environ['N2NMN_DIAG_LIB']), flush=True)
from n2nmn_amd import synth
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.pipeline import PassPipeline
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

mode = sys.argv[1] if len(sys.argv) > 1 else 'throughput_bf16x3'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = Dims()
p = PassPipeline(d, Assembler(list(CLEVR_MODULE_NAMES)), synth.make_weights(d, seed=0), streams=S, kcap=16,
                 mode=None if mode == 'throughput' else mode)
p.fill_all(lambda i: synth.make_inputs(d, seed=500 + i, min_len=1), lambda i: synth.template_layout_batch(d, offset=i))
torch.cuda.synchronize()
widths = [[8, 10], [16, 8], [12, 16]][:S]
p.run(widths, gt=True)
p.run(widths, gt=False)


def alone(si, n):
    for wk in p.workers:
        wk['next'] = 0
    p.run([[n] if k == si else [] for k in range(S)], gt=True)
    return p.bucket(si, 0).scores.cpu().numpy().copy()


ref = [alone(si, 10) for si in range(S)]
again = [alone(si, 10) for si in range(S)]
print('alone vs alone:', [float(np.abs(a - b).max()) for a, b in zip(ref, again)], flush=True)
bad = 0
for it in range(iters):
    if it % 4 == 3:
        p.run(widths, gt=False)              # (a greedy pass in between, as in the test file)
    for wk in p.workers:
        wk['next'] = 0
    p.run([[10]] * S, gt=True)
    diffs = [float(np.abs(p.bucket(si, 0).scores.cpu().numpy() - ref[si]).max()) for si in range(S)]
    if max(diffs) > 1e-5:
        bad += 1
        print('iteration %d: max |concurrent - alone| per worker %s' % (it, ['%.2e' % x for x in diffs]), flush=True)
print('%s, %d streams: %d of %d concurrent rounds differ from the passes run alone by more than 1e-5' % (mode, S, bad, iters))
p.close()
