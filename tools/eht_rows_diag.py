#!/usr/bin/env python3
"""Decoder attention / word vectors of one 192-row pass with and without the row-listed
encoder_h_transform GEMM (n2nmn_debug_set "eht_rows"; `--full` = the GEMM over all T N rows): run twice, the second
run compares with the first's dump."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from n2nmn_amd import synth
from n2nmn_amd.engine import Engine
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.nmn3_assembler import Assembler

d = Dims(N=192)
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
eng.load_weights(synth.make_weights(d, seed=0))
eng.set_mode('throughput')
if '--full' in sys.argv:
    eng.debug_set('eht_rows', '0')
b = synth.make_inputs(d, seed=70)
g = np.concatenate([synth.template_layout_batch(Dims(), offset=1)] * 3, axis=1)
out = eng.seq2seq(b['input_seq_batch'], b['seq_length_batch'], use_gt_layout=True, gt_layout=g)
torch.cuda.synchronize()
cur = {k: out[k].cpu().numpy() for k in ('atts', 'word_vecs', 'token_probs')}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'eht_diag.npz')
if os.path.exists(path) and '--compare' in sys.argv:
    ref = np.load(path)
    lens = b['seq_length_batch']
    for k in cur:
        print(k, 'max |diff|', float(np.abs(cur[k] - ref[k]).max()))
    da = np.abs(cur['atts'] - ref['atts'])            # [Td, T, N]
    bad = np.argwhere(da > 0)
    print('atts entries that differ', len(bad), 'of', da.size)
    if len(bad):
        qs = sorted(set(int(x[2]) for x in bad))
        print('questions', qs[:20], 'lens', [int(lens[q]) for q in qs[:20]])
        print('first', bad[:5].tolist())
else:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, **cur)
    print('saved')
