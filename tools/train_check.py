#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error table of the GPU training step vs the autograd oracle
(the same comparison tests/test_gpu_train.py asserts on).  Usage: train_check.py [template|random|ragged]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import n2nmn_oracle_grad as G            # noqa: E402
from n2nmn_amd import synth                          # noqa: E402
from n2nmn_amd.spec import CLEVR_MODULE_NAMES, Dims  # noqa: E402
import test_gpu_train as TT                          # noqa: E402


def main(kind):
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer
    import torch
    d = Dims(T_decoder=10)
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    eng = Engine(d, asm)
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    tr = Trainer(eng, weight_decay=TT.WD)
    if kind == 'template':
        batch = synth.make_inputs(d, seed=0); gt = synth.template_layout_batch(d)
    elif kind == 'random':
        batch = synth.make_inputs(d, seed=41, min_len=1)
        gt = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=1, max_len=3)
    else:
        small = Dims(T_decoder=10, N=37, T_encoder=17)
        batch = synth.make_inputs(small, seed=7, min_len=1); gt = synth.template_layout_batch(small, offset=3)
    t0 = time.time()
    losses, grads, ref_l, ref_g, ex = TT._run(tr, d, w, batch, gt)
    torch.cuda.synchronize()
    print('gpu+oracle %.1fs' % (time.time() - t0))
    print('losses gpu', losses, 'oracle', [ref_l[k] for k in ('avg_sample_loss', 'seq_likelihood_loss', 'l2_reg', 'total_loss')])
    print('scores maxdiff', float(np.abs(TT.t2n(tr.scores) - ex['scores']).max()))
    N = batch['input_seq_batch'].shape[1]
    Td, T, L, E = gt.shape[0], batch['input_seq_batch'].shape[0], d.lstm_dim, d.embed_dim_txt
    inter = {
        'd_word_vecs': TT.t2n(tr.debug_tensor('d_word_vecs', (Td, N, E))),
        'd_token_scores': TT.t2n(tr.debug_tensor('d_token_scores', (Td, N, 16)))[:, :, :d.num_vocab_nmn],
        'd_encoder_outputs': TT.t2n(tr.debug_tensor('d_encoder_outputs', (T, N, L))),
        'd_encoder_h_transformed': TT.t2n(tr.debug_tensor('d_encoder_h_transformed', (T, N, L))),
        'd_scores': TT.t2n(tr.debug_tensor('d_scores', (N, d.num_choices))),
    }
    rows = TT.grad_report(inter, {k: ex[k] for k in inter}) + TT.grad_report(grads, ref_g)
    print(TT.format_report(rows))
    dw = inter['d_word_vecs'] - ex['d_word_vecs']
    err = np.abs(dw).max(axis=2)
    names = list(CLEVR_MODULE_NAMES)
    for t, n in zip(*np.nonzero(err > 1e-6)):
        print('d_word_vecs row t=%d n=%d tok=%s err=%.3e got_max=%.3e want_max=%.3e' % (
            t, n, names[gt[t, n]], err[t, n], np.abs(inter['d_word_vecs'][t, n]).max(),
            np.abs(ex['d_word_vecs'][t, n]).max()))
    print('selection gaps of bad examples', {int(n): float(ex['selection_gap'][n]) for n in sorted(set(np.nonzero(err > 1e-6)[1].tolist()))})
    print('examples with gap < 1e-5:', np.nonzero(ex['selection_gap'] < 1e-5)[0].tolist())
    def cos(a, b):
        return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    bad_n = sorted(set(np.nonzero(err > 1e-6)[1].tolist()))
    for n in bad_n:
        ts = [t for t in range(Td) if err[t, n] > 1e-6]
        for i in ts:
            for j in ts:
                if i < j:
                    print('n=%d cos(err[%d],err[%d])=%.4f  ratio_norm=%.4f' % (
                        n, i, j, cos(dw[i, n], dw[j, n]),
                        np.linalg.norm(dw[i, n]) / np.linalg.norm(dw[j, n])))
            for j in range(Td):
                c1 = cos(dw[i, n], ex['d_word_vecs'][j, n])
                if abs(c1) > 0.5:
                    print('   n=%d err[%d] ~ want[%d] cos %.4f norm ratio %.4f' % (
                        n, i, j, c1, np.linalg.norm(dw[i, n]) / np.linalg.norm(ex['d_word_vecs'][j, n])))
        print('   layout', [names[k] for k in gt[:, n]], 'len', batch['seq_length_batch'][n])
    print('FAILED %d of %d' % (sum(not r[3] for r in rows), len(rows)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'template')
