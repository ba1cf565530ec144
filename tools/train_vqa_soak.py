#!/usr/bin/env python3
"""Soak test of the models_vqa training step (exp_vqa/train_vqa_gt_layout.py): N steps of forward +
backward + Adam with LSTM / question-prior dropout over a fixed set of synthetic batches at the
reference dimensions (the model must memorise them: the loss falls, every weight stays finite, the
zero-padded hidden units and feature channels stay exactly zero)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(steps=200):
    import torch
    from n2nmn_amd import synth, vqa
    d = vqa.VQADims(N=32)
    eng = vqa.VQAEngine(d)
    eng.load_weights(synth.make_weights_from_shapes(vqa.vqa_variable_shapes(d), seed=0))
    tr = vqa.VQATrainer(eng)
    tr.seed = 7
    dev = eng.engine.device
    layouts = (['_Find', '_Describe'], ['_Find', '_Find', '_And', '_Describe'],
               ['_Find', '_Transform', '_Describe'], ['_Find', '_Transform', '_Find', '_And', '_Describe'])
    rng = np.random.default_rng(0)
    batches, gts = [], []
    for i in range(2):
        lens = rng.integers(3, d.T_encoder + 1, size=d.N).astype(np.int32)
        seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, d.N)).astype(np.int32)
        seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
        feat = torch.relu(torch.randn((d.N, d.H, d.W, d.D), generator=torch.Generator().manual_seed(i)))
        batches.append(dict(input_seq_batch=torch.as_tensor(seq).to(dev),
                            seq_length_batch=torch.as_tensor(lens).to(dev), image_feat_batch=feat.to(dev),
                            answer_label_batch=torch.as_tensor(
                                rng.integers(0, d.num_choices, size=d.N).astype(np.int32)).to(dev)))
        gts.append(np.array([eng.assembler.module_list2tokens(layouts[(n + i) % 4], d.T_decoder)
                             for n in range(d.N)], np.int32).T.copy())
    hist = []
    for it in range(steps):
        losses = tr.step(batches[it % 2], gts[it % 2])
        if it % 20 == 0 or it == steps - 1:
            l = losses.cpu().numpy()
            acc = float((tr.scores.argmax(1).cpu() == batches[it % 2]['answer_label_batch'].cpu()).float().mean())
            hist.append((it, float(l[0]), float(l[1]), acc))
            print('iter %4d  avg_sample_loss %.4f  seq_likelihood_loss %.4f  batch accuracy (dropout on) %.3f'
                  % (it, l[0], l[1], acc), flush=True)
    w = tr.get_weights()
    bad = [k for k, v in w.items() if not torch.isfinite(v).all()]
    assert not bad, bad
    Lr, Lp = d.lstm_dim, eng.idims.lstm_dim
    b = w['neural_module_network/layout_generation/encoder_decoder/encoder/lstm/multi_rnn_cell/cell_1/'
          'basic_lstm_cell/biases']
    assert float(b.view(4, Lp)[:, Lr:].abs().max()) == 0.0, 'padded hidden units moved'
    first, last = hist[0], hist[-1]
    assert last[1] + last[2] < 0.5 * (first[1] + first[2]), 'loss did not fall'
    print('ok: loss %.3f -> %.3f over %d steps, all weights finite, padding still zero'
          % (first[1] + first[2], last[1] + last[2], steps))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
