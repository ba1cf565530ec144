#!/usr/bin/env python3
"""Soak test of the training step: N steps of forward + backward + Adam over a fixed set of synthetic
batches (the model must memorise them: total_loss falls, no NaN / Inf anywhere in the weights)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(steps=600):
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.train import Trainer
    d = Dims(T_decoder=10)
    eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
    eng.load_weights(synth.make_weights(d, seed=0))
    tr = Trainer(eng)
    batches = [{k: torch.as_tensor(v).to(eng.device) for k, v in synth.make_inputs(d, seed=i).items()}
               for i in range(4)]
    gts = [synth.template_layout_batch(d, offset=i) for i in range(4)]
    hist = []
    for it in range(steps):
        losses = tr.step(batches[it % 4], gts[it % 4])
        if it % 50 == 0 or it == steps - 1:
            l = losses.cpu().numpy()
            acc = float((tr.scores.argmax(1).cpu() == batches[it % 4]['answer_label_batch'].cpu()).float().mean())
            hist.append((it, l.tolist(), acc))
            print('iter %4d  avg_sample_loss %.4f  seq_likelihood_loss %.4f  l2_reg %.1f  total %.4f  '
                  'batch accuracy %.3f' % (it, l[0], l[1], l[2], l[3], acc), flush=True)
    w = tr.get_weights()
    bad = [k for k, v in w.items() if not torch.isfinite(v).all()]
    assert not bad, bad
    assert hist[-1][1][3] < 0.25 * hist[0][1][3], 'loss did not fall'
    print('ok: total_loss %.3f -> %.3f over %d steps, all weights finite' % (hist[0][1][3], hist[-1][1][3], steps))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 600)
