#!/usr/bin/env python3
"""Soak test of the policy-gradient objective (exp_clevr/train_clevr_rl_gt_layout.py flow): clone the
ground-truth layouts for a while (train_clevr_gt_layout.py), then fine-tune with REINFORCE on
layouts the decoder samples itself.  Over a small fixed set of synthetic batches the policy must
keep producing valid layouts, the expected loss of its samples must stay near the cloning level or
fall, the policy entropy must fall, and every weight must stay finite."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(clone_steps=300, rl_steps=300):
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
    for it in range(clone_steps):
        losses = tr.step(batches[it % 4], gts[it % 4])
    l = losses.cpu().numpy()
    print('after %d cloning steps: avg_sample_loss %.4f  seq_likelihood_loss %.4f' % (clone_steps, l[0], l[1]))
    rng = np.random.default_rng(0)
    tr.baseline.fill_(0.5)
    hist = []
    for it in range(rl_steps):
        u = rng.random((d.T_decoder, d.N)).astype(np.float32)
        losses, tokens, validity = tr.step_rl(batches[it % 4], u, lr=1e-4)
        if it % 25 == 0 or it == rl_steps - 1:
            l = losses.cpu().numpy()
            same = float((tokens == gts[it % 4]).all(axis=0).mean())
            hist.append((it, l.tolist(), float(validity.mean()), same))
            print('rl iter %4d  avg_sample_loss %.4f  policy_gradient_loss %+.4f  entropy_reg %+.4f  '
                  'baseline %.4f  valid %.3f  layout == gt %.3f' %
                  (it, l[0], l[1], l[4], float(tr.baseline.cpu()[0]), validity.mean(), same), flush=True)
    w = tr.get_weights()
    bad = [k for k, v in w.items() if not torch.isfinite(v).all()]
    assert not bad, bad
    assert all(h[2] == 1.0 for h in hist), 'the automaton let an invalid layout through'
    assert hist[-1][1][4] >= hist[0][1][4] - 1e-3 or hist[-1][1][0] <= hist[0][1][0], 'no progress'
    print('ok: %d RL steps, avg_sample_loss %.3f -> %.3f, entropy_reg %.3f -> %.3f, all weights finite'
          % (rl_steps, hist[0][1][0], hist[-1][1][0], hist[0][1][4], hist[-1][1][4]))


if __name__ == '__main__':
    main(*(int(a) for a in sys.argv[1:3]))
