#!/usr/bin/env python3
"""Randomised cross-check: super-bucketed passes (random K, ragged question lengths incl. length 1,
random teacher-forced layouts, greedy decoding) against one-batch-at-a-time passes of a separate
engine.  Logits must agree to 2e-5, tokens / validity exactly."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(trials=12):
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.superbucket import SuperBucket
    rng = np.random.default_rng(2026)
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    worst = 0.0
    for trial in range(trials):
        Nb = int(rng.choice([8, 16, 40, 64]))
        K = int(rng.integers(1, 9))
        d = Dims(N=Nb)
        w = synth.make_weights(d, seed=trial)
        one = Engine(d, asm)
        one.load_weights(w)
        sb = SuperBucket(d, asm, K)
        sb.load_weights(w)
        use_gt = bool(rng.integers(0, 2))
        batches, gts = [], []
        for k in range(K):
            b = synth.make_inputs(d, seed=1000 * trial + k, min_len=1)
            lens = b['seq_length_batch'].copy()
            lens[rng.integers(0, Nb, size=max(1, Nb // 8))] = 1          # very short questions
            seq = b['input_seq_batch'].copy()
            seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
            b = dict(b, seq_length_batch=lens, input_seq_batch=seq)
            batches.append(b)
            gts.append(synth.template_layout_batch(d, offset=int(rng.integers(0, 10))))
            sb.fill(k, b, gts[-1] if use_gt else None)
        sb.run(use_gt_layout=use_gt)
        for k in range(K):
            s1, t1, v1 = one.forward(batches[k], use_gt_layout=use_gt, gt_layout=gts[k] if use_gt else None)
            s2, t2, v2 = sb.result(k)
            s1 = torch.as_tensor(s1).cpu().numpy()
            err = float(np.abs(s1 - s2.cpu().numpy()).max())
            worst = max(worst, err)
            assert err <= 2e-5, (trial, k, err)
            assert np.array_equal(np.asarray(t1), t2.cpu().numpy()), (trial, k, 'tokens')
            assert np.array_equal(np.asarray(v1).astype(bool), v2.cpu().numpy().astype(bool)), (trial, k)
        print('trial %2d: batch %2d x K=%d, gt=%s ok' % (trial, Nb, K, use_gt), flush=True)
    print('ok: %d trials, worst |logit difference| %.2e' % (trials, worst))


if __name__ == '__main__':
    main()
