#!/usr/bin/env python3
"""BASELINE-size reference-code fixture: the reference's own model files
(models_clevr/nmn3_model.py, nmn3_netgen_att.py, nmn3_modules.py, nmn3_assembler.py, util/cnn.py,
util/empty_safe_conv.py -- unmodified, imported from /root/reference) run under the eager TF1 / Fold
stand-in (oracle/tf1_stub) at the eval configuration of exp_clevr/eval_clevr.py:27-37 -- N = 64,
T_encoder = 45, T_decoder = 20 -- once free-running (greedy decoder, BASELINE configs[2]) and once on
the ten-template ground-truth layouts (configs[1]).  Only logits, tokens and validity are stored
(tests/golden/float_golden_full.npz); weights and inputs are seeded (float_cases.py).

    python tests/golden/make_float_golden_full.py [--check]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_float_golden as G          # noqa: E402  (sets up the stand-in and the reference imports)
import float_cases as FC               # noqa: E402
from n2nmn_amd import synth            # noqa: E402

OUT = os.path.join(HERE, 'float_golden_full.npz')


def generate():
    out = {}
    d, batch, asm, model, exprs, validity, scores = G.build_clevr('full')
    out['greedy/predicted_tokens'] = G.n(model.predicted_tokens).astype(np.int32)
    out['greedy/scores'] = G.n(scores)
    out['greedy/validity'] = validity
    out['greedy/token_probs'] = G.n(model.token_probs)
    print('greedy done: %d valid layouts of %d' % (int(validity.sum()), d.N), flush=True)
    gt = synth.template_layout_batch(d)
    d, batch, asm, model, exprs, validity, scores = G.build_clevr('full', use_gt=True, gt=gt)
    assert validity.all() and np.array_equal(G.n(model.predicted_tokens), gt)
    out['gt/scores'] = G.n(scores)
    out['gt/log_seq_prob'] = G.n(model.log_seq_prob)
    print('gt done', flush=True)
    return out


def main():
    out = generate()
    if '--check' in sys.argv:
        old = np.load(OUT)
        worst = max(float(np.max(np.abs(np.asarray(old[k], np.float64) - np.asarray(out[k], np.float64))))
                    for k in out)
        print('committed fixture vs regenerated: max |diff| = %.3e' % worst)
        assert sorted(old.files) == sorted(out) and worst <= 1e-12
        return
    np.savez_compressed(OUT, **out)
    print('wrote %s (%.1f KB)' % (OUT, os.path.getsize(OUT) / 1024))


if __name__ == '__main__':
    main()
