"""models_vqa training step (exp_vqa/train_vqa_gt_layout.py; train_vqa2_gt_layout.py differs only in
max_iter and the data file) on the HIP path against numbers computed
by the REFERENCE'S OWN CODE: tests/golden/float_golden.npz `vqa_train/*` = the unmodified
models_vqa/*.py with encoder / decoder / question-prior dropout under the TF1 stand-in, the loss
block of the training script, autograd gradients of every variable and one Adam step (float64).
The dropout masks are inputs on both sides (tests/golden/float_cases.py::vqa_dropout_masks).

Bar: logits / log_seq_prob 1e-4 absolute, losses 1e-4 relative, gradient probes 2e-4 * max|g|,
weights after the Adam step 2e-3 of a step (lr) except where the gradient is round-off small."""
import json
import os
import sys

import numpy as np
import pytest

from util import assert_close, t2n

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import float_cases as FC  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'float_golden.npz')
GRAD_RTOL = 2e-4


@pytest.fixture(scope='module')
def fx():
    z = np.load(GOLDEN)
    return z, json.loads(bytes(z['meta_json']).decode())


@pytest.fixture(scope='module')
def setup():
    from n2nmn_amd.vqa import VQAEngine, VQATrainer
    d, batch, gt = FC.vqa_setup()
    batch = dict(batch, answer_label_batch=FC.vqa_labels(d))
    eng = VQAEngine(d)
    w = FC.vqa_weights(d)
    eng.load_weights(w)
    return d, batch, gt, eng, VQATrainer(eng), w


def _probe_check(z, key, meta, got, rtol):
    bad = []
    for name, m in meta.items():
        g = np.asarray(got[name], np.float64).reshape(-1)
        d = np.max(np.abs(g[FC.probe_indices(name, g.size)] - z[key + '/' + name]))
        tol = rtol * m['absmax'] + 1e-7
        nd = abs(np.sqrt(np.sum(g * g)) - m['norm'])
        if not (d <= tol and nd <= 5 * rtol * m['norm'] + 1e-7 and np.isfinite(g).all()):
            bad.append('%s: probe diff %.3e (tol %.3e), norm %.6e vs %.6e' % (
                name, d, tol, np.sqrt(np.sum(g * g)), m['norm']))
    assert not bad, '\n'.join(bad)


def test_vqa_training_step_with_dropout_matches_reference_code(setup, fx):
    d, batch, gt, eng, tr, w = setup
    z, meta = fx
    m = meta['vqa_train']
    eng.load_weights(w)
    tr.masks = FC.vqa_dropout_masks(d)
    tr.forward_backward(batch, gt, reduce=False)
    assert_close('scores', t2n(tr.scores), z['vqa_train/scores'], 1e-4)
    ls = t2n(tr.losses)
    for i, k in enumerate(('avg_sample_loss', 'seq_likelihood_loss')):
        assert abs(ls[i] - m[k]) <= 1e-4 * max(1.0, abs(m[k])), (k, ls[i], m[k])
    assert abs(ls[3] - m['total_loss']) <= 1e-4 * abs(m['total_loss'])
    grads = tr.gradients_reference_shaped()
    assert sorted(grads) == m['variables']
    _probe_check(z, 'vqa_train/grad', m['grad'], grads, GRAD_RTOL)
    # padded hidden units / feature channels receive exactly zero gradient
    flat = t2n(tr.grads)
    assert np.isfinite(flat).all()
    tr.iteration = 0
    tr.apply(1.0)
    w1 = tr.weights_reference_shaped()
    lr = tr.hyper['lr']
    for name, mm in m['adam_w1'].items():
        got = np.asarray(w1[name], np.float64).reshape(-1)[FC.probe_indices(name, w1[name].size)]
        want = z['vqa_train/adam_w1/' + name]
        # the first Adam step moves every weight by lr * g / (|g| + eps): compare at that scale,
        # except where |g| is so small that fp32 round-off decides the sign
        g = np.asarray(grads[name], np.float64).reshape(-1)[FC.probe_indices(name, w1[name].size)]
        ok = np.abs(g) > 1e-3 * m['grad'][name]['absmax']
        assert np.all(np.abs(got - want)[ok] <= 2e-2 * lr + 1e-6), name
        assert np.all(np.abs(got - want) <= 2.001 * lr + 1e-6), name


def test_vqa_training_without_dropout_equals_plain_forward(setup, fx):
    """dropout switched off: the training forward's logits are the eval forward's (fixture vqa_gt)."""
    from n2nmn_amd.vqa import VQATrainer
    d, batch, gt, eng, tr, w = setup
    z, meta = fx
    eng.load_weights(w)
    saved = dict(tr.dropout)
    tr.dropout = {k: False for k in saved}
    try:
        tr.forward_backward(batch, gt, reduce=False)
    finally:
        tr.dropout = saved
    assert_close('scores', t2n(tr.scores), z['vqa_gt/scores'], 1e-4)
    assert np.isfinite(t2n(tr.grads)).all()


def test_drawn_masks_keep_about_half(setup):
    d, batch, gt, eng, tr, w = setup
    eng.load_weights(w)
    tr.masks = None
    tr.forward_backward(batch, gt, reduce=False)
    m = tr._mult['enc0'][..., :d.lstm_dim]
    frac = float((m > 0).float().mean())
    assert 0.45 < frac < 0.55 and float(m.max()) == 2.0
    assert np.isfinite(t2n(tr.losses)[:2]).all()


def _token_validity(asm, tokens, T_dec):
    """[T_dec, N, V] masks of the automaton along the given token sequences
    (nmn3_netgen_att.py:8-15,200-203)."""
    from oracle import n2nmn_oracle as O
    Td, N = tokens.shape
    X = np.tile(np.array([[0, 0, T_dec]], np.int64), (N, 1))
    out = np.zeros((Td, N, asm.P.shape[0]), bool)
    for t in range(Td):
        out[t] = O.valid_tokens(X, asm.W, asm.b)
        X = X + asm.P[tokens[t]]
    return out


def test_vqa_policy_gradient_step_with_dropout_matches_oracle(setup):
    """exp_vqa/train_vqa_rl_gt_layout.py:106-126: the layout is sampled from the network under this
    step's dropout masks, then REINFORCE + entropy + answer loss through the same masks.  Oracle:
    loss_and_grads_rl(vqa_masks=...) on the tokens the GPU sampled (its CLEVR form is pinned to the
    reference's RL loss block, its models_vqa + dropout forward to the vqa_train fixture)."""
    from oracle import n2nmn_oracle as O
    from oracle import n2nmn_oracle_grad as G
    d, batch, gt, eng, tr, w = setup
    eng.load_weights(w)
    masks = FC.vqa_dropout_masks(d)
    tr.masks = masks
    tr.baseline.fill_(1.25)
    uni = np.random.default_rng(3).random((d.T_decoder, d.N)).astype(np.float32)
    losses, tokens, validity = tr.step_rl(batch, uni, update=False)
    losses = t2n(losses)
    assert validity.all()
    asm = eng.assembler
    tv = _token_validity(asm, tokens, d.T_decoder)
    assert tv[np.arange(d.T_decoder)[:, None], np.arange(d.N)[None], tokens].all()
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    ref_l, ref_g, ex = G.loss_and_grads_rl(w64, list(O.VQA_MODULE_NAMES), batch, d.T_decoder,
                                           d.num_choices, tokens, tv, 1.25, weight_decay=0.0,
                                           vqa_masks=masks)
    assert_close('scores', t2n(tr.scores), ex['scores'], 1e-4)
    for i, k in ((0, 'avg_sample_loss'), (1, 'policy_gradient_loss'), (3, 'total_loss'),
                 (4, 'entropy_reg')):
        assert abs(losses[i] - ref_l[k]) <= 1e-4 * max(1.0, abs(ref_l[k])), (k, losses[i], ref_l[k])
    assert abs(float(t2n(tr.baseline)[0]) - ref_l['new_baseline']) <= 1e-5
    grads = tr.gradients_reference_shaped()
    bad = []
    for k, want in ref_g.items():
        tol = GRAD_RTOL * max(float(np.abs(want).max()), 1e-30) + 1e-7
        err = float(np.abs(np.asarray(grads[k], np.float64) - want).max())
        if not err <= tol:
            bad.append('%s: %.3e > %.3e' % (k, err, tol))
    assert not bad, '\n'.join(bad)
