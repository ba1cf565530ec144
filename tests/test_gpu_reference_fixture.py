"""HIP path vs numbers computed by the REFERENCE'S OWN CODE (tests/golden/float_golden.npz, produced
by tests/golden/make_float_golden.py from the unmodified models_clevr / models_vqa files and the loss
blocks of the two CLEVR training scripts, float64).  No oracle in the loop: the fixture is the truth.

Bar (north star): every fp32 forward output within 1e-4 absolute; tokens identical (the fixture's
tokens are forced into the decoder as SURVEY 8c prescribes, and the free-running tokens are also
compared); gradient probes within 2e-4 * max|g| of the reference-code autograd value.
"""
import json
import os
import sys

import numpy as np
import pytest

from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from util import assert_close, t2n

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import float_cases as FC  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4
GRAD_RTOL = 2e-4
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'float_golden.npz')


@pytest.fixture(scope='module')
def fx():
    z = np.load(GOLDEN)
    return z, json.loads(bytes(z['meta_json']).decode())


def _seq2seq_close(z, key, s2s, T_enc):
    for name in ('token_probs', 'neg_entropy', 'word_vecs', 'log_seq_prob'):
        assert_close(key + '/' + name, t2n(s2s[name]), z[key + '/' + name], TOL)
    assert_close(key + '/atts', t2n(s2s['atts']), z[key + '/atts'][..., 0], TOL)


def test_greedy_forward_matches_reference_code(clevr_engine, fx):
    eng, d0, asm, w = clevr_engine
    z, meta = fx
    d, batch = FC.clevr_inputs('greedy')
    want_tok = z['greedy/predicted_tokens']
    free = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec=d.T_decoder,
                       reuse_buffers=False)
    assert np.array_equal(t2n(free['predicted_tokens']), want_tok)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec=d.T_decoder,
                      forced_tokens=want_tok, debug=True)
    assert np.array_equal(t2n(s2s['predicted_tokens']), want_tok)
    _seq2seq_close(z, 'greedy', s2s, d.T_encoder)
    assert_close('encoder_outputs', t2n(s2s['encoder_outputs']), z['greedy/encoder_outputs'], TOL)
    assert_close('encoder_h_transformed', t2n(s2s['encoder_h_transformed']),
                 z['greedy/encoder_h_transformed'], TOL)
    es = t2n(s2s['encoder_states'])                      # [layer][c|h][N][L]
    for l in range(2):
        assert_close('c%d' % l, es[l, 0], z['greedy/encoder_state_c%d' % l], TOL)
        assert_close('h%d' % l, es[l, 1], z['greedy/encoder_state_h%d' % l], TOL)
    packed, validity = asm.assemble_packed(want_tok)
    assert np.array_equal(validity, z['greedy/validity'])
    scores = eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs'])
    assert_close('scores', t2n(scores), z['greedy/scores'], TOL)
    # the whole two-phase path in one call
    scores2, tokens2, validity2 = eng.forward(batch, T_dec=d.T_decoder)
    assert np.array_equal(tokens2, want_tok)
    assert_close('scores (forward)', t2n(scores2), z['greedy/scores'], TOL)


def test_sampled_decoding_matches_reference_code(clevr_engine, fx):
    eng, d0, asm, w = clevr_engine
    z, meta = fx
    d, batch = FC.clevr_inputs('sampled')
    u = FC.sample_uniforms(d).astype(np.float32)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec=d.T_decoder,
                      sample_uniforms=u)
    want_tok = z['sampled/predicted_tokens']
    assert np.array_equal(t2n(s2s['predicted_tokens']), want_tok)
    _seq2seq_close(z, 'sampled', s2s, d.T_encoder)
    packed, validity = asm.assemble_packed(want_tok)
    scores = eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs'])
    assert_close('scores', t2n(scores), z['sampled/scores'], TOL)


def test_every_operator_matches_reference_code(clevr_engine, fx):
    """Modules.<X>Module called directly (exp_shapes/visualize_shapes.ipynb pattern), Nb = 3."""
    from n2nmn_amd.nmn3_modules import Modules
    eng, d0, asm, w = clevr_engine
    z, meta = fx
    d, x = FC.module_inputs()
    mods = Modules(x['image_feat'], x['word_vecs'], d.num_choices, engine=eng)
    for name, nin in FC.MODULE_CALLS:
        args = [x['input_0'], x['input_1']][:nin]
        got = getattr(mods, name)(*args, x['time_idx'], x['batch_idx'])
        assert_close(name, t2n(got), z['modules/' + name], TOL)


def _probe_check(z, key, meta, got, rtol):
    bad = []
    for name, m in meta.items():
        g = np.asarray(got[name], np.float64).reshape(-1)
        d = np.max(np.abs(g[FC.probe_indices(name, g.size)] - z[key + '/' + name]))
        tol = rtol * m['absmax'] + 1e-7
        nd = abs(np.sqrt(np.sum(g * g)) - m['norm'])
        if not (d <= tol and nd <= 5 * rtol * m['norm'] + 1e-7 and np.isfinite(g).all()):
            bad.append('%s: probe diff %.3e (tol %.3e), norm %.6e vs %.6e' % (name, d, tol,
                       np.sqrt(np.sum(g * g)), m['norm']))
    assert not bad, '\n'.join(bad)


@pytest.fixture(scope='module')
def trainer():
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer
    d = Dims(T_decoder=10)
    eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
    w = synth.make_weights(d, seed=FC.WEIGHT_SEED)
    eng.load_weights(w)
    return Trainer(eng, weight_decay=5e-6), eng, w


def test_training_step_matches_reference_code(trainer, fx):
    """exp_clevr/train_clevr_gt_layout.py loss block: losses, logits, every variable's gradient,
    and the weights after one clipped Adam step."""
    tr, eng, w = trainer
    z, meta = fx
    m = meta['gt']
    d, batch = FC.clevr_inputs('gt')
    gt = FC.gt_layouts(d)
    eng.load_weights(w)
    from n2nmn_amd.train import Trainer
    tr = Trainer(eng, weight_decay=m['weight_decay'], max_grad_l2_norm=m['max_grad_l2_norm'])
    scale = tr.forward_backward(batch, gt, reduce=False)
    assert_close('scores', t2n(tr.scores), z['gt/scores'], TOL)
    losses = t2n(tr.losses)
    for i, k in enumerate(('avg_sample_loss', 'seq_likelihood_loss', 'l2_reg', 'total_loss')):
        assert abs(losses[i] - m[k]) <= 1e-4 * max(1.0, abs(m[k])), (k, losses[i], m[k])
    _probe_check(z, 'gt/grad', m['grad'], {k: t2n(v) for k, v in tr.gradients().items()}, GRAD_RTOL)
    tr.apply(scale)
    w1 = {k: t2n(v) for k, v in tr.get_weights().items()}
    bad = []
    for name, mm in m['adam_w1'].items():
        g = w1[name].astype(np.float64).reshape(-1)
        want = z['gt/adam_w1/' + name]
        # first Adam step moves every element by ~lr = 1e-3 * sign(g): compare the UPDATE
        w0 = np.asarray(w[name], np.float64).reshape(-1)[FC.probe_indices(name, g.size)]
        d_got, d_want = g[FC.probe_indices(name, g.size)] - w0, want - w0
        # elements whose gradient is ~0 have an ill-conditioned first step (g / (|g| + eps))
        if np.max(np.abs(d_got - d_want)) > 2e-5:
            bad.append('%s: update diff %.3e' % (name, np.max(np.abs(d_got - d_want))))
    assert len(bad) <= 2, '\n'.join(bad)
    eng.load_weights(w)


def test_policy_gradient_step_matches_reference_code(trainer, fx):
    """exp_clevr/train_clevr_rl_gt_layout.py loss block on the layouts the decoder sampled."""
    tr, eng, w = trainer
    z, meta = fx
    m = meta['sampled']
    d, batch = FC.clevr_inputs('sampled')
    u = FC.sample_uniforms(d).astype(np.float32)
    eng.load_weights(w)
    from n2nmn_amd.train import Trainer
    tr = Trainer(eng, weight_decay=m['weight_decay'])
    tr.rl.update(invalid_expr_loss=m['invalid_expr_loss'], lambda_entropy=m['lambda_entropy'],
                 baseline_decay=m['baseline_decay'])
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec=d.T_decoder,
                      sample_uniforms=u)
    tokens = t2n(s2s['predicted_tokens'])
    assert np.array_equal(tokens, z['sampled/predicted_tokens'])
    tr.forward_backward(batch, tokens, reduce=False, objective=1)
    losses = t2n(tr.losses)
    want = dict(avg_sample_loss=losses[0], policy_gradient_loss=losses[1], l2_reg=losses[2],
                total_loss=losses[3], entropy_reg=losses[4])
    for k, v in want.items():
        assert abs(v - m[k]) <= 1e-4 * max(1.0, abs(m[k])), (k, v, m[k])
    assert abs(float(t2n(tr.baseline)[0]) - m['baseline_after']) < 1e-5
    _probe_check(z, 'sampled/grad', m['grad'], {k: t2n(v) for k, v in tr.gradients().items()},
                 GRAD_RTOL)


@pytest.mark.parametrize('mode', ['greedy', 'gt'])
def test_vqa_model_matches_reference_code(fx, mode):
    """models_vqa NMN3Model (coordinate channels, question prior net) at the reference dimensions."""
    from n2nmn_amd.vqa import VQAEngine
    z, meta = fx
    d, batch, gt = FC.vqa_setup()
    eng = VQAEngine(d)
    eng.load_weights(FC.vqa_weights(d))
    key = 'vqa_' + mode
    want_tok = z[key + '/predicted_tokens']
    if mode == 'gt':
        scores, tokens, validity = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
    else:
        scores, tokens, validity = eng.forward(batch)
    assert np.array_equal(tokens, want_tok)
    assert np.array_equal(validity, z[key + '/validity'])
    assert_close(key + '/scores', t2n(scores), z[key + '/scores'], TOL)


# ---- BASELINE size (exp_clevr/eval_clevr.py:27-37): N = 64, T_encoder = 45, T_decoder = 20 --------------
GOLDEN_FULL = os.path.join(os.path.dirname(GOLDEN), 'float_golden_full.npz')


def test_full_size_batch_matches_reference_code(clevr_engine):
    """configs[1] and configs[2] at the size the metric is quoted on, against logits computed by the
    reference's own model files (make_float_golden_full.py): single batch of 64, default mode."""
    eng, d0, asm, w = clevr_engine
    z = np.load(GOLDEN_FULL)
    d, batch = FC.clevr_inputs('full')
    gt = synth.template_layout_batch(d)
    scores, tokens, validity = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
    assert np.array_equal(tokens, gt) and validity.all()
    assert_close('gt scores', t2n(scores), z['gt/scores'], TOL)
    # free-running decoder: the reference code's tokens forced in, then the whole path
    want_tok = z['greedy/predicted_tokens']
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=want_tok)
    assert_close('token_probs', t2n(s2s['token_probs']), z['greedy/token_probs'], TOL)
    packed, val = asm.assemble_packed(want_tok)
    assert np.array_equal(val, z['greedy/validity'])
    sc = eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs'])
    assert_close('greedy scores (reference tokens)', t2n(sc), z['greedy/scores'], TOL)
    sc2, tok2, val2 = eng.forward(batch)
    same = (tok2 == want_tok).all(axis=0)
    assert same.mean() >= 0.9 and val2.all()
    assert_close('greedy scores (free-running)', t2n(sc2)[same], z['greedy/scores'][same], TOL)


def test_full_size_batch_in_a_throughput_pass_matches_reference_code():
    """the same 64 questions as one slot of an 8-slot pass in 'throughput' mode (lstm_tile_kernel,
    per-question decoder attention, chip-wide walker front end, deferred pooling): the path bench.py
    times, against the reference code's logits"""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    z = np.load(GOLDEN_FULL)
    d, batch = FC.clevr_inputs('full')
    gt = synth.template_layout_batch(d)
    sb = SuperBucket(d, Assembler(list(CLEVR_MODULE_NAMES)), K=8)
    sb.load_weights(FC.clevr_weights())
    sb.engine.set_mode('throughput')
    for k in range(8):
        other = synth.make_inputs(d, seed=300 + k, min_len=1)
        sb.fill(k, batch if k == 5 else other, gt if k == 5 else synth.template_layout_batch(d, offset=k))
    sb.run(use_gt_layout=True)
    assert_close('gt scores, slot 5 of 8', t2n(sb.result(5)[0]), z['gt/scores'], TOL)
    sb.run(use_gt_layout=False)
    sc, tok, val = [t2n(x) for x in sb.result(5)]
    same = (tok == z['greedy/predicted_tokens']).all(axis=0)
    assert same.mean() >= 0.9 and val.all()
    assert_close('greedy scores, slot 5 of 8', sc[same], z['greedy/scores'][same], TOL)
