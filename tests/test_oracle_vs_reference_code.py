"""PIN of the float path: the CPU oracle against numbers computed by the REFERENCE'S OWN CODE.

tests/golden/float_golden.npz was produced by running the unmodified models_clevr / models_vqa /
util files of ronghanghu/n2nmn (and the loss blocks of its two CLEVR training scripts) under the eager
TF1 / Fold stand-in of oracle/tf1_stub, in float64 (tests/golden/make_float_golden.py).  Here the
oracle (oracle/n2nmn_oracle.py numpy, oracle/n2nmn_oracle_grad.py autograd) must reproduce every
recorded number on the same seeded inputs.  Tolerance 1e-10 absolute on forward values of O(1)
(both sides are float64; summation order differs) and 1e-9 relative to max|g| on gradients.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from oracle import n2nmn_oracle_grad as G
from n2nmn_amd.spec import Dims, variable_shapes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import float_cases as FC  # noqa: E402

TOL = 1e-10
GRAD_RTOL = 1e-9
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'float_golden.npz')


@pytest.fixture(scope='module')
def fx():
    z = np.load(GOLDEN)
    meta = json.loads(bytes(z['meta_json']).decode())
    return z, meta


@pytest.fixture(scope='module')
def w64():
    return {k: v.astype(np.float64) for k, v in FC.clevr_weights().items()}


def close(name, got, want, tol=TOL):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    d = float(np.max(np.abs(got - want))) if got.size else 0.0
    assert d <= tol, '%s: max|diff| %.3e > %.1e' % (name, d, tol)


def check_seq2seq(z, key, r):
    dec, enc = r['dec'], r['enc']
    assert np.array_equal(dec['predicted_tokens'], z[key + '/predicted_tokens'])
    for name in ('token_probs', 'neg_entropy', 'word_vecs', 'atts'):
        close(key + '/' + name, dec[name], z[key + '/' + name])
    close(key + '/log_seq_prob', r['log_seq_prob'], z[key + '/log_seq_prob'])
    for l in range(2):
        close(key + '/c%d' % l, enc['states'][l][0], z[key + '/encoder_state_c%d' % l])
        close(key + '/h%d' % l, enc['states'][l][1], z[key + '/encoder_state_h%d' % l])


def check_probes(z, key, meta, got):
    assert sorted(meta) == sorted(got)
    for name, m in meta.items():
        g = np.asarray(got[name], np.float64).reshape(-1)
        want = z[key + '/' + name]
        tol = GRAD_RTOL * max(m['absmax'], 1e-30) + 1e-14
        close(key + '/' + name, g[FC.probe_indices(name, g.size)], want, tol)
        assert abs(np.sqrt(np.sum(g * g)) - m['norm']) <= GRAD_RTOL * max(m['norm'], 1e-30) + 1e-14, name
        assert abs(g.sum() - m['sum']) <= 1e-8 * max(np.abs(g).sum(), 1e-30) + 1e-14, name


def test_variable_names_and_shapes_are_what_the_reference_creates(fx):
    """The names the reference code asked `tf.get_variable` for == n2nmn_amd.spec (the weight
    registration contract of the C-ABI), for models_clevr and models_vqa."""
    z, meta = fx
    assert meta['greedy']['variables'] == sorted(variable_shapes(Dims()))
    from n2nmn_amd.vqa import vqa_variable_shapes
    d, _, _ = FC.vqa_setup()
    assert meta['vqa_greedy']['variables'] == sorted(vqa_variable_shapes(d))
    mods = [k for k in variable_shapes(Dims()) if '/module_variables/' in k]
    assert meta['modules']['variables'] == sorted(mods)


def test_greedy_forward(fx, w64):
    z, meta = fx
    d, batch = FC.clevr_inputs('greedy')
    r = O.forward(w64, FC.NAMES, batch, d.T_decoder, d.num_choices, np.float64)
    check_seq2seq(z, 'greedy', r)
    close('encoder_outputs', r['enc']['outputs'], z['greedy/encoder_outputs'])
    close('encoder_h_transformed', r['enc']['h_transformed'], z['greedy/encoder_h_transformed'])
    assert np.array_equal(r['validity'], z['greedy/validity'])
    close('scores', r['scores'], z['greedy/scores'])
    assert abs(np.mean(r['dec']['neg_entropy']) - meta['greedy']['entropy_reg']) < TOL
    ls = O.losses(w64, r['scores'], batch['answer_label_batch'], r['log_seq_prob'])
    assert abs(ls['l2_reg'] - meta['greedy']['l2_reg']) < 1e-9 * meta['greedy']['l2_reg']
    # the stub really batched per (operator, depth): some call saw more than one instance
    assert max(nb for _, _, nb in meta['greedy']['fold_batches']) > 1


def test_gt_layout_training_step(fx, w64):
    """losses, logits, EVERY variable's gradient (probes), per-tensor clip, one Adam step."""
    z, meta = fx
    m = meta['gt']
    d, batch = FC.clevr_inputs('gt')
    gt = FC.gt_layouts(d)
    losses, grads, ex = G.loss_and_grads(w64, FC.NAMES, batch, d.T_decoder, d.num_choices, gt,
                                         m['weight_decay'])
    close('scores', ex['scores'], z['gt/scores'])
    close('log_seq_prob', ex['log_seq_prob'], z['gt/log_seq_prob'])
    for k in ('total_loss', 'avg_sample_loss', 'seq_likelihood_loss', 'l2_reg'):
        assert abs(losses[k] - m[k]) <= 1e-10 * max(1.0, abs(m[k])), k
    check_probes(z, 'gt/grad', m['grad'], grads)
    clipped = {k: G.clip_by_norm(g, m['max_grad_l2_norm']) for k, g in grads.items()}
    check_probes(z, 'gt/clipped', m['clipped'], clipped)
    zeros = {k: np.zeros_like(v) for k, v in w64.items()}
    w1, _, _ = G.adam_step(w64, grads, zeros, zeros, 1, max_grad_l2_norm=m['max_grad_l2_norm'])
    check_probes(z, 'gt/adam_w1', m['adam_w1'], w1)
    # all fourteen operators ran in this batch
    ops = {name for name, _, _ in m['fold_batches']}
    assert len(ops) == 14, ops


@pytest.mark.parametrize('key', ['sampled', 'sampled_inv'])
def test_sampled_decoding_and_policy_gradient(fx, w64, key):
    z, meta = fx
    m = meta[key]
    d, batch = FC.clevr_inputs('sampled')
    u = FC.sample_uniforms(d)
    r = O.forward(w64, FC.NAMES, batch, d.T_decoder, d.num_choices, np.float64, sample_uniforms=u)
    if key == 'sampled':
        check_seq2seq(z, key, r)
        close('scores', r['scores'], z[key + '/scores'])
        assert np.array_equal(r['validity'], z[key + '/validity'])
        # sampling really left the greedy path somewhere
        g = O.forward(w64, FC.NAMES, batch, d.T_decoder, d.num_choices, np.float64)
        assert not np.array_equal(g['dec']['predicted_tokens'], r['dec']['predicted_tokens'])
    validity_in = z[key + '/validity_in']
    losses, grads, ex = G.loss_and_grads_rl(
        w64, FC.NAMES, batch, d.T_decoder, d.num_choices, r['dec']['predicted_tokens'],
        r['dec']['token_validity'], m['baseline_before'], m['invalid_expr_loss'],
        m['lambda_entropy'], m['weight_decay'], m['baseline_decay'],
        validity_override=validity_in)
    for k in ('total_loss', 'avg_sample_loss', 'policy_gradient_loss', 'entropy_reg', 'l2_reg'):
        assert abs(losses[k] - m[k]) <= 1e-10 * max(1.0, abs(m[k])), k
    assert abs(losses['new_baseline'] - m['baseline_after']) < 1e-12
    check_probes(z, key + '/grad', m['grad'], grads)


def test_direct_operator_calls(fx, w64):
    """Every Modules.<X>Module of models_clevr/nmn3_modules.py at Nb = 3 with explicit attention
    inputs (the exp_shapes/visualize_shapes.ipynb calling pattern)."""
    z, meta = fx
    d, x = FC.module_inputs()
    feat = x['image_feat'].astype(np.float64)[x['batch_idx']]
    flat = x['word_vecs'].astype(np.float64).reshape(-1, d.embed_dim_txt)
    txt = flat[x['time_idx'] * d.N + x['batch_idx']]
    a0, a1 = x['input_0'].astype(np.float64), x['input_1'].astype(np.float64)
    Nb = len(x['time_idx'])
    got = {
        'SceneModule': O.m_scene(w64, Nb, d.H, d.W, np.float64),
        'FindModule': O.m_find(w64, feat, txt),
        'FilterModule': O.m_filter(w64, a0, feat, txt),
        'FindSamePropertyModule': O.m_find_same_property(w64, a0, feat, txt),
        'TransformModule': O.m_transform(w64, a0, txt),
        'AndModule': O.m_and(a0, a1),
        'OrModule': O.m_or(a0, a1),
        'ExistModule': O.m_exist(w64, a0),
        'CountModule': O.m_count(w64, a0),
        'EqualNumModule': O.m_equal_num(w64, a0, a1),
        'MoreNumModule': O.m_more_num(w64, a0, a1),
        'LessNumModule': O.m_less_num(w64, a0, a1),
        'SamePropertyModule': O.m_same_property(w64, a0, a1, feat, txt),
        'DescribeModule': O.m_describe(w64, a0, feat, txt),
    }
    assert sorted(got) == sorted(name for name, _ in FC.MODULE_CALLS)
    for name, val in got.items():
        close(name, val, z['modules/' + name])


@pytest.mark.parametrize('mode', ['greedy', 'gt'])
def test_vqa_model(fx, mode):
    """models_vqa/nmn3_model.py incl. add_spatial_coordinate_map and question_prior_net."""
    z, meta = fx
    d, batch, gt = FC.vqa_setup()
    w = {k: v.astype(np.float64) for k, v in FC.vqa_weights(d).items()}
    kw = dict(use_gt_layout=True, gt_layout=gt) if mode == 'gt' else {}
    r = O.forward_vqa(w, batch, d.T_decoder, d.num_choices, np.float64, use_qpn=True, **kw)
    key = 'vqa_' + mode
    r['log_seq_prob'] = np.sum(np.log(r['dec']['token_probs']), axis=0)
    check_seq2seq(z, key, r)
    assert np.array_equal(r['validity'], z[key + '/validity'])
    close('scores', r['scores'], z[key + '/scores'], 1e-9)
    if mode == 'gt':
        assert {n for n, _, _ in meta[key]['fold_batches']} == \
            {'FindModule', 'TransformModule', 'AndModule', 'DescribeModule'}


@pytest.mark.parametrize('mode', ['greedy', 'gt'])
def test_shapes_model(fx, mode):
    """models_shapes (configs[0]): convnet features, layout generator without the validity automaton,
    Find / Transform / And / Answer; variable names as the reference's ScopedLayers create them."""
    from oracle import n2nmn_oracle_shapes as S
    z, meta = fx
    d, batch, gt, w, nv_txt, nv_nmn = FC.shapes_setup()
    key = 'shapes_' + mode
    names = sorted(S.variable_shapes(nv_txt, nv_nmn))
    if mode == 'gt':
        assert meta[key]['variables'] == names
    else:       # untrained free-running layouts are all invalid: no module was ever instantiated
        assert set(meta[key]['variables']) < set(names)
    kw = dict(use_gt_layout=True, gt_layout=gt) if mode == 'gt' else {}
    r = S.forward(w, batch, **kw)
    close('image_feat_grid', r['feat'], z[key + '/image_feat_grid'])
    assert np.array_equal(r['dec']['predicted_tokens'], z[key + '/predicted_tokens'])
    for name in ('token_probs', 'neg_entropy', 'word_vecs', 'atts'):
        close(key + '/' + name, r['dec'][name], z[key + '/' + name])
    assert np.array_equal(r['validity'], z[key + '/validity'])
    close('scores', r['scores'], z[key + '/scores'])
    if mode == 'gt':
        assert r['validity'].all()
        # the Fold stand-in batched the 12 questions' operators per depth
        assert max(nb for _, _, nb in meta[key]['fold_batches']) >= 12


def test_vqa_training_step_with_dropout(fx):
    """models_vqa with encoder / decoder / question-prior dropout + the loss block of
    exp_vqa/train_vqa_gt_layout.py: losses, logits, EVERY variable's gradient, one Adam step
    (no clipping, weight_decay 0)."""
    z, meta = fx
    m = meta['vqa_train']
    d, batch, gt = FC.vqa_setup()
    batch = dict(batch, answer_label_batch=FC.vqa_labels(d))
    masks = FC.vqa_dropout_masks(d)
    w = {k: v.astype(np.float64) for k, v in FC.vqa_weights(d).items()}
    assert m['variables'] == sorted(w)
    losses, grads, ex = G.loss_and_grads_vqa(w, batch, d.T_decoder, d.num_choices, gt, masks,
                                             m['weight_decay'])
    close('scores', ex['scores'], z['vqa_train/scores'], 1e-9)
    close('log_seq_prob', ex['log_seq_prob'], z['vqa_train/log_seq_prob'])
    for k in ('total_loss', 'avg_sample_loss', 'seq_likelihood_loss'):
        assert abs(losses[k] - m[k]) <= 1e-10 * max(1.0, abs(m[k])), k
    check_probes(z, 'vqa_train/grad', m['grad'], grads)
    zeros = {k: np.zeros_like(v) for k, v in w.items()}
    w1, _, _ = G.adam_step(w, grads, zeros, zeros, 1, max_grad_l2_norm=None)
    check_probes(z, 'vqa_train/adam_w1', m['adam_w1'], w1)
    # dropout really changed the result: the no-dropout logits differ
    assert np.abs(ex['scores'] - z['vqa_gt/scores']).max() > 1e-3


@pytest.mark.skipif(not os.path.isdir('/root/reference/models_clevr'),
                    reason='reference checkout not present (GPU box)')
def test_fixture_is_what_the_reference_code_computes_today():
    """Re-run the reference code under the stub and compare with the committed fixture."""
    gen = os.path.join(os.path.dirname(GOLDEN), 'make_float_golden.py')
    p = subprocess.run([sys.executable, gen, '--check'], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


# ---- BASELINE size: N = 64, T_encoder = 45, T_decoder = 20 (exp_clevr/eval_clevr.py:27-37) ----------
GOLDEN_FULL = os.path.join(os.path.dirname(GOLDEN), 'float_golden_full.npz')


def test_full_size_forward_greedy_and_gt(w64):
    """the oracle at the benchmarked size against the reference code's own logits and tokens
    (tests/golden/float_golden_full.npz, make_float_golden_full.py)"""
    from n2nmn_amd import synth
    z = np.load(GOLDEN_FULL)
    d, batch = FC.clevr_inputs('full')
    assert (d.N, d.T_encoder, d.T_decoder) == (64, 45, 20)
    r = O.forward(w64, FC.NAMES, batch, d.T_decoder, d.num_choices, np.float64)
    assert np.array_equal(r['dec']['predicted_tokens'], z['greedy/predicted_tokens'])
    assert np.array_equal(r['validity'], z['greedy/validity'])
    close('full greedy scores', r['scores'], z['greedy/scores'])
    close('full greedy token_probs', r['dec']['token_probs'], z['greedy/token_probs'])
    gt = synth.template_layout_batch(d)
    r = O.forward(w64, FC.NAMES, batch, d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                  gt_layout=gt)
    close('full gt scores', r['scores'], z['gt/scores'])
    close('full gt log_seq_prob', r['log_seq_prob'], z['gt/log_seq_prob'])


@pytest.mark.skipif(not os.path.isdir('/root/reference/models_clevr'),
                    reason='reference checkout not present (GPU box)')
def test_full_size_fixture_is_what_the_reference_code_computes_today():
    gen = os.path.join(os.path.dirname(GOLDEN), 'make_float_golden_full.py')
    p = subprocess.run([sys.executable, gen, '--check'], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


# ---- models_clevr with LSTM dropout (encoder_dropout = decoder_dropout = True) ------------------------
GOLDEN_DROPOUT = os.path.join(os.path.dirname(GOLDEN), 'float_golden_dropout.npz')


def test_clevr_dropout_fixture_holds_what_the_gpu_test_reads():
    """tests/test_gpu_reference_fixture.py::test_clevr_face_with_lstm_dropout_matches_reference_code
    compares the HIP path with this fixture directly; here: its keys, shapes, and that the recorded
    masks moved the result by several times the 1e-4 parity bar (a face that ignored the flags fails)."""
    z = np.load(GOLDEN_DROPOUT)
    d, batch = FC.clevr_inputs('gt')
    for mode in ('greedy', 'gt'):
        assert z[mode + '/predicted_tokens'].shape == (d.T_decoder, d.N)
        assert z[mode + '/scores'].shape == (d.N, d.num_choices) and z[mode + '/validity'].all()
    assert np.array_equal(z['gt/predicted_tokens'], FC.gt_layouts(d))
    assert float(z['gt/log_seq_prob_without_dropout_maxdiff']) > 5e-4
    m = FC.clevr_dropout_masks(d)
    assert m['enc0'].shape == (d.T_encoder, d.N, d.lstm_dim) and 0.45 < m['enc0'].mean() < 0.55
    # the autograd oracle's teacher-forced seq2seq takes the same masks: log_seq_prob must agree
    import torch
    w = {k: torch.as_tensor(v.astype(np.float64)) for k, v in FC.clevr_weights().items()}
    enc = G.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], drop0=m['enc0'])
    dec = G.decoder_forward_gt(w, enc, d.T_decoder, FC.gt_layouts(d), drop0=m['dec0'])
    tp = dec['token_probs'].detach().numpy()
    close('gt token_probs with dropout (autograd oracle)', tp, z['gt/token_probs'], 1e-9)
    close('gt log_seq_prob with dropout (autograd oracle)', np.sum(np.log(tp), axis=0),
          z['gt/log_seq_prob'], 1e-9)


@pytest.mark.skipif(not os.path.isdir('/root/reference/models_clevr'),
                    reason='reference checkout not present (GPU box)')
def test_clevr_dropout_fixture_is_what_the_reference_code_computes_today():
    gen = os.path.join(os.path.dirname(GOLDEN), 'make_float_golden_dropout.py')
    p = subprocess.run([sys.executable, gen, '--check'], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
