"""Policy-gradient training step (exp_clevr/train_clevr_rl_gt_layout.py:107-129: REINFORCE with an
EMA baseline + entropy regulariser + answer loss) on the GPU against the autograd oracle
(oracle/n2nmn_oracle_grad.py: loss_and_grads_rl, fp64).

Protocol: the decoder samples a layout per question on the GPU from caller-supplied uniforms; the
oracle then takes THOSE tokens and the validity masks the numpy oracle's automaton gives for them
(the masks depend on the tokens only), so a near-tie in the sampling cannot desynchronise the two
sides.  Compared: losses, entropy, the new baseline, the gradient of every variable, d token
logits (policy + entropy terms under the validity masks), d word_vecs.  Tolerances as
tests/test_gpu_train.py."""
import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from oracle import n2nmn_oracle_grad as G
from n2nmn_amd import synth
from n2nmn_amd.spec import CLEVR_MODULE_NAMES, Dims
from test_gpu_train import GRAD_ATOL, GRAD_RTOL, SELECTION_GAP, format_report, grad_report
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
NAMES = list(CLEVR_MODULE_NAMES)
WD = 5e-6


@pytest.fixture(scope='module')
def rl_setup():
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer
    d = Dims(T_decoder=10)
    asm = Assembler(NAMES)
    eng = Engine(d, asm)
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    return Trainer(eng, weight_decay=WD), eng, d, asm, w


def _token_validity(asm, tokens, T_dec):
    """[T_dec, N, V] masks of the automaton along the given token sequences
    (nmn3_netgen_att.py:8-15,200-203)."""
    Td, N = tokens.shape
    X = np.tile(np.array([[0, 0, T_dec]], np.int64), (N, 1))
    out = np.zeros((Td, N, asm.P.shape[0]), bool)
    for t in range(Td):
        out[t] = O.valid_tokens(X, asm.W, asm.b)
        X = X + asm.P[tokens[t]]
    return out


@pytest.mark.parametrize('seed,baseline', [(0, 0.5), (5, 2.75)])
def test_policy_gradient_step_matches_oracle(rl_setup, seed, baseline):
    tr, eng, d, asm, w = rl_setup
    eng.load_weights(w)
    batch = synth.make_inputs(d, seed=60 + seed, min_len=1)
    uni = np.random.default_rng(seed).random((d.T_decoder, d.N)).astype(np.float32)
    tr.baseline.fill_(baseline)
    losses, tokens, validity = tr.step_rl(batch, uni, update=False)   # no optimiser step
    losses = t2n(losses)
    assert validity.all()            # the automaton only lets valid layouts through
    assert len({tuple(c) for c in tokens.T}) > 5          # the batch really has sampled layouts
    tv = _token_validity(asm, tokens, d.T_decoder)
    assert tv[np.arange(d.T_decoder)[:, None], np.arange(d.N)[None], tokens].all()
    ref_l, ref_g, ex = G.loss_and_grads_rl(w, NAMES, batch, d.T_decoder, d.num_choices, tokens, tv,
                                           baseline, weight_decay=WD)
    assert_close('scores', t2n(tr.scores), ex['scores'], 1e-4)
    for i, k in ((0, 'avg_sample_loss'), (1, 'policy_gradient_loss'), (2, 'l2_reg'),
                 (3, 'total_loss'), (4, 'entropy_reg')):
        assert abs(losses[i] - ref_l[k]) <= 1e-4 * max(1.0, abs(ref_l[k])), (k, losses[i], ref_l[k])
    assert abs(float(t2n(tr.baseline)[0]) - ref_l['new_baseline']) <= 1e-5
    N, Td, E = d.N, d.T_decoder, d.embed_dim_txt
    inter_got = {
        'd_word_vecs': t2n(tr.debug_tensor('d_word_vecs', (Td, N, E))),
        'd_token_scores': t2n(tr.debug_tensor('d_token_scores', (Td, N, 16)))[:, :, :d.num_vocab_nmn],
        'd_scores': t2n(tr.debug_tensor('d_scores', (N, d.num_choices))),
    }
    inter_want = {k: ex[k] for k in inter_got}
    clear = ex['selection_gap'] >= SELECTION_GAP
    assert clear.mean() >= 0.5
    inter_got['d_word_vecs'] = inter_got['d_word_vecs'][:, clear]
    inter_want['d_word_vecs'] = inter_want['d_word_vecs'][:, clear]
    grads = {k: t2n(v) for k, v in tr.gradients().items()}
    rows = grad_report(inter_got, inter_want) + grad_report(grads, ref_g)
    bad = [r for r in rows if not r[3]]
    assert not bad, 'gradient mismatch:\n' + format_report(rows)


def test_rl_steps_update_weights_and_baseline(rl_setup):
    """a few full iterations (sampling, loss, backward, Adam with the fine-tuning rate): finite
    losses, moving baseline, changed weights; then the cloning objective still runs on the same
    trainer (the two objectives share every buffer)."""
    tr, eng, d, asm, w = rl_setup
    eng.load_weights(w)
    tr.baseline.fill_(0.5)
    before = {k: t2n(v).copy() for k, v in tr.get_weights().items()}
    rng = np.random.default_rng(1)
    b0 = 0.5
    for it in range(3):
        batch = synth.make_inputs(d, seed=80 + it, min_len=1)
        losses, tokens, validity = tr.step_rl(batch, rng.random((d.T_decoder, d.N)).astype(np.float32))
        l = t2n(losses)
        assert np.isfinite(l).all() and validity.all()
        b1 = float(t2n(tr.baseline)[0])
        assert abs(b1 - (b0 + 0.01 * (l[0] - b0))) <= 1e-5
        b0 = b1
    after = {k: t2n(v) for k, v in tr.get_weights().items()}
    assert any(np.abs(after[k] - before[k]).max() > 0 for k in before)
    assert all(np.isfinite(v).all() for v in after.values())
    batch = synth.make_inputs(d, seed=0)
    tr.forward_backward(batch, synth.template_layout_batch(d), reduce=False)
    assert np.isfinite(t2n(tr.losses)).all() and t2n(tr.losses)[4] == 0.0
