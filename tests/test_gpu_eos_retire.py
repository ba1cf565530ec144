"""Inference option N2NMN_S2S_EOS_RETIRE (include/n2nmn.h): a teacher-forced pass runs a
row's decoder only up to its layout's first <eos>.

The reference's decoder runs all T_dec = 20 steps for every row (models_clevr/nmn3_netgen_att.py:270), but
`exp_clevr/eval_clevr.py:103-135` fetches only predicted_tokens and scores; a layout is read up to its first
<eos> (models_clevr/nmn3_assembler.py:153-170) and a module's text attention is the decoder step of its own
token (models_clevr/nmn3_modules.py:53-57), so the steps from the <eos> on feed neither fetch.  The contract
tested here:
  * predicted_tokens, scores and validity of a retired pass equal the full decoder's pass in EVERY slot of a
    1024-row pass (bit for bit where the same kernels run: layouts given as device tensors, no nesting; 2e-6
    where the step tile is chosen from the host's copy of the lengths or the walker's level routing differs),
    on the template mix (mean 3.2 tokens) and on CLEVR-like layouts of up to 19 tokens, in both throughput
    modes;
  * the decoder's own outputs at ALL T steps are available on demand (Engine.decoder_outputs: atts,
    token_probs, neg_entropy, word_vecs, log_seq_prob) and equal the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
NAMES = list(CLEVR_MODULE_NAMES)
K = 16


@pytest.fixture(scope='module')
def bucket():
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    d = Dims()
    sb = SuperBucket(d, Assembler(NAMES), K=K)
    w = synth.make_weights(d, seed=0)
    sb.load_weights(w)
    batches = [synth.make_inputs(d, seed=700 + k, min_len=1) for k in range(K)]
    return sb, d, w, batches


def _layouts(d, mix, k):
    if mix == 'templates':
        return synth.template_layout_batch(d, offset=k)
    return synth.clevr_like_layout_batch(d, seed=40 + k)


def _pass(sb, retire, n=K):
    sc, tok, val = sb.run(use_gt_layout=True, eos_retire=retire, n_slots=n)
    torch.cuda.synchronize()
    return t2n(sc).copy(), t2n(tok).copy(), t2n(val).copy()


@pytest.mark.parametrize('mode', ['throughput', 'throughput_bf16x3'])
@pytest.mark.parametrize('mix', ['templates', 'clevr_like'])
@pytest.mark.parametrize('host_lengths', [True, False])
def test_retired_pass_equals_the_full_decoder_pass_in_every_slot(bucket, mode, mix, host_lengths):
    sb, d, w, batches = bucket
    eng = sb.engine
    gts = [_layouts(d, mix, k) for k in range(K)]
    for k in range(K):
        # (a layout handed over as a device tensor leaves no host copy of its length: the decoder then
        # issues all T_dec + 1 step launches and every one finds its live rows on the device)
        sb.fill(k, batches[k], gts[k] if host_lengths else torch.as_tensor(gts[k]).cuda())
    eng.set_mode(mode)
    try:
        for _ in range(3 if mix == 'clevr_like' else 1):      # (the walker's level count settles)
            full = _pass(sb, False)
        got = _pass(sb, True)
        again = _pass(sb, True)
    finally:
        eng.set_mode('latency')
    gt_all = np.concatenate(gts, 1)
    assert np.array_equal(got[1], gt_all) and np.array_equal(full[1], gt_all)
    assert np.array_equal(got[2], full[2]) and got[2].all()
    assert np.array_equal(got[0], again[0])
    if mix == 'templates' and not host_lengths:
        assert np.array_equal(got[0], full[0]), 'same kernels, same row arithmetic: the logits must be the same bits'
    worst = 0.0
    for k in range(K):
        c = slice(k * d.N, (k + 1) * d.N)
        worst = max(worst, assert_close('slot %d retired vs full' % k, got[0][c], full[0][c], 2e-6))
    lens = (gt_all != sb.engine.assembler.EOS_idx).sum(0)
    print('%s / %s: mean layout length %.2f (max %d); worst |retired - full| = %.2e' %
          (mode, mix, lens.mean(), lens.max(), worst))
    # one slot against the oracle (the full pass is pinned slot by slot in test_gpu_bench_config.py)
    k = 5
    ref = O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                    gt_layout=gts[k])
    assert_close('retired slot vs oracle', got[0][k * d.N:(k + 1) * d.N], ref['scores'], 1e-4)


def test_partial_pass_and_degenerate_layouts(bucket):
    """a 3-slot pass (192 rows: the K-split tail tiles from step 0 on) with layouts that start with <eos>
    (no live step at all: invalid, zero logits), layouts without any <eos> (every step stays) and garbage
    behind the first <eos> (ignored, as the reference's assembler ignores it)"""
    sb, d, w, batches = bucket
    asm = sb.engine.assembler
    gts = [synth.template_layout_batch(d, offset=k).copy() for k in range(3)]
    gts[0][:, 3] = asm.EOS_idx                                  # starts with <eos>
    gts[1][:, 5] = asm.name2idx_dict['_Find']                   # never ends
    gts[2][6:, 7] = asm.name2idx_dict['_Transform']             # tokens behind the first <eos>
    for k in range(3):
        sb.fill(k, batches[k], gts[k])
    sb.engine.set_mode('throughput')
    try:
        full = _pass(sb, False, 3)
        got = _pass(sb, True, 3)
    finally:
        sb.engine.set_mode('latency')
    assert np.array_equal(got[1], full[1]) and np.array_equal(got[2], full[2])
    assert not got[2][3] and not got[2][d.N + 5] and got[2][2 * d.N + 7]
    assert_close('retired vs full', got[0], full[0], 2e-6)
    assert np.all(got[0][3] == 0) and np.all(got[0][d.N + 5] == 0)


@pytest.mark.parametrize('mode', ['throughput', 'throughput_bf16x3'])
def test_decoder_outputs_on_demand_equal_the_oracle_at_every_step(bucket, mode):
    """training and the debug fetches need every decoder step: after a retired pass the decoder's outputs
    at ALL T_dec steps come from Engine.decoder_outputs (n2nmn_decoder_forward without the flag, on the
    encoder results the context still holds)"""
    sb, d, w, batches = bucket
    n = 2
    gts = [synth.clevr_like_layout_batch(d, seed=90 + k) for k in range(n)]
    for k in range(n):
        sb.fill(k, batches[k], gts[k])
    sb.engine.set_mode(mode)
    try:
        got = _pass(sb, True, n)
        s2s = sb.engine.decoder_outputs()
        torch.cuda.synchronize()
    finally:
        sb.engine.set_mode('latency')
    for k in range(n):
        c = slice(k * d.N, (k + 1) * d.N)
        ref = O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                        gt_layout=gts[k])
        dec = ref['dec']
        assert np.array_equal(t2n(s2s['predicted_tokens'])[:, c], gts[k])
        assert_close('atts, every step', t2n(s2s['atts'])[:, :, c], dec['atts'][..., 0], 1e-4)
        assert_close('token_probs', t2n(s2s['token_probs'])[:, c], dec['token_probs'], 1e-4)
        assert_close('neg_entropy', t2n(s2s['neg_entropy'])[c], dec['neg_entropy'], 1e-4)
        assert_close('word_vecs', t2n(s2s['word_vecs'])[:, c], dec['word_vecs'], 1e-4)
        assert_close('scores of the retired pass', got[0][c], ref['scores'], 1e-4)


def test_flag_is_ignored_where_its_preconditions_fail(clevr_engine):
    """a single batch of 64 in latency mode: the flag changes nothing (include/n2nmn.h)"""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=3)
    gt = synth.template_layout_batch(d, offset=2)
    a, _, _ = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
    a = t2n(a).copy()
    b, tok, val = eng.forward(batch, use_gt_layout=True, gt_layout=gt, eos_retire=True)
    assert np.array_equal(t2n(b), a) and np.array_equal(tok, gt) and val.all()
    out = eng.decoder_outputs()
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, use_gt_layout=True, gt_layout=gt)
    assert_close('atts', t2n(out['atts']), ref['dec']['atts'][..., 0], 1e-4)


@pytest.mark.parametrize('mode', ['throughput', 'throughput_bf16x3'])
@pytest.mark.parametrize('n_slots', [16, 3])
def test_greedy_retired_pass_equals_the_full_greedy_pass(bucket, n_slots, mode):
    """Layouts the decoder chooses itself: a row leaves the recurrence once it has emitted <eos> (after every
    step dec_compact_kernel moves the live rows to the front and the next step runs over that prefix).  The
    recurrent tile computes a row's dot products in the same order wherever the row sits, so tokens, validity
    AND logits of a retired pass equal the full decoder's in every slot -- bit for bit -- at 1024 rows and at
    192; the decoder's outputs at all steps come on demand and equal the fp64 oracle given the GPU's tokens."""
    sb, d, w, batches = bucket
    eng = sb.engine
    asm = eng.assembler
    for k in range(n_slots):
        sb.fill(k, batches[k])
    eng.set_mode(mode)
    eng.set_walk_levels(d.T_decoder - 1)     # (decoder-chosen layouts nest: a fixed level count makes the walker's
    try:                                     #  route, and with it the last bits of the logits, independent of history)
        full = _pass_greedy(sb, False, n_slots)
        got = _pass_greedy(sb, True, n_slots)
        again = _pass_greedy(sb, True, n_slots)
        s2s = eng.decoder_outputs()
        torch.cuda.synchronize()
    finally:
        eng.set_walk_levels(0)
        eng.set_mode('latency')
    assert np.array_equal(got[1], full[1]), 'tokens of a retired greedy pass differ'
    assert np.array_equal(got[2], full[2])
    assert np.array_equal(got[0], full[0]) and np.array_equal(again[0], got[0])
    lens = np.where((got[1] == asm.EOS_idx).any(0), (got[1] == asm.EOS_idx).argmax(0), d.T_decoder)
    print('greedy layouts: mean %.2f tokens, max %d; %d of %d rows end within 3 steps' %
          (lens.mean(), lens.max(), int((lens <= 3).sum()), lens.size))
    assert np.array_equal(t2n(s2s['predicted_tokens']), got[1])
    k = min(2, n_slots - 1)
    c = slice(k * d.N, (k + 1) * d.N)
    ref = O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64,
                    forced_tokens=np.ascontiguousarray(got[1][:, c]))
    assert_close('atts at every step (on demand)', t2n(s2s['atts'])[:, :, c], ref['dec']['atts'][..., 0], 1e-4)
    assert_close('token_probs (on demand)', t2n(s2s['token_probs'])[:, c], ref['dec']['token_probs'], 1e-4)
    assert_close('logits of the retired pass, given its tokens', got[0][c], ref['scores'], 1e-4)


def _pass_greedy(sb, retire, n):
    sc, tok, val = sb.run(use_gt_layout=False, eos_retire=retire, n_slots=n)
    torch.cuda.synchronize()
    return t2n(sc).copy(), t2n(tok).copy(), t2n(val).copy()


def test_sequential_retirement_needs_an_automaton_that_forces_eos():
    """dec_compact_kernel retires a row at its answer operator or <eos> and hands it <eos> with probability 1
    from then on -- exact only if the installed automaton allows nothing else there.  n2nmn_set_validity_tables /
    n2nmn_set_token_ops prove that over every reachable state; with tables that do not have the property (all-zero =
    every token always valid, what models_shapes installs) the flag must be ignored: a greedy pass with it returns what
    the pass without it returns, INCLUDING the tokens behind the first <eos> / answer operator (the full decoder keeps
    choosing by its logits there)."""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    d = Dims()
    asm = Assembler(NAMES)
    sb = SuperBucket(d, asm, K=3)
    w = synth.make_weights(d, seed=0)
    sb.load_weights(w)
    eng = sb.engine
    V = d.num_vocab_nmn
    eng.set_validity_tables(np.zeros((V, 3), np.int32), np.zeros((3, V, 4), np.int32), np.zeros((V, 4), np.int32))
    assert eng.all_tokens_valid
    for k in range(3):
        sb.fill(k, synth.make_inputs(d, seed=900 + k, min_len=1))
    eng.set_mode('throughput')
    eng.set_walk_levels(d.T_decoder - 1)
    try:
        full = _pass_greedy(sb, False, 3)
        got = _pass_greedy(sb, True, 3)
    finally:
        eng.set_walk_levels(0)
        eng.set_mode('latency')
    # without an automaton the free-running decoder emits tokens behind an <eos> / answer operator in some row:
    # exactly the rows a wrongly enabled retirement would have overwritten with <eos>
    tok = full[1]
    fin = np.isin(tok, [asm.EOS_idx] + [i for i, nme in enumerate(NAMES) if nme in
                                        ('_Exist', '_Count', '_EqualNum', '_MoreNum', '_LessNum', '_SameProperty', '_Describe')])
    first = np.where(fin.any(0), fin.argmax(0), d.T_decoder)
    behind = np.arange(d.T_decoder)[:, None] > first[None, :]
    assert (behind & (tok != asm.EOS_idx)).any(), 'the case does not exercise the difference'
    assert np.array_equal(got[1], full[1]) and np.array_equal(got[2], full[2]) and np.array_equal(got[0], full[0])
    # and the reference automaton does qualify (the speed-up of the tests above is real): same engine, tables back
    eng.set_validity_tables(asm.P, asm.W, asm.b)
    assert not eng.all_tokens_valid
