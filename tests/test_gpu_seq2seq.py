"""GPU parity of phase 1 (encoder + attentional decoder) against the fp64 oracle, following the
parity protocol of SURVEY.md 8(c): compare token_scores per step <= 1e-4 with the oracle's tokens
forced on both sides, and require identical greedy tokens wherever the oracle's top-2 margin is
> 1e-3."""
import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _oracle_enc(w, batch):
    return O.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], np.float64)


@pytest.mark.parametrize('n,seed', [(64, 0), (5, 1), (1, 2), (17, 3)])
def test_encoder(clevr_engine, n, seed):
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=seed, n=n)
    if n == 5:
        batch['seq_length_batch'][:] = [d.T_encoder, 1, 2, d.T_encoder, 7]   # extremes
        batch['input_seq_batch'][np.arange(d.T_encoder)[:, None] >=
                                 batch['seq_length_batch'][None, :]] = 0
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], debug=True,
                      phase='encoder')
    enc = _oracle_enc(w, batch)
    assert_close('encoder_outputs', t2n(out['encoder_outputs']), enc['outputs'], TOL)
    assert_close('encoder_h_transformed', t2n(out['encoder_h_transformed']),
                 enc['h_transformed'], TOL)
    st = t2n(out['encoder_states'])
    for l in range(2):
        assert_close('c%d' % l, st[l, 0], enc['states'][l][0], TOL)
        assert_close('h%d' % l, st[l, 1], enc['states'][l][1], TOL)


def _check_decoder(out, dec, d, check_tokens_margin=True):
    assert_close('token_scores', t2n(out['token_scores']), dec['token_scores'], TOL)
    assert_close('atts', t2n(out['atts']), dec['atts'][..., 0], TOL)
    assert_close('word_vecs', t2n(out['word_vecs']), dec['word_vecs'], TOL)
    assert_close('token_probs', t2n(out['token_probs']), dec['token_probs'], TOL)
    assert_close('neg_entropy', t2n(out['neg_entropy']), dec['neg_entropy'], 2e-4)
    lsp = np.sum(np.log(dec['token_probs']), axis=0)
    assert_close('log_seq_prob', t2n(out['log_seq_prob']), lsp, 5e-4)


@pytest.mark.parametrize('n,seed', [(64, 0), (6, 4)])
def test_decoder_greedy(clevr_engine, n, seed):
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=seed, n=n)
    enc = _oracle_enc(w, batch)
    dec = O.decoder_forward(w, enc, asm.P, asm.W, asm.b, d.T_decoder, np.float64)
    # (1) free-running greedy decode: tokens must agree wherever the decision is not a near-tie
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], debug=True)
    tok = t2n(out['predicted_tokens'])
    sc = np.where(dec['token_validity'], dec['token_scores'], -np.inf)
    top2 = np.sort(sc, axis=2)[:, :, -2:]
    margin = top2[:, :, 1] - top2[:, :, 0]
    first_div = np.argmax((tok != dec['predicted_tokens']) | (margin < 1e-3), axis=0)
    for i in range(n):   # compare up to the first near-tie of each question
        upto = first_div[i] if ((tok[:, i] != dec['predicted_tokens'][:, i]) | (margin[:, i] < 1e-3)).any() else d.T_decoder
        assert np.array_equal(tok[:upto, i], dec['predicted_tokens'][:upto, i])
        if upto < d.T_decoder:
            assert margin[upto, i] < 1e-3, 'token flip at a non-tie'
    # every greedy layout is valid under the automaton
    _, validity = asm.assemble_packed(tok)
    assert validity.all()
    # (2) oracle tokens forced on the GPU side -> all per-step quantities comparable
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], debug=True,
                      forced_tokens=dec['predicted_tokens'])
    assert np.array_equal(t2n(out['predicted_tokens']), dec['predicted_tokens'])
    _check_decoder(out, dec, d)


def test_decoder_teacher_forcing(clevr_engine):
    """use_gt_layout: tokens == gt layout, every token valid, probabilities not renormalised
    (nmn3_netgen_att.py:204-207,239-241); T_decoder=10 like train_clevr_gt_layout.py:35."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=11)
    T = 10
    import dataclasses
    d10 = dataclasses.replace(d, T_decoder=T)
    gt = synth.template_layout_batch(d10)
    enc = _oracle_enc(w, batch)
    dec = O.decoder_forward(w, enc, asm.P, asm.W, asm.b, T, np.float64, use_gt_layout=True,
                            gt_layout=gt)
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec=T,
                      use_gt_layout=True, gt_layout=gt, debug=True)
    assert np.array_equal(t2n(out['predicted_tokens']), gt)
    _check_decoder(out, dec, d10)


def test_decoder_sampling(clevr_engine):
    """decoder_sampling=True with caller-supplied uniforms (stand-in for tf.multinomial):
    sampled tokens are always valid; given the oracle's tokens everything else matches."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=12)
    rng = np.random.default_rng(99)
    uni = rng.random((d.T_decoder, d.N)).astype(np.float32)
    enc = _oracle_enc(w, batch)
    dec = O.decoder_forward(w, enc, asm.P, asm.W, asm.b, d.T_decoder, np.float64,
                            sample_uniforms=uni)
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], sample_uniforms=uni,
                      debug=True)
    tok = t2n(out['predicted_tokens'])
    _, validity = asm.assemble_packed(tok)
    assert validity.all()
    # sampling decisions agree except within fp32 distance of a CDF boundary
    agree = (tok == dec['predicted_tokens']).all(axis=0).mean()
    assert agree > 0.8, agree
    assert len(np.unique(tok[0])) >= 2          # it actually samples
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], sample_uniforms=uni,
                      forced_tokens=dec['predicted_tokens'], debug=True)
    _check_decoder(out, dec, d)


def test_short_shapes(clevr_engine):
    """T_enc and T_dec below the context capacity, ragged N."""
    eng, d, asm, w = clevr_engine
    import dataclasses
    ds = dataclasses.replace(d, T_encoder=7, T_decoder=6, N=3)
    batch = synth.make_inputs(ds, seed=13, n=3, min_len=1)
    enc = _oracle_enc(w, batch)
    dec = O.decoder_forward(w, enc, asm.P, asm.W, asm.b, 6, np.float64)
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec=6, debug=True,
                      forced_tokens=dec['predicted_tokens'])
    _check_decoder(out, dec, ds)


def test_capacity_and_argument_errors(clevr_engine):
    eng, d, asm, w = clevr_engine
    import dataclasses
    big = dataclasses.replace(d, N=d.N + 1)
    batch = synth.make_inputs(big, seed=1, n=d.N + 1)
    with pytest.raises(ValueError):
        eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'])
    batch = synth.make_inputs(d, seed=1, n=4)
    with pytest.raises(ValueError):
        eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], use_gt_layout=True)


@pytest.mark.parametrize('n,seed', [(64, 10), (37, 11), (3, 12)])
def test_throughput_mode_tiles(clevr_engine, n, seed):
    """N2NMN_MODE_THROUGHPUT switches the recurrent step kernels to 32-row x 32-column workgroup
    tiles (lstm_step_wide_kernel): encoder, final states and the teacher-forced decoder must match the
    oracle exactly as in the default mode (ragged lengths, partial row blocks, N < 16)."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=seed, n=n, min_len=1)
    gt = synth.template_layout_batch(d, n=n, offset=seed)
    eng.set_mode('throughput')
    try:
        out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], use_gt_layout=True,
                          gt_layout=gt, debug=True)
        enc = _oracle_enc(w, batch)
        assert_close('encoder_outputs', t2n(out['encoder_outputs']), enc['outputs'], TOL)
        st = t2n(out['encoder_states'])
        for l in range(2):
            assert_close('c%d' % l, st[l, 0], enc['states'][l][0], TOL)
            assert_close('h%d' % l, st[l, 1], enc['states'][l][1], TOL)
        P, Wv, bv = O.build_validity_mats(list(asm.module_names))
        dec = O.decoder_forward(w, enc, P, Wv, bv, d.T_decoder, np.float64, use_gt_layout=True,
                                gt_layout=gt)
        _check_decoder(out, dec, d)
    finally:
        eng.set_mode('latency')
    from n2nmn_amd import _lib
    with pytest.raises(ValueError):
        _lib.check(eng._lib.n2nmn_ctx_set_mode(eng._ctx, 7))


def test_sequential_decoder_attention_at_wave_boundaries(clevr_engine):
    """dec_attn_seq_kernel (greedy / sampled decoding, one launch per step): wave w of a question's
    workgroup takes the encoder rows w, w + 16, ... below the question's length and the 16 running
    soft-max states meet in LDS -- lengths 1, 2, T and every multiple of 16 +- 1; the attention of the
    rows past the length is exactly 0 and the rows inside sum to 1 (nmn3_netgen_att.py:190-191)."""
    eng, d, asm, w = clevr_engine
    T = d.T_encoder
    lens = np.array([1, 2, 3, 15, 16, 17, 31, 32, 33, T - 1, T, T, 1, 16, 32, 8] * 4, np.int32)[:d.N]
    batch = synth.make_inputs(d, seed=21, n=len(lens), min_len=1)
    seq = batch['input_seq_batch'].copy()
    seq[np.arange(T)[:, None] >= lens[None, :]] = 0
    batch = dict(batch, seq_length_batch=lens, input_seq_batch=seq)
    enc = _oracle_enc(w, batch)
    dec = O.decoder_forward(w, enc, asm.P, asm.W, asm.b, d.T_decoder, np.float64)
    out = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], debug=True,
                      forced_tokens=dec['predicted_tokens'])
    _check_decoder(out, dec, d)
    atts = t2n(out['atts'])                                      # [T_dec, T_enc, N]
    past = np.arange(T)[None, :, None] >= lens[None, None, :]
    assert np.all(atts[np.broadcast_to(past, atts.shape)] == 0.0)
    assert np.abs(atts.sum(axis=1) - 1.0).max() < 1e-5
    # free-running: every layout valid, tokens under the near-tie rule
    free = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'])
    tok = t2n(free['predicted_tokens'])
    _, validity = asm.assemble_packed(tok)
    assert validity.all()
    from util import greedy_tokens_under_margin_rule
    greedy_tokens_under_margin_rule(tok, dec, 'wave-boundary lengths')
