"""N2NMN_MODE_THROUGHPUT_BF16X3 (opt-in): the recurrent contraction of passes of >= 128 rows on the bf16
matrix cores over three-way split operands (csrc/kernels_lstm_tile3.hip).  The mode must meet the SAME
bars as the exact-fp32 kernels: logits within 1e-4 of numbers produced by the reference's own code
(tests/golden/float_golden_full.npz) and of the oracle in every slot of a pass, decoder tokens under the
top-2-margin rule.  ('throughput_bf16x3' is also a parametrize id of the bench-geometry, fixture, stress and
eos_retire test files.)  Reference: models_clevr/nmn3_netgen_att.py:17-44,73-113,115-322."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import n2nmn_oracle as O
from oracle import n2nmn_oracle_batched as OB
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from util import assert_close, greedy_tokens_under_margin_rule, t2n

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import float_cases as FC  # noqa: E402

pytestmark = pytest.mark.gpu
NAMES = list(CLEVR_MODULE_NAMES)
TOL = 1e-4
GOLDEN_FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'float_golden_full.npz')


def _bucket(K, mode, weights):
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    sb = SuperBucket(Dims(), Assembler(NAMES), K=K)
    sb.load_weights(weights)
    sb.engine.set_mode(mode)
    return sb


def test_full_size_fixture_in_a_bf16x3_pass():
    """the reference code's own logits (N = 64, T_enc = 45, T_dec = 20) as slot 5 of an 8-slot pass"""
    z = np.load(GOLDEN_FULL)
    d, batch = FC.clevr_inputs('full')
    gt = synth.template_layout_batch(d)
    w = FC.clevr_weights()
    sb = _bucket(8, 'throughput_bf16x3', w)
    assert sb.engine.mode == 'throughput_bf16x3'
    for k in range(8):
        other = synth.make_inputs(d, seed=300 + k, min_len=1)
        sb.fill(k, batch if k == 5 else other, gt if k == 5 else synth.template_layout_batch(d, offset=k))
    sb.run(use_gt_layout=True)
    err = assert_close('gt scores, slot 5 of 8 (bf16x3)', t2n(sb.result(5)[0]), z['gt/scores'], TOL)
    sb.run(use_gt_layout=False)
    sc, tok, val = [t2n(x) for x in sb.result(5)]
    assert val.all()
    P, Wv, bv = O.build_validity_mats(NAMES)
    enc = O.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], np.float64)
    dec = O.decoder_forward(w, enc, P, Wv, bv, d.T_decoder, np.float64)
    assert np.array_equal(dec['predicted_tokens'], z['greedy/predicted_tokens'])
    flipped = greedy_tokens_under_margin_rule(tok, dec, 'bf16x3 pass')
    same = (tok == z['greedy/predicted_tokens']).all(axis=0)
    assert_close('greedy scores, slot 5 of 8 (bf16x3)', sc[same], z['greedy/scores'][same], TOL)
    print('bf16x3 vs reference code: gt logits %.2e; %d of %d greedy layouts differ (near-ties)' %
          (err, len(flipped), d.N))


def test_bf16x3_pass_matches_the_oracle_and_the_fp32_pass_in_every_slot():
    """16 slots (1024 rows: 16 row blocks, both jobs, every stage count) with ragged lengths: every slot
    against the fp64 oracle, and against the exact-fp32 throughput pass on the same inputs"""
    d = Dims()
    w = synth.make_weights(d, seed=0)
    K = 16
    out = {}
    host = [(synth.make_inputs(d, seed=900 + k, min_len=1), synth.template_layout_batch(d, offset=k))
            for k in range(K)]
    for mode in ('throughput', 'throughput_bf16x3'):
        sb = _bucket(K, mode, w)
        for k, (b, g) in enumerate(host):
            sb.fill(k, b, g)
        sb.run(use_gt_layout=True)
        out[mode] = t2n(sb.scores).copy()
        sb.run(use_gt_layout=True, n_slots=3)            # 192 rows: ragged last row block
        out[mode + '/3'] = t2n(sb.scores).copy()
        del sb
        torch.cuda.empty_cache()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    w64 = OB.to_torch(w, torch.float64)
    batch = dict(input_seq_batch=np.concatenate([b['input_seq_batch'] for b, _ in host], 1),
                 seq_length_batch=np.concatenate([b['seq_length_batch'] for b, _ in host]),
                 image_feat_batch=np.concatenate([b['image_feat_batch'] for b, _ in host]))
    gt = np.concatenate([g for _, g in host], 1)
    ref = OB.forward(w64, NAMES, batch, d.T_decoder, d.num_choices, True, gt)
    e3 = assert_close('bf16x3 pass vs oracle', out['throughput_bf16x3'], ref['scores'], TOL)
    e1 = assert_close('fp32 pass vs oracle', out['throughput'], ref['scores'], TOL)
    assert_close('bf16x3 3-slot pass vs oracle', out['throughput_bf16x3/3'], ref['scores'][:3 * d.N], TOL)
    dd = float(np.abs(out['throughput_bf16x3'] - out['throughput']).max())
    print('max |logit - oracle|: bf16x3 %.2e, fp32 %.2e; bf16x3 vs fp32 %.2e' % (e3, e1, dd))
    assert dd <= 2e-5


def test_bf16x3_encoder_states_and_decoder_outputs():
    """the seq2seq half alone (debug outputs): encoder outputs / states, token probabilities, attention"""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    d = Dims(N=256)
    eng = Engine(d, Assembler(NAMES))
    w = synth.make_weights(Dims(), seed=0)
    eng.load_weights(w)
    eng.set_mode('throughput_bf16x3')
    batch = synth.make_inputs(d, seed=31, min_len=1)
    gt = synth.template_layout_batch(d)
    ref_enc = O.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], np.float64)
    P, Wv, bv = O.build_validity_mats(NAMES)
    ref_dec = O.decoder_forward(w, ref_enc, P, Wv, bv, d.T_decoder, np.float64, True, gt)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], use_gt_layout=True, gt_layout=gt,
                      debug=True)
    assert_close('encoder_outputs', t2n(s2s['encoder_outputs']), ref_enc['outputs'], 2e-5)
    es = t2n(s2s['encoder_states'])
    for l in range(2):
        assert_close('c%d' % l, es[l, 0], ref_enc['states'][l][0], 2e-5)
        assert_close('h%d' % l, es[l, 1], ref_enc['states'][l][1], 2e-5)
    assert_close('token_probs', t2n(s2s['token_probs']), ref_dec['token_probs'], TOL)
    assert_close('atts', t2n(s2s['atts']), ref_dec['atts'][..., 0], TOL)
    # greedy decoding (one job per launch) from the same states
    free = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'])
    dec = O.decoder_forward(w, ref_enc, P, Wv, bv, d.T_decoder, np.float64)
    greedy_tokens_under_margin_rule(t2n(free['predicted_tokens']), dec, 'bf16x3 greedy')


def test_split_operand_gemm_is_not_part_of_the_product_library(clevr_engine):
    """gemm_dma3_kernel (split-operand bf16 form of the dense contractions) left the library in round 6: passes whose
    conv_image launch ran on it returned wrong logits under concurrent streams and the cause was not found below the
    launch level (profiles/r06_notes.md section 1; the kernel and its accuracy / timing loops live in tools/diag/).
    The debug entry refuses the form instead of silently computing in fp32."""
    import torch
    from n2nmn_amd import _lib
    eng = clevr_engine[0]
    A = torch.randn((1024, 64), device=eng.device)
    B = torch.randn((64, 128), device=eng.device)
    out = torch.zeros((1024, 128), device=eng.device)
    try:
        eng.debug_set('debug_gemm_b3', 1)                # (n2nmn_debug_set, include/n2nmn.h section 7)
        with pytest.raises(ValueError, match='not part of this library'):
            _lib.check(eng._lib.n2nmn_debug_gemm(eng._ctx, A.data_ptr(), B.data_ptr(), None, out.data_ptr(),
                                                 1024, 128, 64, eng.stream()))
        eng.debug_set('debug_gemm_b3', -2)               # two fp32 launches: allowed
        _lib.check(eng._lib.n2nmn_debug_gemm(eng._ctx, A.data_ptr(), B.data_ptr(), None, out.data_ptr(),
                                             1024, 128, 64, eng.stream()))
        torch.cuda.synchronize()
        assert float((out - A @ B).abs().max()) < 1e-3
    finally:
        eng.debug_set('debug_gemm_b3', None)


