"""End-to-end parity of the hot path (exp_clevr/eval_clevr.py:103-135): answer logits within 1e-4
of the fp64 oracle on identical inputs, for BASELINE.json configs[1] (fixed gt layouts) and
configs[2] (layouts from the greedy decoder), plus the reference-shaped session loop."""
import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from n2nmn_amd.spec import CLEVR_MODULE_NAMES, INVALID_EXPR
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
TOL = 1e-4
NAMES = list(CLEVR_MODULE_NAMES)


def test_config2_gt_layouts(clevr_engine):
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=0)
    gt = synth.template_layout_batch(d)
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                    gt_layout=gt)
    assert ref['validity'].all()
    scores, tokens, validity = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
    assert np.array_equal(tokens, gt) and validity.all()
    assert_close('scores', t2n(scores), ref['scores'], TOL)
    assert np.array_equal(np.argmax(t2n(scores), 1), np.argmax(ref['scores'], 1))


def test_gt_layout_host_and_device_paths_agree(clevr_engine):
    """gt_layout as a host array: program assembled up front, no token fetch (sync-free step);
    as a device tensor: predicted_tokens fetched between the phases (the eval_clevr.py flow).
    Same tokens, bit-identical logits -- also over several steps in a row (pinned upload ring)."""
    import torch
    eng, d, asm, w = clevr_engine
    for step in range(6):
        batch = synth.make_inputs(d, seed=40 + step)
        gt = synth.template_layout_batch(d, offset=step)
        s_host, t_host, v_host = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
        s_host = t2n(s_host).copy()
        s_dev, t_dev, v_dev = eng.forward(batch, use_gt_layout=True,
                                          gt_layout=torch.as_tensor(gt).to(eng.device))
        assert np.array_equal(t_host, gt) and np.array_equal(t_dev, gt)
        assert np.array_equal(v_host, v_dev)
        assert np.array_equal(s_host, t2n(s_dev))


def test_config3_greedy_layouts(clevr_engine):
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=21)
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64)
    # phase 1 with the oracle's tokens forced (near-ties may legitimately flip free-running tokens)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'],
                      forced_tokens=ref['dec']['predicted_tokens'])
    tokens = t2n(s2s['predicted_tokens'])
    packed, validity = asm.assemble_packed(tokens)
    assert validity.all()
    scores = eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs'])
    assert_close('scores', t2n(scores), ref['scores'], TOL)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_random_valid_trees(clevr_engine, seed):
    """Deep / wide random layouts (every operator, shared images, up to 19 modules per tree)."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=30 + seed)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=seed,
                                      max_len=[4, 9, None][seed - 1])
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64,
                    forced_tokens=toks)
    assert ref['validity'].all()
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks)
    packed, validity = asm.assemble_packed(t2n(s2s['predicted_tokens']))
    assert validity.all()
    scores = t2n(eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs']))
    assert_close('scores', scores, ref['scores'], TOL)


def test_invalid_layouts_give_zero_logits(clevr_engine):
    """INVALID_EXPR -> zeros(num_choices) (models_clevr/nmn3_model.py:146,155)."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=40, n=16)
    rng = np.random.default_rng(3)
    toks = rng.integers(0, 15, size=(d.T_decoder, 16)).astype(np.int32)
    toks[:, 0] = asm.module_list2tokens(['_Find', '_Count'], d.T_decoder)
    toks[:, 5] = asm.module_list2tokens(['_Find', '_Find', '_SameProperty'], d.T_decoder)
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks)
    exprs, validity = asm.assemble(t2n(s2s['predicted_tokens']))
    assert validity.tolist() == ref['validity'].tolist() and validity[0] and validity[5]
    assert not validity.all()
    scores = t2n(eng.execute(exprs.packed, batch['image_feat_batch'], s2s['word_vecs']))
    assert_close('scores', scores, ref['scores'], TOL)
    assert (scores[~validity] == 0).all()
    # all-invalid batch: nothing to launch, all zeros
    toks[:] = 5
    packed, validity = asm.assemble_packed(toks)
    assert not validity.any() and packed.num_nodes == 0
    scores = t2n(eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs']))
    assert scores.shape == (16, d.num_choices) and (scores == 0).all()


def test_dict_walk_equals_token_path(clevr_engine):
    """compiler.build_feed_dict(plain expr dicts) == packed token path."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=41, n=10)
    gt = synth.template_layout_batch(d, n=10)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], use_gt_layout=True,
                      gt_layout=gt)
    exprs, _ = asm.assemble(gt)
    a = t2n(eng.execute(exprs.packed, batch['image_feat_batch'], s2s['word_vecs'])).copy()
    b = t2n(eng.execute(asm.pack_expr_list([dict(e) for e in exprs]),
                        batch['image_feat_batch'], s2s['word_vecs']))
    assert np.array_equal(a, b)


def test_reference_shaped_session_loop(clevr_engine):
    """The loop body of exp_clevr/eval_clevr.py:103-135 on our drop-in objects."""
    from n2nmn_amd.nmn3_model import NMN3Model
    from n2nmn_amd.runtime import Session, placeholder
    eng, d, asm, w = clevr_engine
    sess = Session()
    input_seq_batch = placeholder('int32', [None, None])
    seq_length_batch = placeholder('int32', [None])
    image_feat_batch = placeholder('float32', [None, d.H, d.W, d.D])
    expr_validity_batch = placeholder('bool', [None])
    model = NMN3Model(image_feat_batch, input_seq_batch, seq_length_batch, T_decoder=d.T_decoder,
                      num_vocab_txt=d.num_vocab_txt, embed_dim_txt=d.embed_dim_txt,
                      num_vocab_nmn=d.num_vocab_nmn, embed_dim_nmn=d.embed_dim_nmn,
                      lstm_dim=d.lstm_dim, num_layers=d.num_layers, assembler=asm,
                      encoder_dropout=False, decoder_dropout=False, decoder_sampling=False,
                      num_choices=d.num_choices, engine=eng)
    batch = synth.make_inputs(d, seed=50)
    h = sess.partial_run_setup([model.predicted_tokens, model.scores],
                               [input_seq_batch, seq_length_batch, image_feat_batch,
                                model.compiler.loom_input_tensor, expr_validity_batch])
    tokens = sess.partial_run(h, model.predicted_tokens, feed_dict={
        input_seq_batch: batch['input_seq_batch'], seq_length_batch: batch['seq_length_batch'],
        image_feat_batch: batch['image_feat_batch']})
    assert tokens.shape == (d.T_decoder, d.N) and tokens.dtype == np.int32
    expr_list, expr_validity_array = asm.assemble(tokens)
    assert expr_validity_array.all()           # greedy decoding under the automaton
    expr_feed = model.compiler.build_feed_dict(expr_list)
    expr_feed[expr_validity_batch] = expr_validity_array
    scores_val = sess.partial_run(h, model.scores, feed_dict=expr_feed)
    assert scores_val.shape == (d.N, d.num_choices)
    # oracle on the same tokens
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=tokens)
    assert_close('scores', scores_val, ref['scores'], 1e-4)
    predictions = np.argmax(scores_val, axis=1)
    assert predictions.shape == (d.N,)
