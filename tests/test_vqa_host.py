"""models_vqa variant, host side (no GPU): variable table, the zero-padding of lstm_dim / feature
depth that n2nmn_amd.vqa applies before loading weights (checked through the ORACLE: padded and
unpadded models must agree), VQA assembler op codes, coordinate map."""
import numpy as np

from n2nmn_amd import synth, vqa
from n2nmn_amd.nmn3_assembler import Assembler
from oracle import n2nmn_oracle as O

SMALL = vqa.VQADims(H=3, W=4, D=14, map_dim=10, embed_dim_txt=8, embed_dim_nmn=8, lstm_dim=12,
                    num_vocab_txt=9, num_choices=6, T_encoder=5, T_decoder=6, N=4, qpn_hidden=7)


def _batch(d, seed=0):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, d.T_encoder + 1, size=d.N).astype(np.int32)
    seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, d.N)).astype(np.int32)
    seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
    feat = np.maximum(rng.standard_normal((d.N, d.H, d.W, d.D)), 0).astype(np.float32)
    return dict(input_seq_batch=seq, seq_length_batch=lens, image_feat_batch=feat)


def test_variable_table_of_reference_config():
    d = vqa.VQADims()
    shapes = vqa.vqa_variable_shapes(d)
    assert shapes['neural_module_network/layout_execution/module_variables/TransformModule/fc_att/'
                  'weights'] == (2050, 1024)            # D + 2 coordinate channels
    assert shapes['neural_module_network/question_prior_net/fc1/weights'] == (2000, 500)
    assert not any('FindSameProperty' in k or 'CountModule' in k for k in shapes)
    di = vqa.internal_dims(d)
    assert (di.lstm_dim, di.D, di.variant, di.qpn_hidden) == (1024, 2064, 1, 500)


def test_coordinate_map_matches_reference_layout():
    f = np.zeros((2, 3, 4, 5), np.float32)
    c = O.add_spatial_coordinate_map(f)
    assert c.shape == (2, 3, 4, 7)
    assert np.allclose(c[0, 0, :, 5], np.linspace(-1, 1, 4))     # x varies along W
    assert np.allclose(c[0, :, 0, 6], np.linspace(-1, 1, 3))     # y varies along H
    assert np.all(c[..., :5] == 0)


def test_zero_padding_leaves_the_model_unchanged():
    d = SMALL
    di = vqa.internal_dims(d)
    assert di.lstm_dim == 128 and di.D == 16
    w = synth.make_weights_from_shapes(vqa.vqa_variable_shapes(d), seed=2, dtype=np.float64)
    wp = {k: vqa.pad_variable(k, v, d, di).astype(np.float64) for k, v in w.items()}
    batch = _batch(d)
    L = d.lstm_dim
    e0 = O.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'])
    e1 = O.encoder_forward(wp, batch['input_seq_batch'], batch['seq_length_batch'])
    assert np.abs(e1['outputs'][:, :, :L] - e0['outputs']).max() < 1e-6
    assert np.abs(e1['outputs'][:, :, L:]).max() == 0.0           # padded units stay exactly 0
    P, Wv, bv = O.build_validity_mats(list(O.VQA_MODULE_NAMES))
    d0 = O.decoder_forward(w, e0, P, Wv, bv, d.T_decoder)
    d1 = O.decoder_forward(wp, e1, P, Wv, bv, d.T_decoder)
    assert np.abs(d0['token_scores'] - d1['token_scores']).max() < 1e-6
    assert np.array_equal(d0['predicted_tokens'], d1['predicted_tokens'])
    q0 = O.question_prior_net(O._cast(w, np.float64), e0['states'])
    q1 = O.question_prior_net(O._cast(wp, np.float64), e1['states'])
    assert np.abs(q0 - q1).max() < 1e-6
    # module side: padded feature channels are zero and so are the padded weight rows
    fc = O.add_spatial_coordinate_map(batch['image_feat_batch'].astype(np.float64))
    fcp = np.concatenate([fc, np.zeros(fc.shape[:3] + (di.D - fc.shape[3],))], axis=3)
    txt = d0['word_vecs'][0]
    a0 = O.vqa_find(O._cast(w, np.float64), fc, txt)
    a1 = O.vqa_find(wp, fcp, txt)
    assert np.abs(a0 - a1).max() < 1e-6
    t0 = O.vqa_transform(O._cast(w, np.float64), a0, fc, txt)
    t1 = O.vqa_transform(wp, a1, fcp, txt)
    assert np.abs(t0 - t1).max() < 1e-6


def test_forward_vqa_runs_and_adds_question_prior():
    d = SMALL
    w = synth.make_weights_from_shapes(vqa.vqa_variable_shapes(d), seed=4, dtype=np.float64)
    batch = _batch(d, 1)
    r = O.forward_vqa(w, batch, d.T_decoder, d.num_choices)
    assert r['validity'].all() and r['scores'].shape == (d.N, d.num_choices)
    r2 = O.forward_vqa(w, batch, d.T_decoder, d.num_choices, use_qpn=False)
    qpn = O.question_prior_net(O._cast(w, np.float64), r['enc']['states'])
    assert np.abs(r['scores'] - (r2['scores'] + qpn)).max() < 1e-12


def test_vqa_assembler_maps_transform_to_the_pooled_product_operator(golden):
    asm = Assembler(list(vqa.VQA_MODULE_NAMES), op_code=vqa.VQA_OP_CODE)
    toks = np.array([asm.module_list2tokens(['_Find', '_Transform', '_Describe'], 6),
                     asm.module_list2tokens(['_Find', '_Find', '_And', '_Describe'], 6)], np.int32).T
    packed, validity = asm.assemble_packed(toks)
    assert validity.all()
    ops = [int(n['op']) for n in packed.nodes()]
    assert ops == [1, 3, 13, 1, 1, 5, 13]          # N2NMN_OP_FIND, _FIND_SAME_PROPERTY, _DESCRIBE, ...
    exprs, _ = asm.assemble(toks)
    assert exprs[0]['input_0']['module'] == '_Transform'
    # validity matrices equal the reference's (golden produced by models_vqa/nmn3_assembler.py)
    g = golden['vqa']
    assert np.array_equal(asm.P, np.array(g['P'])) and np.array_equal(asm.b, np.array(g['b']))
