"""BASELINE.json configs[0]: the SHAPES plumbing of exp_shapes/eval_shapes.py restated on the CPU
(oracle/n2nmn_oracle_shapes.py) against the fixture extracted from the reference's own dataset files
(tests/golden/shapes_golden.json, tests/golden/make_shapes_golden.py), plus -- when the reference
checkout is present (this container, not the GPU box) -- the fixture re-derived from those files."""
import base64
import json
import os

import numpy as np
import pytest

from oracle import n2nmn_oracle_shapes as S
from n2nmn_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('N2NMN_REFERENCE', '/root/reference')


@pytest.fixture(scope='module')
def fx():
    with open(os.path.join(HERE, 'golden', 'shapes_golden.json')) as f:
        g = json.load(f)
    g['images_u8'] = np.frombuffer(base64.b64decode(g['images_u8_b64']), np.uint8).reshape(g['images_shape'])
    g['image_mean'] = np.frombuffer(base64.b64decode(g['image_mean_b64']), np.float32).reshape(g['image_mean_shape'])
    return g


def _batch(g):
    return dict(image_batch=(g['images_u8'].astype(np.float32) - g['image_mean']).astype(np.float32),
                text_seq_batch=np.array(g['text_seq'], np.int32),
                seq_length_batch=np.array(g['seq_length'], np.int32))


def _weights(g):
    return synth.make_weights_from_shapes(S.variable_shapes(len(g['vocab']), len(g['layout_vocab'])),
                                          seed=0, dtype=np.float64)


def test_vocabularies_and_dimensions(fx):
    assert fx['layout_vocab'] == list(S.SHAPES_MODULE_NAMES)
    assert len(fx['vocab']) == 14 and fx['num_questions'] == 64          # train.tiny
    assert fx['images_u8'].shape == (12, 30, 30, 3) and fx['image_mean'].shape == (30, 30, 3)
    assert np.array(fx['text_seq']).shape == (S.DIMS['T_encoder'], 12)
    assert np.array(fx['gt_layout']).shape == (S.DIMS['T_decoder'], 12)


def test_gt_layouts_assemble_valid(fx):
    exprs, validity = S.assemble(np.array(fx['gt_layout'], np.int32))
    assert validity.all()
    assert all(e['module'] == '_Answer' for e in exprs)


def test_forward_matches_golden_outputs(fx):
    w = _weights(fx)
    b = _batch(fx)
    r = S.forward(w, b, use_gt_layout=True, gt_layout=np.array(fx['gt_layout'], np.int32))
    assert r['feat'].shape == (12, 3, 3, 64) and abs(float(r['feat'].sum()) - fx['feat_sum']) < 1e-8
    assert np.abs(r['scores'] - np.array(fx['scores_gt'])).max() < 1e-10
    free = S.forward(w, b)
    assert np.array_equal(free['dec']['predicted_tokens'], np.array(fx['tokens_free']))
    assert np.abs(free['scores'] - np.array(fx['scores_free'])).max() < 1e-10
    # the eval loop's bookkeeping (exp_shapes/eval_shapes.py:171-180)
    predictions = np.argmax(r['scores'], axis=1)
    assert predictions.shape == (12,) and set(np.unique(fx['labels'])) <= {0, 1}


def test_eos_latch(fx):
    """after the first <eos> every later token is <eos> with probability 1 (nmn3_netgen_att.py:211-222)"""
    w = _weights(fx)
    free = S.forward(w, _batch(fx))
    toks, tp = free['dec']['predicted_tokens'], free['dec']['token_probs']
    eos = list(S.SHAPES_MODULE_NAMES).index('<eos>')
    for n in range(toks.shape[1]):
        hit = np.nonzero(toks[:, n] == eos)[0]
        if hit.size:
            assert np.all(toks[hit[0]:, n] == eos) and np.all(tp[hit[0] + 1:, n] == 1.0)


def test_convnet_against_torch_conv2d(fx):
    import torch
    import torch.nn.functional as F
    w = _weights(fx)
    x = _batch(fx)['image_batch'].astype(np.float64)
    got = S.shapes_convnet(w, x)
    t = torch.as_tensor(x).permute(0, 3, 1, 2)
    k1 = torch.as_tensor(w[S._CNN + 'conv_1/weights']).permute(3, 2, 0, 1)
    c1 = F.relu(F.conv2d(t, k1, torch.as_tensor(w[S._CNN + 'conv_1/biases']), stride=10))
    k2 = torch.as_tensor(w[S._CNN + 'conv_2/weights']).permute(3, 2, 0, 1)
    c2 = F.relu(F.conv2d(c1, k2, torch.as_tensor(w[S._CNN + 'conv_2/biases'])))
    assert np.abs(got - c2.permute(0, 2, 3, 1).numpy()).max() < 1e-10


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'exp_shapes', 'shapes_dataset')),
                    reason='reference checkout not present (GPU box)')
def test_fixture_equals_reference_dataset_files(fx):
    """re-derive the fixture from the reference's files: vocabulary, seed-3 shuffle, tokenisation"""
    sp = S.load_split(REF, 'train.tiny')
    assert sp['vocab'] == fx['vocab'] and sp['layout_vocab'] == fx['layout_vocab']
    assert sp['order'][:12].tolist() == fx['order_head']
    assert np.array_equal(sp['text_seq'][:, :12], np.array(fx['text_seq']))
    assert np.array_equal(sp['gt_layout'][:, :12], np.array(fx['gt_layout']))
    assert np.array_equal(sp['images_u8'][:12], fx['images_u8'])
    assert sp['labels'][:12].tolist() == fx['labels']
    # every split of the dataset: all ground-truth layouts are valid programs
    for split in ('train.tiny', 'train.small', 'val', 'test'):
        s2 = S.load_split(REF, split)
        _, validity = S.assemble(s2['gt_layout'])
        assert validity.all(), split
        assert s2['images_u8'].shape[1:] == (30, 30, 3) and s2['seq_length'].max() <= S.DIMS['T_encoder']
