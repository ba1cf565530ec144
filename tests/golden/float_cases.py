"""Seeded inputs of the reference-code float fixture (tests/golden/float_golden.npz).

Shared by the generator (tests/golden/make_float_golden.py, which runs the reference's own model code
on these inputs) and by the tests that compare the oracle / the HIP path against the fixture.  The
fixture holds OUTPUTS only: weights and inputs are re-created here from seeds, so nothing large is
committed.  Pure numpy + n2nmn_amd.spec / synth (no GPU, no reference checkout needed).
"""
from __future__ import annotations

import zlib

import numpy as np

from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

WEIGHT_SEED = 0
NAMES = list(CLEVR_MODULE_NAMES)

# name -> (N, T_encoder, T_decoder, input seed, min question length)
CLEVR_CASES = {
    'greedy': dict(N=6, T_enc=9, T_dec=8, seed=101, min_len=1),
    'gt': dict(N=12, T_enc=11, T_dec=10, seed=102, min_len=1),
    'sampled': dict(N=6, T_enc=9, T_dec=10, seed=103, min_len=2),
    # BASELINE.json configs[1] / [2] at full size (exp_clevr/eval_clevr.py:27-37): separate fixture
    # tests/golden/float_golden_full.npz (logits, tokens, validity only)
    'full': dict(N=64, T_enc=45, T_dec=20, seed=106, min_len=5),
}
# two layouts that add the operators the ten SURVEY templates do not use (_Scene, _LessNum)
EXTRA_LAYOUTS = (('_Scene', '_Find', '_LessNum'),
                 ('_Find', '_Scene', '_Or', '_Transform', '_Describe'))
PROBES_PER_TENSOR = 64


def clevr_dims(case: str) -> Dims:
    c = CLEVR_CASES[case]
    return Dims(N=c['N'], T_encoder=c['T_enc'], T_decoder=c['T_dec'])


def clevr_weights(dtype=np.float32):
    """The seed-0 synthetic weights every CLEVR test uses (shapes do not depend on N / T)."""
    return synth.make_weights(Dims(), seed=WEIGHT_SEED, dtype=dtype)


def clevr_inputs(case: str):
    c = CLEVR_CASES[case]
    d = clevr_dims(case)
    batch = synth.make_inputs(d, seed=c['seed'], min_len=c['min_len'])
    batch['seq_length_batch'][0] = c['min_len']            # always one shortest question
    batch['input_seq_batch'][c['min_len']:, 0] = 0
    return d, batch


def gt_layouts(d: Dims) -> np.ndarray:
    """[T_dec, N] int32: the ten SURVEY 8(d) templates, then EXTRA_LAYOUTS."""
    cols = [synth.module_list2tokens(t, d.T_decoder)
            for t in list(synth.CLEVR_LAYOUT_TEMPLATES) + list(EXTRA_LAYOUTS)]
    assert len(cols) == d.N
    return np.ascontiguousarray(np.array(cols, np.int32).T)


def clevr_dropout_masks(d: Dims, seed: int = 108):
    """{0, 1} keep masks (keep_prob 0.5) of DropoutWrapper on the output of LSTM layer 0, per step:
    'enc0' [T_enc, N, L], 'dec0' [T_dec, N, L] (models_clevr/nmn3_netgen_att.py:17-44)."""
    rng = np.random.default_rng(seed)
    return dict(enc0=(rng.random((d.T_encoder, d.N, d.lstm_dim)) < 0.5).astype(np.float32),
                dec0=(rng.random((d.T_decoder, d.N, d.lstm_dim)) < 0.5).astype(np.float32))


def sample_uniforms(d: Dims, seed: int = 7) -> np.ndarray:
    return np.random.default_rng(seed).random((d.T_decoder, d.N))


# ---- direct operator calls (exp_shapes/visualize_shapes.ipynb style: explicit attention inputs)
MODULE_CALLS = (          # (method, number of attention inputs)
    ('SceneModule', 0), ('FindModule', 0), ('FilterModule', 1), ('FindSamePropertyModule', 1),
    ('TransformModule', 1), ('AndModule', 2), ('OrModule', 2), ('ExistModule', 1),
    ('CountModule', 1), ('EqualNumModule', 2), ('MoreNumModule', 2), ('LessNumModule', 2),
    ('SamePropertyModule', 2), ('DescribeModule', 1))


def module_inputs(Nb: int = 3, N: int = 4, T: int = 5, seed: int = 104):
    d = Dims(N=N, T_decoder=T)
    feat = synth.make_inputs(d, seed=seed)['image_feat_batch']
    rng = np.random.default_rng(seed)
    word_vecs = (0.5 * rng.standard_normal((T, N, d.embed_dim_txt))).astype(np.float32)
    in0 = (2.0 * rng.standard_normal((Nb, d.H, d.W, 1))).astype(np.float32)
    in1 = (2.0 * rng.standard_normal((Nb, d.H, d.W, 1))).astype(np.float32)
    time_idx = rng.integers(0, T, size=Nb).astype(np.int32)
    batch_idx = rng.integers(0, N, size=Nb).astype(np.int32)
    return d, dict(image_feat=feat, word_vecs=word_vecs, input_0=in0, input_1=in1,
                   time_idx=time_idx, batch_idx=batch_idx)


# ---- gradient probes: a fixture cannot hold 9.3 M gradients, so per variable it stores the L2 norm,
# the sum, max|g| and the values at PROBES_PER_TENSOR seeded flat positions
def probe_indices(name: str, numel: int) -> np.ndarray:
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    k = min(PROBES_PER_TENSOR, numel)
    return np.sort(rng.choice(numel, size=k, replace=False))


def probe(name: str, tensor) -> dict:
    flat = np.asarray(tensor, np.float64).reshape(-1)
    return dict(norm=float(np.sqrt(np.sum(flat * flat))), sum=float(flat.sum()),
                absmax=float(np.max(np.abs(flat))) if flat.size else 0.0,
                values=flat[probe_indices(name, flat.size)])


# ---- models_vqa case (exp_vqa/eval_vqa2.py dimensions, few questions)
VQA_CASE = dict(N=4, T_enc=7, T_dec=6, seed=105, min_len=1)
VQA_LAYOUTS = (('_Find', '_Describe'), ('_Find', '_Find', '_And', '_Describe'),
               ('_Find', '_Transform', '_Describe'), ('_Find', '_Transform', '_Find', '_And', '_Describe'))


def vqa_setup():
    from n2nmn_amd.vqa import VQADims, vqa_variable_shapes, VQA_MODULE_NAMES
    c = VQA_CASE
    d = VQADims(N=c['N'], T_encoder=c['T_enc'], T_decoder=c['T_dec'])
    rng = np.random.default_rng(c['seed'])
    feat = np.maximum(rng.standard_normal((d.N, d.H, d.W, d.D)), 0).astype(np.float32)
    lens = rng.integers(c['min_len'], d.T_encoder + 1, size=d.N).astype(np.int32)
    lens[0] = c['min_len']
    seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, d.N)).astype(np.int32)
    seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
    batch = dict(input_seq_batch=seq, seq_length_batch=lens, image_feat_batch=feat)
    gt = np.ascontiguousarray(np.array(
        [synth.module_list2tokens(t, d.T_decoder, VQA_MODULE_NAMES) for t in VQA_LAYOUTS],
        np.int32).T)
    return d, batch, gt


def vqa_labels(d):
    """answer_label_batch of the VQA training case."""
    return (np.random.default_rng(VQA_CASE['seed'] + 1).integers(0, d.num_choices, size=d.N)
            .astype(np.int32))


def vqa_dropout_masks(d):
    """{0, 1} keep masks of the VQA training case (keep_prob 0.5): TF draws them from its RNG, here
    they are inputs.  enc0 / dec0: DropoutWrapper on the output of LSTM layer 0 (per step);
    qpn_h / qpn_fc1: the two tf.nn.dropout calls of question_prior_net."""
    rng = np.random.default_rng(VQA_CASE['seed'] + 2)
    bern = lambda *shape: (rng.random(shape) < 0.5).astype(np.float32)
    return dict(enc0=bern(d.T_encoder, d.N, d.lstm_dim), dec0=bern(d.T_decoder, d.N, d.lstm_dim),
                qpn_h=bern(d.N, d.num_layers * d.lstm_dim), qpn_fc1=bern(d.N, d.qpn_hidden))


def vqa_weights(d, dtype=np.float32):
    from n2nmn_amd.vqa import vqa_variable_shapes
    return synth.make_weights_from_shapes(vqa_variable_shapes(d), seed=WEIGHT_SEED, dtype=dtype)


def shapes_setup():
    """SHAPES case (BASELINE.json configs[0]): the 12 questions of tests/golden/shapes_golden.json
    (the reference's own dataset files: images, text, ground-truth layouts) with seed-0 weights."""
    import base64
    import json
    import os
    from oracle import n2nmn_oracle_shapes as S
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shapes_golden.json')) as f:
        g = json.load(f)
    img = np.frombuffer(base64.b64decode(g['images_u8_b64']), np.uint8).reshape(g['images_shape'])
    mean = np.frombuffer(base64.b64decode(g['image_mean_b64']), np.float32).reshape(g['image_mean_shape'])
    batch = dict(image_batch=(img.astype(np.float32) - mean).astype(np.float32),
                 text_seq_batch=np.array(g['text_seq'], np.int32),
                 seq_length_batch=np.array(g['seq_length'], np.int32))
    gt = np.array(g['gt_layout'], np.int32)
    w = synth.make_weights_from_shapes(S.variable_shapes(len(g['vocab']), len(g['layout_vocab'])),
                                       seed=0, dtype=np.float64)
    return S.DIMS, batch, gt, w, len(g['vocab']), len(g['layout_vocab'])
