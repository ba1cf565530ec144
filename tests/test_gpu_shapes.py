"""SHAPES dimensions through the HIP path: models_shapes' module set (Find map_dim 500, Transform
with a 3x3 kernel, And, Answer = fc([min, mean, max])) and its seq2seq (lstm_dim 256, 14-word
vocabulary) are instances of the CLEVR kernels (Answer has the arithmetic of ExistModule), so the
fixture batch of BASELINE.json configs[0] runs on the GPU with teacher-forced layouts and must match
the SHAPES oracle (features from the oracle's convnet, which is not on the hot path)."""
import base64
import json
import os

import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from oracle import n2nmn_oracle_shapes as S
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES_OP = {'_Find': 1, '_Transform': 4, '_And': 5, '_Answer': 7}      # Answer -> N2NMN_OP_EXIST


def test_shapes_fixture_batch_on_gpu():
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    with open(os.path.join(HERE, 'golden', 'shapes_golden.json')) as f:
        g = json.load(f)
    imgs = np.frombuffer(base64.b64decode(g['images_u8_b64']), np.uint8).reshape(g['images_shape'])
    mean = np.frombuffer(base64.b64decode(g['image_mean_b64']), np.float32).reshape(g['image_mean_shape'])
    batch = dict(image_batch=(imgs.astype(np.float32) - mean).astype(np.float32),
                 text_seq_batch=np.array(g['text_seq'], np.int32),
                 seq_length_batch=np.array(g['seq_length'], np.int32))
    gt = np.array(g['gt_layout'], np.int32)
    sw = synth.make_weights_from_shapes(S.variable_shapes(14, 5), seed=0)
    ref = S.forward(sw, batch, use_gt_layout=True, gt_layout=gt)
    assert np.abs(ref['scores'] - np.array(g['scores_gt'])).max() < 1e-4      # same fixture, fp32 weights

    dd = S.DIMS
    d = Dims(H=3, W=3, D=dd['feat_dim'], map_dim=dd['map_dim'], embed_dim_txt=300, embed_dim_nmn=300,
             lstm_dim=dd['lstm_dim'], num_vocab_txt=14, num_vocab_nmn=5, num_choices=2,
             T_encoder=dd['T_encoder'], T_decoder=dd['T_decoder'], N=12, kernel_size=3)
    asm = Assembler(list(S.SHAPES_MODULE_NAMES), op_code=SHAPES_OP, input_num=S.ARITY,
                    output_type=S.OUT_TYPE)
    eng = Engine(d, asm)
    w = synth.make_weights(d, seed=1)                 # CLEVR-only variables: unused fillers
    mod = 'neural_module_network/layout_execution/'
    for k, v in sw.items():
        if k.startswith(mod):          # <X>Module/<X>Module/... (the ScopedLayer naming of models_shapes)
            name = k[len(mod):].split('/', 1)[1].replace('AnswerModule/', 'ExistModule/')
            w['neural_module_network/layout_execution/module_variables/' + name] = v
        elif 'image_feature_cnn' not in k:
            w[k] = v                                  # encoder / decoder: same names
    eng.load_weights(w)
    feat = ref['feat'].astype(np.float32)             # convnet output from the oracle
    s2s = eng.seq2seq(batch['text_seq_batch'], batch['seq_length_batch'], dd['T_decoder'],
                      use_gt_layout=True, gt_layout=gt, debug=True)
    assert_close('token_scores', t2n(s2s['token_scores']), ref['dec']['token_scores'], 1e-4)
    assert_close('word_vecs', t2n(s2s['word_vecs']), ref['dec']['word_vecs'], 1e-4)
    packed, validity = asm.assemble_packed(t2n(s2s['predicted_tokens']))
    assert validity.all()
    scores = t2n(eng.execute(packed, feat, s2s['word_vecs']))
    assert_close('scores', scores, ref['scores'], 1e-4)
    exprs, _ = asm.assemble(gt)
    assert exprs[0]['module'] == '_Answer'
