"""CPU ORACLE for the SHAPES variant (BASELINE.json configs[0]: exp_shapes/eval_shapes.py on the
reference's CPU-runnable case) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `models_shapes/*` on top of the shared pieces of oracle/n2nmn_oracle.py:
  * image features: shapes_convnet = conv 10x10 stride 10 VALID 3->64 + ReLU, 1x1 64->64 + ReLU
    (models_shapes/shapes_convnet.py:8-17) on mean-subtracted 30x30x3 images -> [N,3,3,64]
  * layout generator: the same encoder; the decoder has NO validity automaton -- plain argmax, an
    `is_eos_predicted` latch that forces <eos> / probability 1 / zero entropy after the first <eos>,
    neg_entropy = sum p*log(max(1e-5, p))   (models_shapes/nmn3_netgen_att.py:161-222)
  * modules: Find (map_dim 500), Transform (3x3 conv), And, Answer = fc([min, mean, max])
    (models_shapes/nmn3_modules.py:27-144)
  * data plumbing of exp_shapes/eval_shapes.py:60-114 (vocabularies, np.random.seed(3) shuffle,
    tokenisation, mean subtraction): `load_split`.
PARITY STATUS: PINNED -- the unmodified models_shapes/*.py (NMN3ModelAtt, shapes_convnet, the SHAPES
assembler) run under the eager TF1/Fold stand-in (tests/golden/make_float_golden.py::case_shapes ->
float_golden.npz `shapes_*`), and this module equals those outputs to 1e-10, variable names
included (tests/test_oracle_vs_reference_code.py); the data plumbing is pinned to the reference's
own dataset files through tests/golden/shapes_golden.json.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import n2nmn_oracle as O

SHAPES_MODULE_NAMES = ('_Find', '_Transform', '_And', '_Answer', '<eos>')   # exp_shapes/data/vocabulary_layout.txt
ARITY = {'_Find': 0, '_Transform': 1, '_And': 2, '_Answer': 1}           # models_shapes/nmn3_assembler.py:9-18
OUT_TYPE = {'_Find': 'att', '_Transform': 'att', '_And': 'att', '_Answer': 'ans'}
_P = 'neural_module_network/'
_CNN = _P + 'image_feature_cnn/shapes_convnet/'
_MOD = _P + 'layout_execution/'
# eval-time dimensions: exp_shapes/eval_shapes.py:27-39
DIMS = dict(H_im=30, W_im=30, num_choices=2, embed_dim_txt=300, embed_dim_nmn=300, lstm_dim=256,
            T_encoder=15, T_decoder=11, N=256, map_dim=500, feat_dim=64, kernel_size=3)


def variable_shapes(num_vocab_txt=14, num_vocab_nmn=5, d=DIMS):
    L, E, M, C = d['lstm_dim'], d['embed_dim_txt'], d['map_dim'], d['num_choices']
    s = {}
    s[_CNN + 'conv_1/weights'] = (10, 10, 3, 64); s[_CNN + 'conv_1/biases'] = (64,)
    s[_CNN + 'conv_2/weights'] = (1, 1, 64, d['feat_dim']); s[_CNN + 'conv_2/biases'] = (d['feat_dim'],)
    enc, dec = O._ENC, O._DEC
    s[enc + 'embedding_mat'] = (num_vocab_txt, E)
    for which, base in (('encoder', enc), ('decoder', dec)):
        for layer, k_in in ((0, E + L), (1, 2 * L)):
            b = base + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/' % layer
            s[b + 'weights'] = (k_in, 4 * L); s[b + 'biases'] = (4 * L,)
    s[enc + 'encoder_h_transform/weights'] = (L, L); s[enc + 'encoder_h_transform/biases'] = (L,)
    s[dec + 'embedding_mat'] = (num_vocab_nmn, d['embed_dim_nmn']); s[dec + 'go_embedding'] = (1, d['embed_dim_nmn'])
    s[dec + 'att_prediction/v'] = (L,)
    s[dec + 'att_prediction/weights'] = (L, L); s[dec + 'att_prediction/biases'] = (L,)
    s[dec + 'token_prediction/weights'] = (2 * L, num_vocab_nmn); s[dec + 'token_prediction/biases'] = (num_vocab_nmn,)
    for scope, name, shape in (('FindModule', 'conv_image', (d['feat_dim'], M)), ('FindModule', 'fc_text', (E, M)),
                               ('FindModule', 'conv_eltwise', (M, 1)),
                               ('TransformModule', 'conv_maps', (d['kernel_size'], d['kernel_size'], 1, M)),
                               ('TransformModule', 'text_fc', (E, M)), ('TransformModule', 'conv_eltwise', (M, 1)),
                               ('AnswerModule', 'fc_scores', (3, C))):
        # td.ScopedLayer(modules.<X>Module, name_or_scope='<X>Module') opens the scope once and the
        # module function opens it again (models_shapes/nmn3_model.py:60-80, nmn3_modules.py:27,62,
        # 112): the variables the reference creates are layout_execution/<X>Module/<X>Module/...
        # (checked by running the reference's model code, tests/test_oracle_vs_reference_code.py)
        s[_MOD + scope + '/' + scope + '/' + name + '/weights'] = shape
        s[_MOD + scope + '/' + scope + '/' + name + '/biases'] = (shape[-1],)
    return s


def shapes_convnet(w, images):
    """models_shapes/shapes_convnet.py:8-17.  images [N,30,30,3] (mean-subtracted) -> [N,3,3,64]."""
    x = np.asarray(images)
    N = x.shape[0]
    K1 = w[_CNN + 'conv_1/weights']                      # [10,10,3,64], stride 10, VALID
    patches = x.reshape(N, 3, 10, 3, 10, 3).transpose(0, 1, 3, 2, 4, 5).reshape(N, 3, 3, 300)
    c1 = np.maximum(patches @ K1.reshape(300, -1) + w[_CNN + 'conv_1/biases'], 0)
    K2 = w[_CNN + 'conv_2/weights']
    return np.maximum(c1 @ K2.reshape(K2.shape[2], K2.shape[3]) + w[_CNN + 'conv_2/biases'], 0)


def decoder_forward(w, enc, T_dec, eos_idx, dtype=np.float64, use_gt_layout=False, gt_layout=None):
    """models_shapes/nmn3_netgen_att.py:105-279: greedy or teacher-forced; <eos> latch."""
    w = O._cast(w, dtype)
    (c0, h0), (c1, h1) = enc['states']
    N = h0.shape[0]
    demb = w[O._DEC + 'embedding_mat']
    x = np.tile(w[O._DEC + 'go_embedding'], (N, 1))
    v = w[O._DEC + 'att_prediction/v']
    Wa, ba = w[O._DEC + 'att_prediction/weights'], w[O._DEC + 'att_prediction/biases']
    Wy, by = w[O._DEC + 'token_prediction/weights'], w[O._DEC + 'token_prediction/biases']
    W0, b0 = O._lstm_w(w, 'decoder', 0)
    W1, b1 = O._lstm_w(w, 'decoder', 1)
    eht, eout, nf = enc['h_transformed'], enc['outputs'], enc['not_finished']
    T_enc = eht.shape[0]
    tokens = np.zeros((T_dec, N), np.int32)
    tprobs = np.zeros((T_dec, N), dtype)
    atts = np.zeros((T_dec, T_enc, N, 1), dtype)
    scores_all = np.zeros((T_dec, N, demb.shape[0]), dtype)
    neg_ent = np.zeros(N, dtype)
    is_eos = np.zeros(N, bool)
    feats = np.zeros((T_dec, N, Wy.shape[0]), dtype)       # [h1, ctx]: what token_prediction sees (fixtures)
    for t in range(T_dec):
        c0, h0 = O._lstm_cell(x, c0, h0, W0, b0)
        c1, h1 = O._lstm_cell(h0, c1, h1, W1, b1)
        q = h1 @ Wa + ba
        e = np.sum(np.tanh(q[None] + eht) * v, axis=2, keepdims=True)
        att = O._softmax(e, axis=0) * nf
        att = att / np.sum(att, axis=0, keepdims=True)
        ctx = np.sum(att * eout, axis=0)
        feats[t] = np.concatenate([h1, ctx], axis=1)
        sc = feats[t] @ Wy + by
        tok = np.argmax(sc, axis=1).astype(np.int32)                     # :193
        if use_gt_layout:
            tok = np.asarray(gt_layout[t], np.int32)                     # :194-196
        p = O._softmax(sc, axis=1)
        tp = p[np.arange(N), tok]
        ne = np.sum(p * np.log(np.maximum(dtype(1e-5), p)), axis=1)      # :207-208
        tok_old = tok
        tok = np.where(is_eos, eos_idx, tok).astype(np.int32)            # :215-220
        tp = np.where(is_eos, dtype(1.0), tp)
        ne = np.where(is_eos, dtype(0.0), ne)
        is_eos = is_eos | (tok_old == eos_idx)                           # :221-222
        neg_ent += ne
        x = demb[tok]
        tokens[t] = tok; tprobs[t] = tp; atts[t] = att; scores_all[t] = sc
    word_vecs = np.sum(atts * enc['embedded'][None], axis=1)
    return dict(predicted_tokens=tokens, token_probs=tprobs, neg_entropy=neg_ent, atts=atts,
                word_vecs=word_vecs, token_scores=scores_all, token_features=feats)


def assemble(tokens):
    """models_shapes/nmn3_assembler.py:44-120 (same stack decoding, SHAPES tables)."""
    names = list(SHAPES_MODULE_NAMES)
    saved = (dict(O.ARITY), dict(O.OUT_TYPE))
    try:
        O.ARITY.update(ARITY); O.OUT_TYPE.update(OUT_TYPE)
        return O.assemble(names, tokens)
    finally:
        O.ARITY.clear(); O.ARITY.update(saved[0]); O.OUT_TYPE.clear(); O.OUT_TYPE.update(saved[1])


def _mw(w):
    """module variables under the names oracle/n2nmn_oracle.py's operators expect"""
    out = {}
    for k, v in w.items():
        if k.startswith(_MOD):
            scope, rest = k[len(_MOD):].split('/', 1)          # <X>Module/<X>Module/... -> <X>Module/...
            out[O._MOD + rest] = v
    return out


def eval_expr(mw, expr, feat, word_vecs, num_choices, dtype):
    if expr['module'] == O.INVALID:
        return np.zeros(num_choices, dtype)

    def rec(e):
        t, n = e['time_idx'], e['batch_idx']
        f = feat[n:n + 1]
        txt = word_vecs[t, n][None]                       # gather_nd([t, b]), nmn3_modules.py:20-25
        ins = [rec(e[k]) for k in ('input_0', 'input_1') if k in e]
        m = e['module']
        if m == '_Find': return O.m_find(mw, f, txt)
        if m == '_Transform': return O.m_transform(mw, ins[0], txt)
        if m == '_And': return O.m_and(ins[0], ins[1])
        if m == '_Answer':                                # :122-144 = fc([min, mean, max])
            g = ins[0].reshape(1, -1)
            red = np.stack([g.min(1), g.mean(1, dtype=g.dtype), g.max(1)], axis=1)
            return red @ mw[O._MOD + 'AnswerModule/fc_scores/weights'] + mw[O._MOD + 'AnswerModule/fc_scores/biases']
        raise KeyError(m)

    return rec(expr)[0]


def forward(w, batch, T_dec=DIMS['T_decoder'], num_choices=DIMS['num_choices'], dtype=np.float64,
            use_gt_layout=False, gt_layout=None):
    """exp_shapes/eval_shapes.py:150-180 for one batch: images -> convnet; seq2seq; assemble; modules."""
    wc = O._cast(w, dtype)
    feat = shapes_convnet(wc, np.asarray(batch['image_batch'], dtype))
    enc = O.encoder_forward(w, batch['text_seq_batch'], batch['seq_length_batch'], dtype)
    eos = list(SHAPES_MODULE_NAMES).index('<eos>')
    dec = decoder_forward(w, enc, T_dec, eos, dtype, use_gt_layout, gt_layout)
    exprs, validity = assemble(dec['predicted_tokens'])
    mw = _mw(wc)
    scores = np.stack([eval_expr(mw, e, feat, dec['word_vecs'], num_choices, dtype) for e in exprs])
    return dict(feat=feat, enc=enc, dec=dec, expr_list=exprs, validity=validity, scores=scores)


def load_split(root, image_set, T_encoder=DIMS['T_encoder'], T_decoder=DIMS['T_decoder']):
    """The data plumbing of exp_shapes/eval_shapes.py:60-114 for one split under `root`
    (= the reference checkout).  Returns numpy arrays in the shuffled (seed 3) order."""
    with open(os.path.join(root, 'exp_shapes/data/vocabulary_shape.txt')) as f:
        vocab = [s.strip() for s in f.readlines()]
    vdict = {s: i for i, s in enumerate(vocab)}
    with open(os.path.join(root, 'exp_shapes/data/vocabulary_layout.txt')) as f:
        layout_vocab = [s.strip() for s in f.readlines()]
    ldict = {s: i for i, s in enumerate(layout_vocab)}
    ds = os.path.join(root, 'exp_shapes/shapes_dataset')
    with open(os.path.join(ds, '%s.query_str.txt' % image_set)) as f:
        questions = [l.strip() for l in f.readlines()]
    with open(os.path.join(ds, '%s.output' % image_set)) as f:
        labels = [l.strip() == 'true' for l in f.readlines()]
    images = np.load(os.path.join(ds, '%s.input.npy' % image_set))
    with open(os.path.join(root, 'exp_shapes/data/%s.query_layout_symbols.json' % image_set)) as f:
        layouts = json.load(f)
    n = len(questions)
    rs = np.random.RandomState(3)                          # np.random.seed(3); permutation  (:89-90)
    order = rs.permutation(n)
    questions = [questions[i] for i in order]
    labels = [labels[i] for i in order]
    images = images[order]
    layouts = [layouts[i] for i in order]
    text = np.zeros((T_encoder, n), np.int32)
    lens = np.zeros(n, np.int32)
    gt = np.zeros((T_decoder, n), np.int32)
    eos = ldict['<eos>']
    for q in range(n):
        toks = questions[q].split()
        lens[q] = len(toks)
        for t, tk in enumerate(toks):
            text[t, q] = vdict[tk]
        if len(layouts[q]) >= T_decoder:
            raise ValueError('Not enough time steps to add <eos>')
        gt[:, q] = [ldict[m] for m in layouts[q]] + [eos] * (T_decoder - len(layouts[q]))
    mean = np.load(os.path.join(root, 'exp_shapes/data/image_mean.npy'))
    return dict(vocab=vocab, layout_vocab=layout_vocab, text_seq=text, seq_length=lens, gt_layout=gt,
                images_u8=images, image_mean=mean, labels=np.array(labels, np.int32), order=order)
