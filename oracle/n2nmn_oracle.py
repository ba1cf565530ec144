"""CPU ORACLE for the N2NMN CLEVR forward path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (n2nmn_amd/) never imports it and has no CPU fallback.

What it is: a plain-numpy restatement of the reference's algorithm, written from the reference's
semantics (file:line cited on every function, relative to ronghanghu/n2nmn) plus the TensorFlow
1.0.0 / TensorFlow-Fold 0.0.1 op semantics listed in SURVEY.md Appendix A.  It runs in float64
(the ground truth the tolerances are set against) or float32 (`dtype=`).

PARITY PINNING STATUS
  * integer / host logic (validity matrices P/W/b, RPN assembler): PINNED -- checked against
    golden vectors produced by importing the reference's own numpy code
    (tests/golden/make_assembler_golden.py -> tests/golden/assembler_golden.json).
  * floating-point path (LSTM encoder/decoder, module operators, losses): PINNED since round 2 to
    the reference's OWN model code.  TF 1.0.0 / TF-Fold 0.0.1 cannot be installed, so
    tests/golden/make_float_golden.py imports the unmodified models_clevr/*.py, models_vqa/*.py and
    util/cnn.py under oracle/tf1_stub/ (eager float64 stand-ins for the ~60 TF ops and the Fold
    blocks they call), runs them on the seeded inputs of tests/golden/float_cases.py and commits the
    outputs as tests/golden/float_golden.npz; tests/test_oracle_vs_reference_code.py holds this
    module to that fixture at 1e-10.  What is still restated rather than executed is TensorFlow
    LIBRARY code (BasicLSTMCell, dynamic_rnn, raw_rnn, Adam, the arithmetic of each primitive op):
    DESIGN.md section 6 lists it.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------
# module tables -- models_clevr/nmn3_assembler.py:9-41
# ----------------------------------------------------------------------------------------
ARITY = {'_Scene': 0, '_Find': 0, '_Filter': 1, '_FindSameProperty': 1, '_Transform': 1,
         '_And': 2, '_Or': 2, '_Count': 1, '_Exist': 1, '_EqualNum': 2, '_MoreNum': 2,
         '_LessNum': 2, '_SameProperty': 2, '_Describe': 1}
OUT_TYPE = {'_Scene': 'att', '_Find': 'att', '_Filter': 'att', '_FindSameProperty': 'att',
            '_Transform': 'att', '_And': 'att', '_Or': 'att', '_Count': 'ans', '_Exist': 'ans',
            '_EqualNum': 'ans', '_MoreNum': 'ans', '_LessNum': 'ans', '_SameProperty': 'ans',
            '_Describe': 'ans'}
INVALID = 'INVALID_EXPR'

_P = 'neural_module_network/'
_ENC = _P + 'layout_generation/encoder_decoder/encoder/'
_DEC = _P + 'layout_generation/encoder_decoder/decoder/'
_MOD = _P + 'layout_execution/module_variables/'


# ----------------------------------------------------------------------------------------
# validity automaton matrices -- models_clevr/nmn3_assembler.py:50-119
# ----------------------------------------------------------------------------------------
def build_validity_mats(module_names):
    """State x = (#att on stack, #ans on stack, T_remain).  Token s is allowed iff
    all_c(x . W[:, s, c] - b[s, c] >= 0); after emitting s the state moves by P[s].
    Follows nmn3_assembler.py:50-119 constraint by constraint."""
    V = len(module_names)
    P = np.zeros((V, 3), np.int32)
    W = np.zeros((3, V, 4), np.int32)
    b = np.zeros((V, 4), np.int32)
    a_in = np.array([ARITY.get(s, 0) for s in module_names])
    a_out = np.array([int(OUT_TYPE.get(s) == 'att') for s in module_names])
    r_out = np.array([int(OUT_TYPE.get(s) == 'ans') for s in module_names])
    absorb = a_in - a_out                                           # :77
    max_absorb_nonans = int(np.max(absorb * (r_out == 0)))          # :78
    max_absorb_ans = int(np.max(absorb * (r_out != 0)))             # :79
    for s, name in enumerate(module_names):
        P[s] = (a_out[s] - a_in[s], r_out[s], -1)                   # :72-75
        if name == '<eos>':
            W[1, s, 0] = 1                                          # #ans >= 1      (:112-116)
            b[s, 0] = 1
            continue
        W[0, s, 0] = 1                                              # #att >= arity  (:85-86)
        b[s, 0] = a_in[s]
        if r_out[s]:                                                # answer: #att <= arity (:92-94)
            W[0, s, 1] = -1
            b[s, 1] = -a_in[s]
        else:                                                       # non-answer: T_remain >= 3 (:95-97)
            W[2, s, 1] = 1
            b[s, 1] = 3
        W[1, s, 2] = -1                                             # #ans <= 0      (:101)
        if not r_out[s]:                                            # enough time left (:102-114)
            W[0, s, 3] = -1
            W[2, s, 3] = max_absorb_nonans
            b[s, 3] = 3 * max_absorb_nonans - max_absorb_ans - absorb[s]
    return P, W, b


def valid_tokens(X, W, b):
    """nmn3_netgen_att.py:8-11: all(X.W - b >= 0, axis=constraints).  X:[N,3] int."""
    return np.all(np.tensordot(X, W, axes=1) - b >= 0, axis=2)


# ----------------------------------------------------------------------------------------
# RPN assembler -- models_clevr/nmn3_assembler.py:145-222
# ----------------------------------------------------------------------------------------
def _invalid(names, toks, msg):
    return {'module': INVALID, 'expr_str': ' '.join(names[int(i)] for i in toks), 'error': msg}


def assemble_one(names, toks, batch_idx):
    """nmn3_assembler.py:153-212.  Never raises; violations come back as INVALID_EXPR dicts."""
    eos = list(names).index('<eos>')
    toks = np.asarray(toks)
    if not np.any(toks == eos):                                      # :172-173
        return _invalid(names, toks, 'cannot find <eos>')
    stack = []
    for t, tok in enumerate(toks):                                   # :177
        if tok == eos:
            break
        name = names[int(tok)]
        node = {'module': name, 'output_type': OUT_TYPE[name], 'time_idx': t,
                'batch_idx': batch_idx}                              # :183-185
        k = ARITY[name]
        if len(stack) < k:                                           # :189-191
            return _invalid(names, toks, 'not enough input for ' + name)
        for j in reversed(range(k)):                                 # :194-199: input_{k-1} = top
            top = stack.pop()
            if top['output_type'] != 'att':
                return _invalid(names, toks, 'input incompatible for ' + name)
            node['input_%d' % j] = top
        stack.append(node)
    if len(stack) != 1:                                              # :205-206
        return _invalid(names, toks,
                        'final stack size not equal to 1 (%d remains)' % len(stack))
    if stack[0]['output_type'] != 'ans':                             # :209-211
        return _invalid(names, toks, 'result type must be ans, not att')
    return stack[0]


def assemble(names, tokens):
    """nmn3_assembler.py:214-222.  tokens [T, N] -> (expr_list, validity[N] bool)."""
    tokens = np.asarray(tokens)
    exprs = [assemble_one(names, tokens[:, n], n) for n in range(tokens.shape[1])]
    return exprs, np.array([e['module'] != INVALID for e in exprs], bool)


# ----------------------------------------------------------------------------------------
# small numeric helpers (TF semantics)
# ----------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softmax(x, axis):
    x = x - np.max(x, axis=axis, keepdims=True)
    e = np.exp(x)
    return e / np.sum(e, axis=axis, keepdims=True)


def _l2n(x, axis):
    # tf.nn.l2_normalize(x, dim, epsilon=1e-12) = x * rsqrt(max(sum(x^2), eps))
    ss = np.sum(x * x, axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(ss, x.dtype.type(1e-12)))


def _cast(w, dtype):
    return {k: np.asarray(v, dtype) for k, v in w.items()}


def _lstm_cell(x, c, h, Wm, bm):
    """tf.contrib.rnn.BasicLSTMCell (TF 1.0.0), used at nmn3_netgen_att.py:23,33 (Appendix A.1):
    z = [x, h].W + b; i, j, f, o = split(z, 4); c' = c*sig(f + 1) + sig(i)*tanh(j);
    h' = tanh(c')*sig(o)."""
    z = np.concatenate([x, h], axis=1) @ Wm + bm
    i, j, f, o = np.split(z, 4, axis=1)
    c2 = c * _sigmoid(f + z.dtype.type(1.0)) + _sigmoid(i) * np.tanh(j)
    h2 = np.tanh(c2) * _sigmoid(o)
    return c2, h2


def _lstm_w(w, which, layer):
    base = (_ENC if which == 'encoder' else _DEC) + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/' % layer
    return w[base + 'weights'], w[base + 'biases']


# ----------------------------------------------------------------------------------------
# encoder -- models_clevr/nmn3_netgen_att.py:73-113
# ----------------------------------------------------------------------------------------
def encoder_forward(w, input_seq, seq_len, dtype=np.float64, drop0=None):
    """drop0 [T, N, L] of {0, 1} keep masks (encoder_dropout=True, models_vqa/nmn3_netgen_att.py:17-44: the
    DropoutWrapper(output_keep_prob=0.5) around layer 0 scales what FEEDS layer 1 by mask / 0.5; the recurrent state
    is not touched) -- the same statement as oracle/n2nmn_oracle_grad.py:encoder_forward"""
    w = _cast(w, dtype)
    T, N = input_seq.shape
    emb = w[_ENC + 'embedding_mat']
    E = emb[input_seq]                                   # :87-88  [T, N, E]
    L = w[_ENC + 'encoder_h_transform/weights'].shape[0]
    W0, b0 = _lstm_w(w, 'encoder', 0)
    W1, b1 = _lstm_w(w, 'encoder', 1)
    c0 = np.zeros((N, L), dtype); h0 = np.zeros((N, L), dtype)
    c1 = np.zeros((N, L), dtype); h1 = np.zeros((N, L), dtype)
    outs = np.zeros((T, N, L), dtype)
    for t in range(T):                                   # dynamic_rnn(sequence_length) A.2
        act = (t < seq_len)[:, None]
        nc0, nh0 = _lstm_cell(E[t], c0, h0, W0, b0)
        nc1, nh1 = _lstm_cell(nh0 if drop0 is None else nh0 * (np.asarray(drop0[t], dtype) * dtype(2.0)), c1, h1, W1, b1)
        outs[t] = np.where(act, nh1, 0)                  # zero output past the length
        c0 = np.where(act, nc0, c0); h0 = np.where(act, nh0, h0)   # state carried through
        c1 = np.where(act, nc1, c1); h1 = np.where(act, nh1, h1)
    eht = outs.reshape(T * N, L) @ w[_ENC + 'encoder_h_transform/weights'] \
        + w[_ENC + 'encoder_h_transform/biases']         # :102-106
    eht = eht.reshape(T, N, L)
    not_fin = (np.arange(T)[:, None] < seq_len[None, :]).astype(dtype)[:, :, None]  # :110-113
    return dict(embedded=E, outputs=outs, h_transformed=eht, not_finished=not_fin,
                states=((c0, h0), (c1, h1)))


# ----------------------------------------------------------------------------------------
# decoder -- models_clevr/nmn3_netgen_att.py:115-322
# ----------------------------------------------------------------------------------------
def decoder_forward(w, enc, P, Wv, bv, T_dec, dtype=np.float64, use_gt_layout=False,
                    gt_layout=None, sample_uniforms=None, forced_tokens=None, drop0=None):
    """Greedy (default), teacher-forced (use_gt_layout + gt_layout[T_dec,N]) or sampled decoding.

    sample_uniforms[T_dec, N] in [0,1): replaces tf.multinomial (:216-217), whose RNG stream is
    not reproducible, by inverse-CDF sampling from softmax(scores - 50*(1-valid)); the fall-back
    to the greedy token when the sample is invalid (:219-232) is kept.
    forced_tokens[T_dec, N]: parity-protocol hook -- overrides the chosen token AFTER
    validity/probabilities are computed with the normal rules (unlike use_gt_layout it does not
    change the validity mask).
    drop0 [T_dec, N, L] of {0, 1}: decoder_dropout=True, as in encoder_forward."""
    w = _cast(w, dtype)
    (c0, h0), (c1, h1) = enc['states']                   # :177 initial state = encoder state
    N, L = h0.shape
    demb = w[_DEC + 'embedding_mat']
    V = demb.shape[0]
    x = np.tile(w[_DEC + 'go_embedding'], (N, 1))        # :178
    v = w[_DEC + 'att_prediction/v']
    Wa, ba = w[_DEC + 'att_prediction/weights'], w[_DEC + 'att_prediction/biases']
    Wy, by = w[_DEC + 'token_prediction/weights'], w[_DEC + 'token_prediction/biases']
    W0, b0 = _lstm_w(w, 'decoder', 0)
    W1, b1 = _lstm_w(w, 'decoder', 1)
    eht, eout, nf = enc['h_transformed'], enc['outputs'], enc['not_finished']
    T_enc = eht.shape[0]
    X = np.tile(np.array([[0, 0, T_dec]], np.int64), (N, 1))        # :284
    tokens = np.zeros((T_dec, N), np.int32)
    tprobs = np.zeros((T_dec, N), dtype)
    atts = np.zeros((T_dec, T_enc, N, 1), dtype)
    scores_all = np.zeros((T_dec, N, V), dtype)
    valid_all = np.zeros((T_dec, N, V), bool)
    neg_ent = np.zeros(N, dtype)
    one = dtype(1.0)
    for t in range(T_dec):                               # raw_rnn: exactly T_dec cell calls (A.3)
        c0, h0 = _lstm_cell(x, c0, h0, W0, b0)
        c1, h1 = _lstm_cell(h0 if drop0 is None else h0 * (np.asarray(drop0[t], dtype) * dtype(2.0)), c1, h1, W1, b1)
        out = h1
        q = out @ Wa + ba                                # :184-187
        e = np.sum(np.tanh(q[None] + eht) * v, axis=2, keepdims=True)       # [T_enc, N, 1]
        att = _softmax(e, axis=0) * nf                   # :190 softmax over ALL rows, then mask
        att = att / np.sum(att, axis=0, keepdims=True)   # :191
        ctx = np.sum(att * eout, axis=0)                 # :193
        sc = np.concatenate([out, ctx], axis=1) @ Wy + by            # :196-198
        valid = valid_tokens(X, Wv, bv)                  # :200-203
        if use_gt_layout:
            valid = np.ones_like(valid)                  # :204-207 logical_or(valid, True)
        vm = valid.astype(dtype)
        masked = np.where(valid, sc, np.min(sc) - one)   # :234-236 (global min over [N, V])
        greedy = np.argmax(masked, axis=1).astype(np.int32)          # :238 first max index
        if sample_uniforms is not None:
            p_s = _softmax(sc - (one - vm) * dtype(50.0), axis=1)    # :213
            cdf = np.cumsum(p_s, axis=1)
            u = np.asarray(sample_uniforms[t], dtype)[:, None]
            samp = np.minimum(np.sum(cdf <= u * cdf[:, -1:], axis=1), V - 1).astype(np.int32)
            ok = valid[np.arange(N), samp]               # :222-225
            tok = np.where(ok, samp, greedy).astype(np.int32)        # :232
        else:
            tok = greedy
        if use_gt_layout:
            tok = np.asarray(gt_layout[t], np.int32)     # :239-241
        if forced_tokens is not None:
            tok = np.asarray(forced_tokens[t], np.int32)
        p = _softmax(sc, axis=1) * vm                    # :245
        p = p / np.sum(p, axis=1, keepdims=True)         # :247
        tprobs[t] = p[np.arange(N), tok]                 # :251-256
        neg_ent += np.sum(p * np.log(np.maximum(dtype(1e-5), p + (one - vm))), axis=1)  # :258-260
        X = X + P[tok]                                   # :13-15,263-264
        x = demb[tok]                                    # :268
        tokens[t] = tok; atts[t] = att; scores_all[t] = sc; valid_all[t] = valid
    word_vecs = np.sum(atts * enc['embedded'][None], axis=1)        # :312  [T_dec, N, E]
    return dict(predicted_tokens=tokens, token_probs=tprobs, neg_entropy=neg_ent, atts=atts,
                word_vecs=word_vecs, token_scores=scores_all, token_validity=valid_all,
                final_states=((c0, h0), (c1, h1)))


# ----------------------------------------------------------------------------------------
# module operators -- models_clevr/nmn3_modules.py
# All take gathered inputs:  feat [Nb,H,W,D], txt [Nb,E], att maps [Nb,H,W,1].
# ----------------------------------------------------------------------------------------
def _fc(w, scope, x):                                    # util/cnn.py:87-119 (xw_plus_b)
    return x @ w[_MOD + scope + '/weights'] + w[_MOD + scope + '/biases']


def _conv1x1(w, scope, x):                               # util/empty_safe_conv.py:8-32
    shp = x.shape
    y = x.reshape(-1, shp[-1]) @ w[_MOD + scope + '/weights'] + w[_MOD + scope + '/biases']
    return y.reshape(shp[:-1] + (y.shape[-1],))


def _conv_same(w, scope, x):
    """tf.nn.conv2d NHWC, stride 1, SAME, cross-correlation (util/cnn.py:29-32; A.4)."""
    K = w[_MOD + scope + '/weights']                     # [kh, kw, cin, cout]
    kh, kw, cin, cout = K.shape
    N, H, Wd, _ = x.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((N, H + 2 * ph, Wd + 2 * pw, cin), x.dtype)
    xp[:, ph:ph + H, pw:pw + Wd] = x
    y = np.zeros((N, H, Wd, cout), x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            y += xp[:, dy:dy + H, dx:dx + Wd, :] @ K[dy, dx]
    return y + w[_MOD + scope + '/biases']


def _att_pool(feat, att):
    """softmax over H*W of the attention logits, then attention-weighted feature sum
    (nmn3_modules.py:170-174, 432-440, 482-487)."""
    N, H, Wd, _ = att.shape
    a = _softmax(att.reshape(N, H * Wd), axis=1).reshape(N, H, Wd, 1)
    return np.sum(feat * a, axis=(1, 2))


def m_scene(w, n, H, Wd, dtype):                         # :60-72
    return np.full((n, H, Wd, 1), 3.0, dtype)


def m_find(w, feat, txt):                                # :74-111
    img = _conv1x1(w, 'FindModule/conv_image', feat)
    t = _fc(w, 'FindModule/fc_text', txt)[:, None, None, :]
    return _conv1x1(w, 'FindModule/conv_eltwise', _l2n(img * t, 3))


def m_filter(w, in0, feat, txt):                         # :113-132 (Find weights, then And)
    return np.minimum(in0, m_find(w, feat, txt))


def m_find_same_property(w, in0, feat, txt):             # :134-183
    s = 'FindSamePropertyModule/'
    img = _conv1x1(w, s + 'conv_image', feat)
    t = _fc(w, s + 'fc_text', txt)[:, None, None, :]
    a = _fc(w, s + 'fc_att', _att_pool(feat, in0))[:, None, None, :]
    return _conv1x1(w, s + 'conv_eltwise', _l2n(img * t * a, 3))


def m_transform(w, in0, txt):                            # :185-216
    s = 'TransformModule/'
    maps = _conv_same(w, s + 'conv_maps', in0)
    t = _fc(w, s + 'text_fc', txt)[:, None, None, :]
    return _conv1x1(w, s + 'conv_eltwise', _l2n(maps * t, 3))


def m_and(in0, in1):                                     # :218-236
    return np.minimum(in0, in1)


def m_or(in0, in1):                                      # :238-256
    return np.maximum(in0, in1)


def m_exist(w, in0):                                     # :258-280
    f = in0.reshape(in0.shape[0], -1)
    red = np.stack([f.min(1), f.mean(1, dtype=f.dtype), f.max(1)], axis=1)
    return _fc(w, 'ExistModule/fc_scores', red)


def m_count(w, in0):                                     # :282-304 (row-major y*W + x flatten)
    f = in0.reshape(in0.shape[0], -1)
    cat = np.concatenate([f, f.min(1, keepdims=True), f.max(1, keepdims=True)], axis=1)
    return _fc(w, 'CountModule/fc_scores', cat)


def _m_compare(w, scope, in0, in1):                      # :306-400
    f0 = in0.reshape(in0.shape[0], -1)
    f1 = in1.reshape(in1.shape[0], -1)
    cat = np.concatenate([f0, f0.min(1, keepdims=True), f0.max(1, keepdims=True),
                          f1, f1.min(1, keepdims=True), f1.max(1, keepdims=True)], axis=1)
    return _fc(w, scope + '/fc_scores', cat)


def m_equal_num(w, in0, in1):
    return _m_compare(w, 'EqualNumModule', in0, in1)


def m_more_num(w, in0, in1):
    return _m_compare(w, 'MoreNumModule', in0, in1)


def m_less_num(w, in0, in1):
    return _m_compare(w, 'LessNumModule', in0, in1)


def m_same_property(w, in0, in1, feat, txt):             # :402-452
    s = 'SamePropertyModule/'
    t = _fc(w, s + 'fc_text', txt)
    a0 = _fc(w, s + 'fc_att_0', _att_pool(feat, in0))
    a1 = _fc(w, s + 'fc_att_1', _att_pool(feat, in1))
    return _fc(w, s + 'fc_eltwise', _l2n(a0 * t * a1, 1))


def m_describe(w, in0, feat, txt):                       # :454-495
    s = 'DescribeModule/'
    t = _fc(w, s + 'fc_text', txt)
    a = _fc(w, s + 'fc_att', _att_pool(feat, in0))
    return _fc(w, s + 'fc_eltwise', _l2n(t * a, 1))


# ----------------------------------------------------------------------------------------
# tree interpreter (replaces TF-Fold; pattern of exp_shapes/visualize_shapes.ipynb code cell 9,
# `eval_module` / `eval_expr`; Fold semantics: SURVEY Appendix A.5)
# ----------------------------------------------------------------------------------------
def eval_expr(w, expr, image_feat, word_vecs, num_choices, dtype):
    """Evaluate ONE example's expression tree; invalid -> zeros(num_choices)
    (nmn3_model.py:146,155)."""
    if expr['module'] == INVALID:
        return np.zeros(num_choices, dtype)
    N_full = word_vecs.shape[1]
    flat = word_vecs.reshape(-1, word_vecs.shape[-1])     # nmn3_modules.py:19-24

    def rec(e):
        t, n = e['time_idx'], e['batch_idx']
        feat = image_feat[n:n + 1]                        # :49-51
        txt = flat[t * N_full + n][None]                  # :53-57
        ins = [rec(e[k]) for k in ('input_0', 'input_1') if k in e]
        m = e['module']
        H, Wd = image_feat.shape[1:3]
        if m == '_Scene': return m_scene(w, 1, H, Wd, dtype)
        if m == '_Find': return m_find(w, feat, txt)
        if m == '_Filter': return m_filter(w, ins[0], feat, txt)
        if m == '_FindSameProperty': return m_find_same_property(w, ins[0], feat, txt)
        if m == '_Transform': return m_transform(w, ins[0], txt)
        if m == '_And': return m_and(ins[0], ins[1])
        if m == '_Or': return m_or(ins[0], ins[1])
        if m == '_Exist': return m_exist(w, ins[0])
        if m == '_Count': return m_count(w, ins[0])
        if m == '_EqualNum': return m_equal_num(w, ins[0], ins[1])
        if m == '_MoreNum': return m_more_num(w, ins[0], ins[1])
        if m == '_LessNum': return m_less_num(w, ins[0], ins[1])
        if m == '_SameProperty': return m_same_property(w, ins[0], ins[1], feat, txt)
        if m == '_Describe': return m_describe(w, ins[0], feat, txt)
        raise KeyError(m)

    return rec(expr)[0]


def execute_layouts(w, expr_list, image_feat, word_vecs, num_choices, dtype=np.float64):
    """scores [len(expr_list), num_choices]; row i <-> expr_list[i] (A.5)."""
    w = _cast(w, dtype)
    image_feat = np.asarray(image_feat, dtype)
    word_vecs = np.asarray(word_vecs, dtype)
    return np.stack([eval_expr(w, e, image_feat, word_vecs, num_choices, dtype)
                     for e in expr_list])


# ----------------------------------------------------------------------------------------
# whole forward -- exp_clevr/eval_clevr.py:103-135 + models_clevr/nmn3_model.py:15-166
# ----------------------------------------------------------------------------------------
def forward(w, module_names, batch, T_dec, num_choices, dtype=np.float64, use_gt_layout=False,
            gt_layout=None, sample_uniforms=None, forced_tokens=None):
    P, Wv, bv = build_validity_mats(module_names)
    enc = encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], dtype)
    dec = decoder_forward(w, enc, P, Wv, bv, T_dec, dtype, use_gt_layout, gt_layout,
                          sample_uniforms, forced_tokens)
    exprs, validity = assemble(module_names, dec['predicted_tokens'])
    scores = execute_layouts(w, exprs, batch['image_feat_batch'], dec['word_vecs'],
                             num_choices, dtype)
    log_seq_prob = np.sum(np.log(dec['token_probs']), axis=0)       # nmn3_model.py:46
    return dict(enc=enc, dec=dec, expr_list=exprs, validity=validity, scores=scores,
                log_seq_prob=log_seq_prob)


# ----------------------------------------------------------------------------------------
# losses -- exp_clevr/train_clevr_gt_layout.py:104-124 (config 4; forward values only)
# ----------------------------------------------------------------------------------------
def losses(w, scores, labels, log_seq_prob, weight_decay=5e-6):
    z = scores - scores.max(1, keepdims=True)
    ce = np.log(np.exp(z).sum(1)) - z[np.arange(len(labels)), labels]
    l2 = sum(0.5 * np.sum(np.asarray(v, np.float64) ** 2) for k, v in w.items()
             if k.endswith('weights'))                   # nmn3_model.py:163-166
    avg_sample_loss = ce.mean()
    seq_likelihood_loss = np.mean(-log_seq_prob)
    return dict(avg_sample_loss=avg_sample_loss, seq_likelihood_loss=seq_likelihood_loss,
                l2_reg=l2, total_loss=seq_likelihood_loss + avg_sample_loss + weight_decay * l2)


# ========================================================================================
# models_vqa variant (BASELINE.json configs[4]; exp_vqa/eval_vqa2.py:27-39,103-137)
# Same AttentionSeq2Seq (models_vqa/nmn3_netgen_att.py is byte-identical to models_clevr's);
# four modules over the feature grid with two coordinate channels appended; question prior net.
# ========================================================================================
VQA_MODULE_NAMES = ('_Find', '_Transform', '_And', '_Describe', '<eos>')
_QPN = _P + 'question_prior_net/'


def add_spatial_coordinate_map(feat):
    """models_vqa/nmn3_modules.py:11-31: concat [x_map, y_map], x = linspace(-1,1,W) along W,
    y = linspace(-1,1,H) along H (TF computes the linspace in float32)."""
    N, H, W, _ = feat.shape
    x = np.linspace(-1.0, 1.0, W, dtype=np.float32).astype(feat.dtype)
    y = np.linspace(-1.0, 1.0, H, dtype=np.float32).astype(feat.dtype)
    xm = np.broadcast_to(x[None, None, :, None], (N, H, W, 1))
    ym = np.broadcast_to(y[None, :, None, None], (N, H, W, 1))
    return np.concatenate([feat, xm, ym], axis=3)


def vqa_find(w, feat, txt):                              # models_vqa/nmn3_modules.py:82-121
    return m_find(w, feat, txt)


def vqa_transform(w, in0, feat, txt):                    # :123-171 ("Same as FindSamePropertyModule")
    s = 'TransformModule/'
    img = _conv1x1(w, s + 'conv_image', feat)
    t = _fc(w, s + 'fc_text', txt)[:, None, None, :]
    a = _fc(w, s + 'fc_att', _att_pool(feat, in0))[:, None, None, :]
    return _conv1x1(w, s + 'conv_eltwise', _l2n(img * t * a, 3))


def vqa_describe(w, in0, feat, txt):                     # :193-240 (encoder_states is None)
    return m_describe(w, in0, feat, txt)


def question_prior_net(w, enc_states):
    """models_vqa/question_prior_net.py:10-28 without dropout: fc2(relu(fc1(concat h)))."""
    h = np.concatenate([s[1] for s in enc_states], axis=1)
    fc1 = np.maximum(h @ w[_QPN + 'fc1/weights'] + w[_QPN + 'fc1/biases'], 0)
    return fc1 @ w[_QPN + 'fc2/weights'] + w[_QPN + 'fc2/biases']


def eval_expr_vqa(w, expr, feat_c, word_vecs, num_choices, dtype):
    if expr['module'] == INVALID:
        return np.zeros(num_choices, dtype)
    N_full = word_vecs.shape[1]
    flat = word_vecs.reshape(-1, word_vecs.shape[-1])

    def rec(e):
        t, n = e['time_idx'], e['batch_idx']
        feat = feat_c[n:n + 1]
        txt = flat[t * N_full + n][None]
        ins = [rec(e[k]) for k in ('input_0', 'input_1') if k in e]
        m = e['module']
        if m == '_Find': return vqa_find(w, feat, txt)
        if m == '_Transform': return vqa_transform(w, ins[0], feat, txt)
        if m == '_And': return m_and(ins[0], ins[1])
        if m == '_Describe': return vqa_describe(w, ins[0], feat, txt)
        raise KeyError(m)

    return rec(expr)[0]


def forward_vqa(w, batch, T_dec, num_choices, dtype=np.float64, use_qpn=True, use_gt_layout=False,
                gt_layout=None, forced_tokens=None):
    """exp_vqa/eval_vqa2.py:103-135 + models_vqa/nmn3_model.py:15-121 (scores = scores_nmn +
    scores_qpn; the eval script's `scores_val[:, 0] = -1e10` is host post-processing, not here)."""
    names = list(VQA_MODULE_NAMES)
    P, Wv, bv = build_validity_mats(names)
    enc = encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], dtype)
    dec = decoder_forward(w, enc, P, Wv, bv, T_dec, dtype, use_gt_layout, gt_layout, None,
                          forced_tokens)
    exprs, validity = assemble(names, dec['predicted_tokens'])
    wc = _cast(w, dtype)
    feat_c = add_spatial_coordinate_map(np.asarray(batch['image_feat_batch'], dtype))
    wv = np.asarray(dec['word_vecs'], dtype)
    scores_nmn = np.stack([eval_expr_vqa(wc, e, feat_c, wv, num_choices, dtype) for e in exprs])
    scores = scores_nmn + question_prior_net(wc, enc['states']) if use_qpn else scores_nmn
    return dict(enc=enc, dec=dec, expr_list=exprs, validity=validity, scores_nmn=scores_nmn,
                scores=scores)
