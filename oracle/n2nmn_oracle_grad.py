"""CPU ORACLE for the N2NMN CLEVR *training step* (config 4) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it is: the forward of oracle/n2nmn_oracle.py restated on torch-CPU float64 tensors so that
torch.autograd yields d(total_loss)/d(variable) for every trainable variable of the reference's
behavioural-cloning objective (exp_clevr/train_clevr_gt_layout.py:104-124):

    total_loss = mean_n(-log_seq_prob[n]) + mean_n(softmax_CE(scores[n], label[n]))
                 + weight_decay * sum_{v name ends with 'weights'} 0.5*||v||^2
                                                    (models_clevr/nmn3_model.py:161-166)

with teacher-forced decoding (use_gt_layout=True: every token valid, token_prob =
softmax(token_scores)[gt], models_clevr/nmn3_netgen_att.py:204-207,239-256), followed by the
reference's optimiser: per-tensor tf.clip_by_norm(g, 10) and tf.train.AdamOptimizer() with its
TF 1.0.0 defaults (train_clevr_gt_layout.py:112-120).

The forward VALUES of this module are checked against the numpy oracle in
tests/test_oracle_grad.py (same weights/inputs, fp64 round-off); the gradients are checked there
against central finite differences of the numpy oracle's loss.  PARITY STATUS: PINNED since round 2
-- every variable's gradient of both objectives, the per-tensor clip, one Adam step and the EMA
baseline are held (1e-9 relative) to what the reference's own model files and the loss blocks of
exp_clevr/train_clevr_gt_layout.py / train_clevr_rl_gt_layout.py compute under the eager TF1/Fold
stand-in (tests/golden/make_float_golden.py -> float_golden.npz, tests/test_oracle_vs_reference_code.py).

TF gradient conventions restated here (TF 1.0.0 math_grad.py):
  * tf.minimum / tf.maximum (And / Or / Filter): ties send the gradient to the FIRST argument
    (xmask = x <= y for minimum, x >= y for maximum).
  * tf.reduce_min / reduce_max (Exist / Count / *Num): the gradient is split equally between all
    positions equal to the extremum.
  * tf.nn.l2_normalize(x, dim, eps): x * rsqrt(max(sum x^2, eps)) -- differentiated as written.
"""
from __future__ import annotations

import numpy as np
import torch

from . import n2nmn_oracle as O

_ENC, _DEC, _MOD = O._ENC, O._DEC, O._MOD
F64 = torch.float64


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=F64)


# Discrete selections (which argument of a min/max wins, which pixel is the extremum) make the
# gradient discontinuous: when two candidates are closer than fp32 round-off, an fp32
# implementation may legitimately route the gradient to the other candidate.  The oracle records,
# per example, the smallest gap it saw, so that parity tests can apply the same rule SURVEY.md 8(c)
# states for the decoder's argmax (compare only where the oracle's margin is clear).
_MARGIN = {'example': None, 'min_gap': {}}


def _note_gap(gap: float):
    n = _MARGIN['example']
    if n is not None:
        _MARGIN['min_gap'][n] = min(_MARGIN['min_gap'].get(n, float('inf')), float(gap))


class _MinMax2(torch.autograd.Function):
    """tf.minimum / tf.maximum with TF's tie rule (first argument wins)."""

    @staticmethod
    def forward(ctx, x, y, is_min):
        mask = (x <= y) if is_min else (x >= y)
        d = (x - y).abs()
        if d.numel() and bool((d > 0).any()):
            _note_gap(float(d[d > 0].min()))
        ctx.save_for_backward(mask)
        return torch.where(mask, x, y)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        z = torch.zeros_like(g)
        return torch.where(mask, g, z), torch.where(mask, z, g), None


def tf_minimum(x, y):
    return _MinMax2.apply(x, y, True)


def tf_maximum(x, y):
    return _MinMax2.apply(x, y, False)


class _ReduceExt(torch.autograd.Function):
    """tf.reduce_min / reduce_max over the last axis; gradient split between tied positions."""

    @staticmethod
    def forward(ctx, x, is_min):
        y = x.min(dim=-1, keepdim=True).values if is_min else x.max(dim=-1, keepdim=True).values
        ind = (x == y).to(x.dtype)
        d = (x - y).abs()
        if bool((d > 0).any()):
            _note_gap(float(d[d > 0].min()))          # runner-up vs extremum
        ctx.save_for_backward(ind)
        return y

    @staticmethod
    def backward(ctx, g):
        (ind,) = ctx.saved_tensors
        return g * ind / ind.sum(dim=-1, keepdim=True), None


def tf_reduce_min(x):
    return _ReduceExt.apply(x, True)


def tf_reduce_max(x):
    return _ReduceExt.apply(x, False)


def _l2n(x, dim):
    ss = torch.sum(x * x, dim=dim, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=1e-12))


def _lstm_cell(x, c, h, Wm, bm):
    z = torch.cat([x, h], dim=1) @ Wm + bm                     # Appendix A.1
    i, j, f, o = torch.chunk(z, 4, dim=1)
    c2 = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.tanh(c2) * torch.sigmoid(o)
    return c2, h2


def _lstm_w(w, which, layer):
    base = (_ENC if which == 'encoder' else _DEC) + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/' % layer
    return w[base + 'weights'], w[base + 'biases']


def encoder_forward(w, input_seq, seq_len, drop0=None):
    """models_clevr/nmn3_netgen_att.py:73-113 (dynamic_rnn with sequence_length, Appendix A.2).
    drop0 [T, N, L] of {0, 1} (encoder_dropout=True, models_vqa/nmn3_netgen_att.py:17-44): the
    DropoutWrapper(output_keep_prob=0.5) around every layer but the last scales the OUTPUT that
    feeds the next layer by mask / 0.5; the recurrent state h is not touched."""
    T, N = input_seq.shape
    seq = torch.as_tensor(np.asarray(input_seq)).long()
    lens = torch.as_tensor(np.asarray(seq_len)).long()
    E = w[_ENC + 'embedding_mat'][seq]                          # [T, N, E]
    L = w[_ENC + 'encoder_h_transform/weights'].shape[0]
    W0, b0 = _lstm_w(w, 'encoder', 0)
    W1, b1 = _lstm_w(w, 'encoder', 1)
    z = torch.zeros(N, L, dtype=F64)
    c0, h0, c1, h1 = z, z, z, z
    outs = []
    for t in range(T):
        act = (t < lens)[:, None]
        nc0, nh0 = _lstm_cell(E[t], c0, h0, W0, b0)
        x1 = nh0 if drop0 is None else nh0 * (_t(drop0[t]) * 2.0)
        nc1, nh1 = _lstm_cell(x1, c1, h1, W1, b1)
        outs.append(torch.where(act, nh1, torch.zeros_like(nh1)))
        c0 = torch.where(act, nc0, c0); h0 = torch.where(act, nh0, h0)
        c1 = torch.where(act, nc1, c1); h1 = torch.where(act, nh1, h1)
    outs = torch.stack(outs)
    eht = (outs.reshape(T * N, L) @ w[_ENC + 'encoder_h_transform/weights']
           + w[_ENC + 'encoder_h_transform/biases']).reshape(T, N, L)
    nf = (torch.arange(T)[:, None] < lens[None, :]).to(F64)[:, :, None]
    return dict(embedded=E, outputs=outs, h_transformed=eht, not_finished=nf,
                states=((c0, h0), (c1, h1)))


def decoder_forward_gt(w, enc, T_dec, gt_layout, drop0=None):
    """Teacher-forced decoder (nmn3_netgen_att.py:175-312 with use_gt_layout=True): all tokens
    valid, predicted_token = gt, token_prob = softmax(token_scores)[gt].  drop0 [T_dec, N, L]:
    decoder_dropout=True, as in encoder_forward."""
    (c0, h0), (c1, h1) = enc['states']
    N = h0.shape[0]
    gt = torch.as_tensor(np.asarray(gt_layout)).long()
    demb = w[_DEC + 'embedding_mat']
    x = w[_DEC + 'go_embedding'].expand(N, -1)
    v = w[_DEC + 'att_prediction/v']
    Wa, ba = w[_DEC + 'att_prediction/weights'], w[_DEC + 'att_prediction/biases']
    Wy, by = w[_DEC + 'token_prediction/weights'], w[_DEC + 'token_prediction/biases']
    W0, b0 = _lstm_w(w, 'decoder', 0)
    W1, b1 = _lstm_w(w, 'decoder', 1)
    eht, eout, nf = enc['h_transformed'], enc['outputs'], enc['not_finished']
    atts, scores = [], []
    for t in range(T_dec):
        c0, h0 = _lstm_cell(x, c0, h0, W0, b0)
        c1, h1 = _lstm_cell(h0 if drop0 is None else h0 * (_t(drop0[t]) * 2.0), c1, h1, W1, b1)
        out = h1
        q = out @ Wa + ba
        e = torch.sum(torch.tanh(q[None] + eht) * v, dim=2, keepdim=True)
        att = torch.softmax(e, dim=0) * nf                       # :190
        att = att / torch.sum(att, dim=0, keepdim=True)          # :191
        ctx = torch.sum(att * eout, dim=0)
        sc = torch.cat([out, ctx], dim=1) @ Wy + by
        x = demb[gt[t]]
        atts.append(att); scores.append(sc)
    atts = torch.stack(atts)                                     # [T_dec, T_enc, N, 1]
    scores = torch.stack(scores)                                 # [T_dec, N, V]
    p = torch.softmax(scores, dim=2)                             # validity_mult == 1 everywhere
    p = p / torch.sum(p, dim=2, keepdim=True)                    # :245-247
    tprobs = torch.gather(p, 2, gt[:, :, None])[:, :, 0]         # :251-256
    word_vecs = torch.sum(atts * enc['embedded'][None], dim=1)   # :312
    return dict(token_probs=tprobs, atts=atts, word_vecs=word_vecs, token_scores=scores)


# ---- module operators (models_clevr/nmn3_modules.py) on torch tensors ---------------------
def _fc(w, scope, x):
    return x @ w[_MOD + scope + '/weights'] + w[_MOD + scope + '/biases']


def _conv1x1(w, scope, x):
    shp = x.shape
    y = x.reshape(-1, shp[-1]) @ w[_MOD + scope + '/weights'] + w[_MOD + scope + '/biases']
    return y.reshape(tuple(shp[:-1]) + (y.shape[-1],))


def _conv_same(w, scope, x):
    K = w[_MOD + scope + '/weights']                             # [kh, kw, cin, cout]
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), K.permute(3, 2, 0, 1),
                                   padding=(K.shape[0] // 2, K.shape[1] // 2))
    return y.permute(0, 2, 3, 1) + w[_MOD + scope + '/biases']


def _att_pool(feat, att):
    n, H, Wd, _ = att.shape
    a = torch.softmax(att.reshape(n, H * Wd), dim=1).reshape(n, H, Wd, 1)
    return torch.sum(feat * a, dim=(1, 2))


def _flat_ext(in0):
    f = in0.reshape(in0.shape[0], -1)
    return f, tf_reduce_min(f), tf_reduce_max(f)


def _compare(w, scope, in0, in1):
    f0, mn0, mx0 = _flat_ext(in0)
    f1, mn1, mx1 = _flat_ext(in1)
    return _fc(w, scope + '/fc_scores', torch.cat([f0, mn0, mx0, f1, mn1, mx1], dim=1))


def eval_expr(w, expr, image_feat, word_vecs, num_choices):
    if expr['module'] == O.INVALID:
        return torch.zeros(num_choices, dtype=F64)
    N_full = word_vecs.shape[1]
    flat = word_vecs.reshape(-1, word_vecs.shape[-1])
    H, Wd = image_feat.shape[1:3]

    def find(scope, feat, txt, extra=None):
        img = _conv1x1(w, scope + 'conv_image', feat)
        t = _fc(w, scope + 'fc_text', txt)[:, None, None, :]
        el = img * t if extra is None else img * t * extra
        return _conv1x1(w, scope + 'conv_eltwise', _l2n(el, 3))

    def rec(e):
        t, n = e['time_idx'], e['batch_idx']
        feat = image_feat[n:n + 1]
        txt = flat[t * N_full + n][None]
        ins = [rec(e[k]) for k in ('input_0', 'input_1') if k in e]
        m = e['module']
        if m == '_Scene':
            return torch.full((1, H, Wd, 1), 3.0, dtype=F64)
        if m == '_Find':
            return find('FindModule/', feat, txt)
        if m == '_Filter':                                       # And(input_0, Find(...))  :129-130
            return tf_minimum(ins[0], find('FindModule/', feat, txt))
        if m == '_FindSameProperty':
            s = 'FindSamePropertyModule/'
            a = _fc(w, s + 'fc_att', _att_pool(feat, ins[0]))[:, None, None, :]
            return find(s, feat, txt, a)
        if m == '_Transform':
            s = 'TransformModule/'
            maps = _conv_same(w, s + 'conv_maps', ins[0])
            tt = _fc(w, s + 'text_fc', txt)[:, None, None, :]
            return _conv1x1(w, s + 'conv_eltwise', _l2n(maps * tt, 3))
        if m == '_And':
            return tf_minimum(ins[0], ins[1])
        if m == '_Or':
            return tf_maximum(ins[0], ins[1])
        if m == '_Exist':
            f, mn, mx = _flat_ext(ins[0])
            return _fc(w, 'ExistModule/fc_scores',
                       torch.cat([mn, f.mean(dim=1, keepdim=True), mx], dim=1))
        if m == '_Count':
            f, mn, mx = _flat_ext(ins[0])
            return _fc(w, 'CountModule/fc_scores', torch.cat([f, mn, mx], dim=1))
        if m == '_EqualNum':
            return _compare(w, 'EqualNumModule', ins[0], ins[1])
        if m == '_MoreNum':
            return _compare(w, 'MoreNumModule', ins[0], ins[1])
        if m == '_LessNum':
            return _compare(w, 'LessNumModule', ins[0], ins[1])
        if m == '_SameProperty':
            s = 'SamePropertyModule/'
            tt = _fc(w, s + 'fc_text', txt)
            a0 = _fc(w, s + 'fc_att_0', _att_pool(feat, ins[0]))
            a1 = _fc(w, s + 'fc_att_1', _att_pool(feat, ins[1]))
            return _fc(w, s + 'fc_eltwise', _l2n(a0 * tt * a1, 1))
        if m == '_Describe':
            s = 'DescribeModule/'
            tt = _fc(w, s + 'fc_text', txt)
            a = _fc(w, s + 'fc_att', _att_pool(feat, ins[0]))
            return _fc(w, s + 'fc_eltwise', _l2n(tt * a, 1))
        raise KeyError(m)

    return rec(expr)[0]


def train_forward(wt, module_names, batch, T_dec, num_choices, gt_layout, weight_decay=5e-6):
    """wt: name -> torch fp64 tensor (requires_grad as the caller wishes).  Returns dict of torch
    scalars / tensors incl. the intermediates whose gradients the GPU tests compare."""
    enc = encoder_forward(wt, batch['input_seq_batch'], batch['seq_length_batch'])
    dec = decoder_forward_gt(wt, enc, T_dec, gt_layout)
    exprs, validity = O.assemble(module_names, np.asarray(gt_layout))
    feat = _t(batch['image_feat_batch'])
    _MARGIN['min_gap'] = {}
    rows = []
    for n, e in enumerate(exprs):
        _MARGIN['example'] = n
        rows.append(eval_expr(wt, e, feat, dec['word_vecs'], num_choices))
    _MARGIN['example'] = None
    scores = torch.stack(rows)
    labels = torch.as_tensor(np.asarray(batch['answer_label_batch'])).long()
    log_seq_prob = torch.sum(torch.log(dec['token_probs']), dim=0)     # nmn3_model.py:46
    ce = torch.logsumexp(scores, dim=1) - scores[torch.arange(len(labels)), labels]
    avg_sample_loss = ce.mean()                                        # :110
    seq_likelihood_loss = torch.mean(-log_seq_prob)                    # :111
    l2 = sum(0.5 * torch.sum(v * v) for k, v in wt.items() if k.endswith('weights'))
    total = seq_likelihood_loss + avg_sample_loss + weight_decay * l2   # :113-114
    return dict(enc=enc, dec=dec, expr_list=exprs, validity=validity, scores=scores,
                log_seq_prob=log_seq_prob, avg_sample_loss=avg_sample_loss,
                seq_likelihood_loss=seq_likelihood_loss, l2_reg=l2, total_loss=total)


def loss_and_grads(w, module_names, batch, T_dec, num_choices, gt_layout, weight_decay=5e-6):
    """numpy in / numpy out.  Returns (losses dict of floats, grads name -> ndarray fp64,
    extras: gradients w.r.t. intermediates + forward scores)."""
    wt = {k: _t(v).clone().requires_grad_(True) for k, v in w.items()}
    r = train_forward(wt, module_names, batch, T_dec, num_choices, gt_layout, weight_decay)
    inter = dict(word_vecs=r['dec']['word_vecs'], token_scores=r['dec']['token_scores'],
                 scores=r['scores'], encoder_outputs=r['enc']['outputs'],
                 encoder_h_transformed=r['enc']['h_transformed'], atts=r['dec']['atts'])
    for v in inter.values():
        if v.requires_grad:
            v.retain_grad()
    r['total_loss'].backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape)))
             for k, v in wt.items()}
    losses = {k: float(r[k].detach()) for k in ('avg_sample_loss', 'seq_likelihood_loss', 'l2_reg',
                                       'total_loss')}
    extras = {'d_' + k: (v.grad.numpy().copy() if v.grad is not None else None)
              for k, v in inter.items()}
    extras['scores'] = r['scores'].detach().numpy().copy()
    # smallest min/max selection gap per example (inf: no discrete selection in its layout)
    extras['selection_gap'] = np.array([_MARGIN['min_gap'].get(n, np.inf)
                                        for n in range(len(r['expr_list']))])
    extras['log_seq_prob'] = r['log_seq_prob'].detach().numpy().copy()
    return losses, grads, extras


# ---- policy-gradient objective: exp_clevr/train_clevr_rl_gt_layout.py:107-129 ----------------
# ---- models_vqa training (exp_vqa/train_vqa_gt_layout.py:83-116) ------------------------------
_QPN = O._QPN


def add_spatial_coordinate_map(feat):
    """models_vqa/nmn3_modules.py:11-31 (linspace computed in float32 like TF)."""
    N, H, W, _ = feat.shape
    x = _t(np.linspace(-1.0, 1.0, W, dtype=np.float32))
    y = _t(np.linspace(-1.0, 1.0, H, dtype=np.float32))
    xm = x[None, None, :, None].expand(N, H, W, 1)
    ym = y[None, :, None, None].expand(N, H, W, 1)
    return torch.cat([feat, xm, ym], dim=3)


def eval_expr_vqa(w, expr, feat_c, word_vecs, num_choices):
    """models_vqa/nmn3_modules.py:82-240: _Find, _Transform (the three-way product), _And, _Describe."""
    if expr['module'] == O.INVALID:
        return torch.zeros(num_choices, dtype=F64)
    N_full = word_vecs.shape[1]
    flat = word_vecs.reshape(-1, word_vecs.shape[-1])

    def find(scope, feat, txt, extra=None):
        img = _conv1x1(w, scope + 'conv_image', feat)
        t = _fc(w, scope + 'fc_text', txt)[:, None, None, :]
        el = img * t if extra is None else img * t * extra
        return _conv1x1(w, scope + 'conv_eltwise', _l2n(el, 3))

    def rec(e):
        t, n = e['time_idx'], e['batch_idx']
        feat = feat_c[n:n + 1]
        txt = flat[t * N_full + n][None]
        ins = [rec(e[k]) for k in ('input_0', 'input_1') if k in e]
        m = e['module']
        if m == '_Find':
            return find('FindModule/', feat, txt)
        if m == '_Transform':
            s = 'TransformModule/'
            a = _fc(w, s + 'fc_att', _att_pool(feat, ins[0]))[:, None, None, :]
            return find(s, feat, txt, a)
        if m == '_And':
            return tf_minimum(ins[0], ins[1])
        if m == '_Describe':
            s = 'DescribeModule/'
            tt = _fc(w, s + 'fc_text', txt)
            a = _fc(w, s + 'fc_att', _att_pool(feat, ins[0]))
            return _fc(w, s + 'fc_eltwise', _l2n(tt * a, 1))
        raise KeyError(m)

    return rec(expr)[0]


def question_prior_net(w, enc_states, drop_h=None, drop_fc1=None):
    """models_vqa/question_prior_net.py:10-28; drop_* are {0, 1} keep masks (qpn_dropout=True,
    keep_prob 0.5 -> kept activations are doubled)."""
    h = torch.cat([st[1] for st in enc_states], dim=1)
    if drop_h is not None:
        h = h * (_t(drop_h) * 2.0)
    fc1 = torch.relu(h @ w[_QPN + 'fc1/weights'] + w[_QPN + 'fc1/biases'])
    if drop_fc1 is not None:
        fc1 = fc1 * (_t(drop_fc1) * 2.0)
    return fc1 @ w[_QPN + 'fc2/weights'] + w[_QPN + 'fc2/biases']


def train_forward_vqa(wt, batch, T_dec, num_choices, gt_layout, masks=None, weight_decay=0.0):
    """exp_vqa/train_vqa_gt_layout.py:83-116: total = mean(-log_seq_prob) + mean(CE(scores_nmn +
    scores_qpn)) + weight_decay * l2 (weight_decay = 0 there; NO gradient clipping, :119-123).
    masks: dict enc0 [T_enc, N, L], dec0 [T_dec, N, L], qpn_h [N, 2L], qpn_fc1 [N, hidden] of {0, 1}
    keep masks (TF draws them from its RNG; here they are inputs), or None for no dropout."""
    mk = masks or {}
    names = list(O.VQA_MODULE_NAMES)
    enc = encoder_forward(wt, batch['input_seq_batch'], batch['seq_length_batch'], mk.get('enc0'))
    dec = decoder_forward_gt(wt, enc, T_dec, gt_layout, mk.get('dec0'))
    exprs, validity = O.assemble(names, np.asarray(gt_layout))
    feat_c = add_spatial_coordinate_map(_t(batch['image_feat_batch']))
    _MARGIN['min_gap'] = {}
    rows = []
    for n, e in enumerate(exprs):
        _MARGIN['example'] = n
        rows.append(eval_expr_vqa(wt, e, feat_c, dec['word_vecs'], num_choices))
    _MARGIN['example'] = None
    scores_nmn = torch.stack(rows)
    scores = scores_nmn + question_prior_net(wt, enc['states'], mk.get('qpn_h'), mk.get('qpn_fc1'))
    labels = torch.as_tensor(np.asarray(batch['answer_label_batch'])).long()
    log_seq_prob = torch.sum(torch.log(dec['token_probs']), dim=0)
    ce = torch.logsumexp(scores, dim=1) - scores[torch.arange(len(labels)), labels]
    avg_sample_loss = ce.mean()
    seq_likelihood_loss = torch.mean(-log_seq_prob)
    l2 = sum(0.5 * torch.sum(v * v) for k, v in wt.items() if k.endswith('weights'))
    total = seq_likelihood_loss + avg_sample_loss + weight_decay * l2
    return dict(enc=enc, dec=dec, expr_list=exprs, validity=validity, scores=scores,
                scores_nmn=scores_nmn, log_seq_prob=log_seq_prob, avg_sample_loss=avg_sample_loss,
                seq_likelihood_loss=seq_likelihood_loss, l2_reg=l2, total_loss=total)


def loss_and_grads_vqa(w, batch, T_dec, num_choices, gt_layout, masks=None, weight_decay=0.0):
    """numpy in / numpy out: (losses, grads name -> ndarray fp64, extras)."""
    wt = {k: _t(v).clone().requires_grad_(True) for k, v in w.items()}
    r = train_forward_vqa(wt, batch, T_dec, num_choices, gt_layout, masks, weight_decay)
    r['total_loss'].backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape)))
             for k, v in wt.items()}
    losses = {k: float(r[k].detach()) for k in ('avg_sample_loss', 'seq_likelihood_loss', 'l2_reg',
                                                 'total_loss')}
    extras = dict(scores=r['scores'].detach().numpy().copy(),
                  log_seq_prob=r['log_seq_prob'].detach().numpy().copy(),
                  selection_gap=np.array([_MARGIN['min_gap'].get(n, np.inf)
                                          for n in range(len(r['expr_list']))]))
    return losses, grads, extras


def decoder_forward_tokens(w, enc, T_dec, tokens, token_validity, drop0=None):
    """Decoder run on GIVEN tokens with the automaton's validity masks (constants: they depend on
    the tokens only): what the sampling decoder computed when it drew `tokens`
    (nmn3_netgen_att.py:175-312 with decoder_sampling=True), as a differentiable function of the
    weights.  token_validity [T_dec, N, V] bool."""
    (c0, h0), (c1, h1) = enc['states']
    N = h0.shape[0]
    tok = torch.as_tensor(np.asarray(tokens)).long()
    vm = torch.as_tensor(np.asarray(token_validity, np.float64))
    demb = w[_DEC + 'embedding_mat']
    x = w[_DEC + 'go_embedding'].expand(N, -1)
    v = w[_DEC + 'att_prediction/v']
    Wa, ba = w[_DEC + 'att_prediction/weights'], w[_DEC + 'att_prediction/biases']
    Wy, by = w[_DEC + 'token_prediction/weights'], w[_DEC + 'token_prediction/biases']
    W0, b0 = _lstm_w(w, 'decoder', 0)
    W1, b1 = _lstm_w(w, 'decoder', 1)
    eht, eout, nf = enc['h_transformed'], enc['outputs'], enc['not_finished']
    atts, scores = [], []
    for t in range(T_dec):
        c0, h0 = _lstm_cell(x, c0, h0, W0, b0)
        c1, h1 = _lstm_cell(h0 if drop0 is None else h0 * (_t(drop0[t]) * 2.0), c1, h1, W1, b1)
        q = h1 @ Wa + ba
        e = torch.sum(torch.tanh(q[None] + eht) * v, dim=2, keepdim=True)
        att = torch.softmax(e, dim=0) * nf
        att = att / torch.sum(att, dim=0, keepdim=True)
        ctx = torch.sum(att * eout, dim=0)
        sc = torch.cat([h1, ctx], dim=1) @ Wy + by
        x = demb[tok[t]]
        atts.append(att); scores.append(sc)
    atts = torch.stack(atts)
    scores = torch.stack(scores)                                 # [T_dec, N, V]
    p = torch.softmax(scores, dim=2) * vm                        # :245
    p = p / torch.sum(p, dim=2, keepdim=True)                    # :247
    tprobs = torch.gather(p, 2, tok[:, :, None])[:, :, 0]        # :251-256
    neg_entropy = torch.sum(p * torch.log(torch.clamp(p + (1.0 - vm), min=1e-5)), dim=(0, 2))  # :258-260
    word_vecs = torch.sum(atts * enc['embedded'][None], dim=1)
    return dict(token_probs=tprobs, neg_entropy=neg_entropy, atts=atts, word_vecs=word_vecs,
                token_scores=scores)


def train_forward_rl(wt, module_names, batch, T_dec, num_choices, tokens, token_validity, baseline,
                     invalid_expr_loss=0.5, lambda_entropy=0.005, weight_decay=5e-6,
                     validity_override=None, vqa_masks=None):
    """Loss of train_clevr_rl_gt_layout.py:107-129 for one batch whose layouts `tokens` were sampled
    by the decoder.  baseline: python float (tf.Variable, not trainable).
    vqa_masks (dict, possibly empty) selects the models_vqa network of
    exp_vqa/train_vqa_rl_gt_layout.py:106-126 (same loss; module_names = VQA_MODULE_NAMES) with the
    given dropout keep masks (enc0 / dec0 / qpn_h / qpn_fc1, see train_forward_vqa)."""
    mk = vqa_masks or {}
    enc = encoder_forward(wt, batch['input_seq_batch'], batch['seq_length_batch'], mk.get('enc0'))
    dec = decoder_forward_tokens(wt, enc, T_dec, tokens, token_validity, mk.get('dec0'))
    exprs, validity = O.assemble(module_names, np.asarray(tokens))
    if validity_override is not None:           # expr_validity_batch is a placeholder (:86): the
        validity = np.asarray(validity_override, bool)   # loss sees whatever the caller feeds
    feat = _t(batch['image_feat_batch'])
    if vqa_masks is not None:
        feat = add_spatial_coordinate_map(feat)
    _MARGIN['min_gap'] = {}
    rows = []
    for n, e in enumerate(exprs):
        _MARGIN['example'] = n
        if validity[n]:
            rows.append(eval_expr_vqa(wt, e, feat, dec['word_vecs'], num_choices)
                        if vqa_masks is not None else
                        eval_expr(wt, e, feat, dec['word_vecs'], num_choices))
        else:                                   # INVALID_EXPR: zero logits (nmn3_model.py:146,155)
            rows.append(torch.zeros(num_choices, dtype=torch.float64))
    _MARGIN['example'] = None
    scores = torch.stack(rows)
    if vqa_masks is not None:
        scores = scores + question_prior_net(wt, enc['states'], mk.get('qpn_h'), mk.get('qpn_fc1'))
    labels = torch.as_tensor(np.asarray(batch['answer_label_batch'])).long()
    log_seq_prob = torch.sum(torch.log(dec['token_probs']), dim=0)
    ce = torch.logsumexp(scores, dim=1) - scores[torch.arange(len(labels)), labels]
    valid_t = torch.as_tensor(np.asarray(validity, bool))
    final = torch.where(valid_t, ce, torch.full_like(ce, invalid_expr_loss))      # :112-114
    avg_sample_loss = final.mean()                                                # :119
    policy = torch.mean((final - baseline).detach() * log_seq_prob)               # :123-124
    entropy_reg = dec['neg_entropy'].mean()                                       # nmn3_model.py:162
    l2 = sum(0.5 * torch.sum(v * v) for k, v in wt.items() if k.endswith('weights'))
    total = policy + avg_sample_loss + lambda_entropy * entropy_reg + weight_decay * l2   # :126-129
    return dict(enc=enc, dec=dec, expr_list=exprs, validity=validity, scores=scores,
                log_seq_prob=log_seq_prob, avg_sample_loss=avg_sample_loss,
                policy_gradient_loss=policy, entropy_reg=entropy_reg, l2_reg=l2, total_loss=total)


def loss_and_grads_rl(w, module_names, batch, T_dec, num_choices, tokens, token_validity, baseline,
                      invalid_expr_loss=0.5, lambda_entropy=0.005, weight_decay=5e-6,
                      baseline_decay=0.99, validity_override=None, vqa_masks=None):
    """numpy in / numpy out; like loss_and_grads.  losses additionally hold 'new_baseline'
    (baseline + (1 - decay) * (avg_sample_loss - baseline), :120-122)."""
    wt = {k: _t(v).clone().requires_grad_(True) for k, v in w.items()}
    r = train_forward_rl(wt, module_names, batch, T_dec, num_choices, tokens, token_validity,
                         baseline, invalid_expr_loss, lambda_entropy, weight_decay,
                         validity_override, vqa_masks)
    inter = dict(word_vecs=r['dec']['word_vecs'], token_scores=r['dec']['token_scores'],
                 scores=r['scores'])
    for v in inter.values():
        if v.requires_grad:
            v.retain_grad()
    r['total_loss'].backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape)))
             for k, v in wt.items()}
    losses = {k: float(r[k].detach()) for k in ('avg_sample_loss', 'policy_gradient_loss',
                                                'entropy_reg', 'l2_reg', 'total_loss')}
    losses['new_baseline'] = baseline + (1.0 - baseline_decay) * (losses['avg_sample_loss'] - baseline)
    extras = {'d_' + k: (v.grad.numpy().copy() if v.grad is not None else None)
              for k, v in inter.items()}
    extras['scores'] = r['scores'].detach().numpy().copy()
    extras['validity'] = np.asarray(r['validity'], bool)
    extras['selection_gap'] = np.array([_MARGIN['min_gap'].get(n, np.inf)
                                        for n in range(len(r['expr_list']))])
    extras['log_seq_prob'] = r['log_seq_prob'].detach().numpy().copy()
    extras['neg_entropy'] = r['dec']['neg_entropy'].detach().numpy().copy()
    return losses, grads, extras


# ---- optimiser: exp_clevr/train_clevr_gt_layout.py:112-120 -------------------------------
def clip_by_norm(g, c):
    """tf.clip_by_norm(g, c) = g * c / max(||g||_2, c)  (Appendix A.4)."""
    return g * (c / max(float(np.sqrt(np.sum(np.asarray(g, np.float64) ** 2))), c))


def adam_step(w, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
              max_grad_l2_norm=10.0):
    """One tf.train.AdamOptimizer() step (TF 1.0.0 defaults) on per-tensor clipped gradients.
    step = 1 for the first update.  All dicts name -> ndarray; returns (w', m', v')."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    w2, m2, v2 = {}, {}, {}
    for k in w:
        g = np.asarray(grads[k], np.float64)
        if max_grad_l2_norm is not None:           # exp_vqa/train_vqa_gt_layout.py:119-123: no clipping
            g = clip_by_norm(g, max_grad_l2_norm)
        m2[k] = beta1 * m[k] + (1.0 - beta1) * g
        v2[k] = beta2 * v[k] + (1.0 - beta2) * g * g
        w2[k] = np.asarray(w[k], np.float64) - lr_t * m2[k] / (np.sqrt(v2[k]) + eps)
    return w2, m2, v2
