"""Batched torch-CPU port of the hot path -- TEST INFRASTRUCTURE / bench.py's `cpu_baseline` leg only.

SURVEY.md section 8(d) asks for the CPU number beside the GPU number to come from a port that mirrors
the reference's TensorFlow op granularity (the reference's own TF1 CPU path cannot run here: no
TensorFlow, no network): one batched matmul per LSTM gate block and time step
(models_clevr/nmn3_netgen_att.py:73-113, 175-304), `conv2d` for TransformModule
(models_clevr/nmn3_modules.py:185-216), and TensorFlow-Fold's dynamic batching -- ONE call per
(module type, tree depth) over the stacked instances (models_clevr/nmn3_model.py:55-159) -- on
`torch.get_num_threads()` host threads, float32.  It is checked against the numpy oracle
(oracle/n2nmn_oracle.py, itself pinned to the reference's code by
tests/test_oracle_vs_reference_code.py) in tests/test_oracle_batched.py and, in-process, by bench.py
before it is timed.  Never imported by the product.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import n2nmn_oracle as O

_ENC, _DEC, _MOD = O._ENC, O._DEC, O._MOD


def to_torch(w, dtype=torch.float32):
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in w.items()}


def _lstm(x, c, h, Wm, bm):
    z = torch.cat([x, h], 1) @ Wm + bm                     # one batched matmul per cell and step
    i, j, f, o = z.chunk(4, dim=1)
    c2 = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    return c2, torch.tanh(c2) * torch.sigmoid(o)


def _cell(w, which, layer):
    base = (_ENC if which == 'encoder' else _DEC) + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/' % layer
    return w[base + 'weights'], w[base + 'biases']


def seq2seq(w, input_seq, seq_len, P, Wv, bv, T_dec, use_gt_layout=False, gt_layout=None):
    dt = w[_ENC + 'embedding_mat'].dtype
    seq = torch.as_tensor(np.asarray(input_seq)).long()
    lens = torch.as_tensor(np.asarray(seq_len)).long()
    T, N = seq.shape
    Emb = w[_ENC + 'embedding_mat'][seq]                   # [T, N, E]
    L = w[_ENC + 'encoder_h_transform/weights'].shape[0]
    (W0, b0), (W1, b1) = _cell(w, 'encoder', 0), _cell(w, 'encoder', 1)
    c0 = torch.zeros(N, L, dtype=dt); h0 = c0.clone(); c1 = c0.clone(); h1 = c0.clone()
    outs = []
    for t in range(T):
        act = (t < lens)[:, None]
        nc0, nh0 = _lstm(Emb[t], c0, h0, W0, b0)
        nc1, nh1 = _lstm(nh0, c1, h1, W1, b1)
        outs.append(torch.where(act, nh1, torch.zeros_like(nh1)))
        c0 = torch.where(act, nc0, c0); h0 = torch.where(act, nh0, h0)
        c1 = torch.where(act, nc1, c1); h1 = torch.where(act, nh1, h1)
    eout = torch.stack(outs)
    eht = (eout.reshape(T * N, L) @ w[_ENC + 'encoder_h_transform/weights'] +
           w[_ENC + 'encoder_h_transform/biases']).reshape(T, N, L)
    nf = (torch.arange(T)[:, None] < lens[None, :]).to(dt)[:, :, None]
    # decoder (raw_rnn: exactly T_dec cell calls)
    demb = w[_DEC + 'embedding_mat']
    V = demb.shape[0]
    v = w[_DEC + 'att_prediction/v']
    Wa, ba = w[_DEC + 'att_prediction/weights'], w[_DEC + 'att_prediction/biases']
    Wy, by = w[_DEC + 'token_prediction/weights'], w[_DEC + 'token_prediction/biases']
    (D0, d0), (D1, d1) = _cell(w, 'decoder', 0), _cell(w, 'decoder', 1)
    Pt = torch.as_tensor(np.asarray(P)).long()
    Wt = torch.as_tensor(np.asarray(Wv)).long()
    bt = torch.as_tensor(np.asarray(bv)).long()
    X = torch.tensor([[0, 0, T_dec]]).repeat(N, 1)
    x = w[_DEC + 'go_embedding'].repeat(N, 1)
    gt = torch.as_tensor(np.asarray(gt_layout)).long() if gt_layout is not None else None
    tokens, tprobs, atts, tscores, tvalid = [], [], [], [], []
    neg_ent = torch.zeros(N, dtype=dt)
    for t in range(T_dec):
        c0, h0 = _lstm(x, c0, h0, D0, d0)
        c1, h1 = _lstm(h0, c1, h1, D1, d1)
        q = h1 @ Wa + ba
        e = (torch.tanh(q[None] + eht) * v).sum(2, keepdim=True)
        att = torch.softmax(e, dim=0) * nf
        att = att / att.sum(0, keepdim=True)
        ctx = (att * eout).sum(0)
        sc = torch.cat([h1, ctx], 1) @ Wy + by
        valid = ((torch.tensordot(X, Wt, dims=1) - bt) >= 0).all(2)
        if use_gt_layout:
            valid = torch.ones_like(valid)
        vm = valid.to(dt)
        masked = torch.where(valid, sc, torch.full_like(sc, float(sc.min()) - 1.0))
        tok = masked.argmax(1)
        if use_gt_layout:
            tok = gt[t]
        p = torch.softmax(sc, 1) * vm
        p = p / p.sum(1, keepdim=True)
        tprobs.append(p.gather(1, tok[:, None])[:, 0])
        neg_ent = neg_ent + (p * torch.log(torch.clamp(p + (1 - vm), min=1e-5))).sum(1)
        X = X + Pt[tok]
        x = demb[tok]
        tokens.append(tok); atts.append(att); tscores.append(sc); tvalid.append(valid)
    atts = torch.stack(atts)                               # [T_dec, T, N, 1]
    word_vecs = (atts * Emb[None]).sum(1)                  # [T_dec, N, E]
    return dict(predicted_tokens=torch.stack(tokens).to(torch.int32), token_probs=torch.stack(tprobs),
                neg_entropy=neg_ent, atts=atts, word_vecs=word_vecs,
                token_scores=torch.stack(tscores), token_validity=torch.stack(tvalid))


def _fc(w, s, x):
    return x @ w[_MOD + s + '/weights'] + w[_MOD + s + '/biases']


def _l2n(x, dim):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim, keepdim=True), min=1e-12))


def _pool(feat, att):
    nb = att.shape[0]
    p = torch.softmax(att.reshape(nb, -1), 1).reshape(att.shape)
    return (feat * p).sum((1, 2))


def _module(w, name, ins, feat, txt, C):
    """One batched call, like the Fold-compiled op of that module at one depth."""
    nb = txt.shape[0]
    if name == '_Scene':
        return torch.full((nb,) + tuple(feat.shape[1:3]) + (1,), 3.0, dtype=feat.dtype)
    if name in ('_Find', '_Filter'):
        img = _fc(w, 'FindModule/conv_image', feat)
        t = _fc(w, 'FindModule/fc_text', txt)[:, None, None]
        out = _fc(w, 'FindModule/conv_eltwise', _l2n(img * t, 3))
        return torch.minimum(ins[0], out) if name == '_Filter' else out
    if name == '_FindSameProperty':
        s = 'FindSamePropertyModule/'
        img = _fc(w, s + 'conv_image', feat)
        t = _fc(w, s + 'fc_text', txt)[:, None, None]
        a = _fc(w, s + 'fc_att', _pool(feat, ins[0]))[:, None, None]
        return _fc(w, s + 'conv_eltwise', _l2n(img * t * a, 3))
    if name == '_Transform':
        K = w[_MOD + 'TransformModule/conv_maps/weights'].permute(3, 2, 0, 1)
        maps = F.conv2d(ins[0].permute(0, 3, 1, 2), K, w[_MOD + 'TransformModule/conv_maps/biases'],
                        padding=K.shape[-1] // 2).permute(0, 2, 3, 1)
        t = _fc(w, 'TransformModule/text_fc', txt)[:, None, None]
        return _fc(w, 'TransformModule/conv_eltwise', _l2n(maps * t, 3))
    if name == '_And':
        return torch.minimum(ins[0], ins[1])
    if name == '_Or':
        return torch.maximum(ins[0], ins[1])
    f = [a.reshape(nb, -1) for a in ins]
    if name == '_Exist':
        x = torch.stack([f[0].min(1).values, f[0].mean(1), f[0].max(1).values], 1)
        return _fc(w, 'ExistModule/fc_scores', x)
    if name == '_Count':
        x = torch.cat([f[0], f[0].min(1, True).values, f[0].max(1, True).values], 1)
        return _fc(w, 'CountModule/fc_scores', x)
    if name in ('_EqualNum', '_MoreNum', '_LessNum'):
        x = torch.cat([f[0], f[0].min(1, True).values, f[0].max(1, True).values,
                       f[1], f[1].min(1, True).values, f[1].max(1, True).values], 1)
        return _fc(w, name[1:] + 'Module/fc_scores', x)
    if name == '_SameProperty':
        s = 'SamePropertyModule/'
        ev = _fc(w, s + 'fc_att_0', _pool(feat, ins[0])) * _fc(w, s + 'fc_text', txt) * \
            _fc(w, s + 'fc_att_1', _pool(feat, ins[1]))
        return _fc(w, s + 'fc_eltwise', _l2n(ev, 1))
    if name == '_Describe':
        s = 'DescribeModule/'
        ev = _fc(w, s + 'fc_text', txt) * _fc(w, s + 'fc_att', _pool(feat, ins[0]))
        return _fc(w, s + 'fc_eltwise', _l2n(ev, 1))
    raise KeyError(name)


def execute_layouts(w, expr_list, image_feat, word_vecs, num_choices):
    """Fold semantics: per (module, depth) one batched op; invalid -> zeros."""
    feat_all = image_feat
    N_full = word_vecs.shape[1]
    flat = word_vecs.reshape(-1, word_vecs.shape[-1])
    nodes = []

    def build(e):
        kids = [build(e[k]) for k in ('input_0', 'input_1') if k in e]
        nd = dict(m=e['module'], t=e['time_idx'], n=e['batch_idx'], kids=kids,
                  depth=1 + max([k['depth'] for k in kids] + [0]), val=None)
        nodes.append(nd)
        return nd

    roots = [None if e['module'] == O.INVALID else build(e) for e in expr_list]
    for depth in range(1, max([nd['depth'] for nd in nodes] + [0]) + 1):
        groups = {}
        for nd in nodes:
            if nd['depth'] == depth:
                groups.setdefault(nd['m'], []).append(nd)
        for name, grp in groups.items():
            bidx = torch.tensor([g['n'] for g in grp])
            tidx = torch.tensor([g['t'] for g in grp])
            feat = feat_all[bidx]                          # tf.gather copy, nmn3_modules.py:49-51
            txt = flat[tidx * N_full + bidx]
            ins = [torch.stack([g['kids'][k]['val'] for g in grp]) for k in range(len(grp[0]['kids']))]
            out = _module(w, name, ins, feat, txt, num_choices)
            for i, g in enumerate(grp):
                g['val'] = out[i]
    zero = torch.zeros(num_choices, dtype=word_vecs.dtype)
    return torch.stack([zero if r is None else r['val'] for r in roots])


def forward(w, module_names, batch, T_dec, num_choices, use_gt_layout=False, gt_layout=None):
    """w: name -> torch tensor (to_torch).  Returns dict with scores [N, C] (numpy), tokens, validity."""
    with torch.no_grad():
        P, Wv, bv = O.build_validity_mats(module_names)
        s2s = seq2seq(w, batch['input_seq_batch'], batch['seq_length_batch'], P, Wv, bv, T_dec,
                      use_gt_layout, gt_layout)
        tokens = s2s['predicted_tokens'].numpy()
        exprs, validity = O.assemble(module_names, tokens)       # host, like eval_clevr.py:125-128
        feat = torch.as_tensor(np.asarray(batch['image_feat_batch'])).to(s2s['word_vecs'].dtype)
        scores = execute_layouts(w, exprs, feat, s2s['word_vecs'], num_choices)
    return dict(scores=scores.numpy(), predicted_tokens=tokens, validity=validity, s2s=s2s)
