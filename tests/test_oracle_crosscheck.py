"""Second opinion on the float path of the oracle (round 1's only anchor; since round 2 the oracle is
also pinned to the reference's own code, tests/test_oracle_vs_reference_code.py):
an INDEPENDENT restatement on torch-CPU library ops -- torch.nn.LSTMCell (different gate order and
bias handling than TF's BasicLSTMCell, mapped explicitly), F.conv2d, F.normalize, torch.softmax --
must agree with the numpy oracle to fp64 round-off."""
import dataclasses

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

SMALL = Dims(H=4, W=5, D=32, map_dim=18, embed_dim_txt=12, embed_dim_nmn=12, lstm_dim=16,
             num_vocab_txt=11, num_choices=7, T_encoder=6, T_decoder=8, N=5)
MOD = 'neural_module_network/layout_execution/module_variables/'
ENC = 'neural_module_network/layout_generation/encoder_decoder/encoder/'
DEC = 'neural_module_network/layout_generation/encoder_decoder/decoder/'


def T(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float64)


def _tf_lstm_as_torch_cell(Wm, bm, in_dim, L):
    """TF: z=[x,h].W+b, gates (i, j, f, o), forget bias +1.  torch: gates (i, f, g, o)."""
    cell = torch.nn.LSTMCell(in_dim, L).double()
    Wt = T(Wm)
    order = [0, 2, 1, 3]                     # torch (i,f,g,o) <- tf (i,f,j,o) indices of (i,j,f,o)
    cols = torch.cat([Wt[:, g * L:(g + 1) * L] for g in order], dim=1)
    b = torch.cat([T(bm)[g * L:(g + 1) * L] for g in order]).clone()
    b[L:2 * L] += 1.0                        # forget_bias
    with torch.no_grad():
        cell.weight_ih.copy_(cols[:in_dim].T)
        cell.weight_hh.copy_(cols[in_dim:].T)
        cell.bias_ih.copy_(b)
        cell.bias_hh.zero_()
    return cell


def test_encoder_against_torch_lstmcell():
    d = SMALL
    w = synth.make_weights(d, seed=3, dtype=np.float64)
    batch = synth.make_inputs(d, seed=3, n=d.N, min_len=1)
    enc = O.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], np.float64)
    L, E = d.lstm_dim, d.embed_dim_txt
    base = ENC + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/'
    c0 = _tf_lstm_as_torch_cell(w[base % 0 + 'weights'], w[base % 0 + 'biases'], E, L)
    c1 = _tf_lstm_as_torch_cell(w[base % 1 + 'weights'], w[base % 1 + 'biases'], L, L)
    emb = T(w[ENC + 'embedding_mat'])
    seq = torch.as_tensor(batch['input_seq_batch']).long()
    lens = torch.as_tensor(batch['seq_length_batch'])
    h0 = torch.zeros(d.N, L, dtype=torch.float64); cc0 = h0.clone(); h1 = h0.clone(); cc1 = h0.clone()
    outs = []
    with torch.no_grad():
        for t in range(d.T_encoder):
            act = (t < lens)[:, None]
            nh0, nc0 = c0(emb[seq[t]], (h0, cc0))
            nh1, nc1 = c1(nh0, (h1, cc1))
            outs.append(torch.where(act, nh1, torch.zeros_like(nh1)))
            h0 = torch.where(act, nh0, h0); cc0 = torch.where(act, nc0, cc0)
            h1 = torch.where(act, nh1, h1); cc1 = torch.where(act, nc1, cc1)
    outs = torch.stack(outs).numpy()
    assert np.abs(outs - enc['outputs']).max() < 1e-12
    assert np.abs(h1.numpy() - enc['states'][1][1]).max() < 1e-12
    assert np.abs(cc0.numpy() - enc['states'][0][0]).max() < 1e-12
    eht = outs.reshape(-1, L) @ w[ENC + 'encoder_h_transform/weights'] + \
        w[ENC + 'encoder_h_transform/biases']
    assert np.abs(eht.reshape(enc['h_transformed'].shape) - enc['h_transformed']).max() < 1e-12


def test_decoder_step_quantities_against_torch():
    d = SMALL
    names = list(CLEVR_MODULE_NAMES)
    w = synth.make_weights(d, seed=4, dtype=np.float64)
    batch = synth.make_inputs(d, seed=4, n=d.N, min_len=1)
    P, Wv, bv = O.build_validity_mats(names)
    enc = O.encoder_forward(w, batch['input_seq_batch'], batch['seq_length_batch'], np.float64)
    dec = O.decoder_forward(w, enc, P, Wv, bv, d.T_decoder, np.float64)
    # re-derive step 0 attention / logits with torch ops
    L = d.lstm_dim
    base = DEC + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/'
    c0 = _tf_lstm_as_torch_cell(w[base % 0 + 'weights'], w[base % 0 + 'biases'], d.embed_dim_nmn, L)
    c1 = _tf_lstm_as_torch_cell(w[base % 1 + 'weights'], w[base % 1 + 'biases'], L, L)
    with torch.no_grad():
        x = T(w[DEC + 'go_embedding']).repeat(d.N, 1)
        (ec0, eh0), (ec1, eh1) = enc['states']
        h0, cc0 = c0(x, (T(eh0), T(ec0)))
        h1, cc1 = c1(h0, (T(eh1), T(ec1)))
        q = h1 @ T(w[DEC + 'att_prediction/weights']) + T(w[DEC + 'att_prediction/biases'])
        e = (torch.tanh(q[None] + T(enc['h_transformed'])) * T(w[DEC + 'att_prediction/v'])).sum(2)
        att = torch.softmax(e, dim=0) * T(enc['not_finished'][..., 0])
        att = att / att.sum(0, keepdim=True)
        ctx = (att[..., None] * T(enc['outputs'])).sum(0)
        sc = torch.cat([h1, ctx], 1) @ T(w[DEC + 'token_prediction/weights']) + \
            T(w[DEC + 'token_prediction/biases'])
    assert np.abs(att.numpy() - dec['atts'][0, :, :, 0]).max() < 1e-12
    assert np.abs(sc.numpy() - dec['token_scores'][0]).max() < 1e-12
    # first token is _Scene or _Find; every decoded layout is valid
    assert set(dec['predicted_tokens'][0]) <= {0, 1}
    assert O.assemble(names, dec['predicted_tokens'])[1].all()
    # word_vecs = attention-weighted input embeddings
    wv = np.einsum('dtn,tne->dne', dec['atts'][..., 0], enc['embedded'])
    assert np.abs(wv - dec['word_vecs']).max() < 1e-12


def _w(w, k):
    return T(w[MOD + k])


def test_module_operators_against_torch_ops():
    d = SMALL
    rng = np.random.default_rng(5)
    w = synth.make_weights(d, seed=5, dtype=np.float64)
    nb = 6
    feat = np.maximum(rng.standard_normal((nb, d.H, d.W, d.D)), 0)
    txt = rng.standard_normal((nb, d.embed_dim_txt)) * 0.3
    a0 = rng.standard_normal((nb, d.H, d.W, 1)) * 2
    a1 = rng.standard_normal((nb, d.H, d.W, 1)) * 2
    ft, tt, t0, t1 = T(feat), T(txt), T(a0), T(a1)

    def pool(att):
        p = torch.softmax(att.reshape(nb, -1), dim=1).reshape(nb, d.H, d.W, 1)
        return (ft * p).sum((1, 2))

    def lin(x, scope):
        return x @ _w(w, scope + '/weights') + _w(w, scope + '/biases')

    with torch.no_grad():
        # Find
        img = lin(ft, 'FindModule/conv_image')
        find = lin(F.normalize(img * lin(tt, 'FindModule/fc_text')[:, None, None], dim=3,
                               eps=1e-6), 'FindModule/conv_eltwise')
        # Transform: F.conv2d is NCHW cross-correlation
        K = _w(w, 'TransformModule/conv_maps/weights').permute(3, 2, 0, 1)
        maps = F.conv2d(t0.permute(0, 3, 1, 2), K, _w(w, 'TransformModule/conv_maps/biases'),
                        padding=d.kernel_size // 2).permute(0, 2, 3, 1)
        tr = lin(F.normalize(maps * lin(tt, 'TransformModule/text_fc')[:, None, None], dim=3,
                             eps=1e-6), 'TransformModule/conv_eltwise')
        # FindSameProperty
        s = 'FindSamePropertyModule/'
        fsp = lin(F.normalize(lin(ft, s + 'conv_image') * lin(tt, s + 'fc_text')[:, None, None] *
                              lin(pool(t0), s + 'fc_att')[:, None, None], dim=3, eps=1e-6),
                  s + 'conv_eltwise')
        # Describe / SameProperty
        s = 'DescribeModule/'
        desc = lin(F.normalize(lin(tt, s + 'fc_text') * lin(pool(t0), s + 'fc_att'), dim=1,
                               eps=1e-6), s + 'fc_eltwise')
        s = 'SamePropertyModule/'
        same = lin(F.normalize(lin(pool(t0), s + 'fc_att_0') * lin(tt, s + 'fc_text') *
                               lin(pool(t1), s + 'fc_att_1'), dim=1, eps=1e-6), s + 'fc_eltwise')
        f0 = t0.reshape(nb, -1); f1 = t1.reshape(nb, -1)
        exist = lin(torch.stack([f0.min(1).values, f0.mean(1), f0.max(1).values], 1),
                    'ExistModule/fc_scores')
        count = lin(torch.cat([f0, f0.min(1, True).values, f0.max(1, True).values], 1),
                    'CountModule/fc_scores')
        more = lin(torch.cat([f0, f0.min(1, True).values, f0.max(1, True).values,
                              f1, f1.min(1, True).values, f1.max(1, True).values], 1),
                   'MoreNumModule/fc_scores')
    w64 = O._cast(w, np.float64)
    checks = {
        'find': (find, O.m_find(w64, feat, txt)),
        'filter': (torch.minimum(t0, find), O.m_filter(w64, a0, feat, txt)),
        'transform': (tr, O.m_transform(w64, a0, txt)),
        'fsp': (fsp, O.m_find_same_property(w64, a0, feat, txt)),
        'describe': (desc, O.m_describe(w64, a0, feat, txt)),
        'same': (same, O.m_same_property(w64, a0, a1, feat, txt)),
        'exist': (exist, O.m_exist(w64, a0)),
        'count': (count, O.m_count(w64, a0)),
        'more': (more, O.m_more_num(w64, a0, a1)),
        'and': (torch.minimum(t0, t1), O.m_and(a0, a1)),
        'or': (torch.maximum(t0, t1), O.m_or(a0, a1)),
    }
    for k, (got, want) in checks.items():
        assert np.abs(got.numpy() - want).max() < 1e-10, k


def test_fp32_oracle_is_within_budget_of_fp64():
    """the fp32 run of the oracle (the timed CPU baseline) stays well inside the 1e-4 bar"""
    d = dataclasses.replace(Dims(), N=8)
    names = list(CLEVR_MODULE_NAMES)
    w = synth.make_weights(d, seed=0)
    batch = synth.make_inputs(d, seed=0, n=8)
    gt = synth.template_layout_batch(d, n=8)
    a = O.forward(w, names, batch, d.T_decoder, d.num_choices, np.float64, True, gt)
    b = O.forward(w, names, batch, d.T_decoder, d.num_choices, np.float32, True, gt)
    assert np.abs(a['scores'] - b['scores']).max() < 2e-5


def test_losses_match_torch_cross_entropy():
    rng = np.random.default_rng(0)
    sc = rng.standard_normal((9, 28)); lab = rng.integers(0, 28, 9)
    lsp = -np.abs(rng.standard_normal(9))
    out = O.losses({'x/weights': np.ones((3, 3))}, sc, lab, lsp)
    ce = F.cross_entropy(T(sc), torch.as_tensor(lab).long()).item()
    assert abs(out['avg_sample_loss'] - ce) < 1e-12
    assert abs(out['l2_reg'] - 4.5) < 1e-12
    assert abs(out['total_loss'] - (np.mean(-lsp) + ce + 5e-6 * 4.5)) < 1e-12
