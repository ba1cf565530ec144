"""The reference's exp_clevr/eval_clevr.py driving the HIP engine, as far as that can be arranged when the
reference checkout and the GPU never meet (missing #4): tests/golden/eval_driver_trace.npz
is the recording of what the UNMODIFIED script asked of the drop-in's Python face on the CPU box --
constructor keywords of NMN3Model, placeholders, every `sess.partial_run_setup` / `sess.partial_run` with
its feeds, the answers it wrote (tests/golden/make_eval_driver_trace.py; re-recorded and compared on every
CPU run by tests/test_reference_driver_source.py).  Here the same calls are issued, in the same order,
to the same drop-in classes -- NMN3Model builds its own HIP engine from the placeholder shapes exactly
as it does for the script -- and every value returned is compared with what the oracle returned to the
script."""
import json
import os

import numpy as np
import pytest

import eval_driver_common as EC
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims
from util import greedy_tokens_under_margin_rule

pytestmark = pytest.mark.gpu
TRACE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_driver_trace.npz')


def test_replay_of_the_reference_scripts_session_on_the_hip_engine():
    from oracle import n2nmn_oracle as O
    from n2nmn_amd.nmn3_assembler import Assembler, PackedLayouts
    from n2nmn_amd.nmn3_model import NMN3Model
    from n2nmn_amd.runtime import Session, placeholder
    z = np.load(TRACE)
    meta = json.loads(bytes(z['meta']))
    d = Dims()
    # ---- what the script built -----------------------------------------------------------------------
    ph = {k: placeholder(dt, shape) for k, (dt, shape) in meta['placeholders'].items()}
    asm = Assembler(list(synth.CLEVR_MODULE_NAMES))
    kw = dict(meta['model_kwargs'])
    assert kw.pop('assembler') == 'Assembler'
    model = NMN3Model(ph['image_feat_grid'], ph['text_seq_batch'], ph['seq_length_batch'], assembler=asm, **kw)
    ph['loom_input_tensor'] = model.compiler.loom_input_tensor
    w = synth.make_weights(d, seed=0)
    model.load_weights(w)                       # (the script: tf.train.Saver().restore of the same arrays)
    assert type(model.engine).__module__ == 'n2nmn_amd.engine'
    sess = Session()

    def holder(role):
        if role not in ph:                      # 'extra:<dtype>:<shape>': made by the script, never read by the model
            _, dt, shape = role.split(':')
            ph[role] = placeholder(dt, json.loads(shape.replace('None', 'null')))
        return ph[role]

    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    handles, answers, worst = {}, [], 0.0
    for k, call in enumerate(meta['calls']):
        hid = call['handle']
        if hid not in handles:
            st = meta['setups'][hid]
            handles[hid] = sess.partial_run_setup([getattr(model, f) for f in st['fetches']],
                                                  [holder(r) for r in st['feeds']])
        feeds = {}
        for role, f in call['feeds'].items():
            v = z[f['key']]
            if f['kind'] == 'image_ids':
                v = np.concatenate([EC.feature_of(int(i), d) for i in v], axis=0)
            elif f['kind'] == 'packed':
                v = PackedLayouts.from_nodes([(r[0], r[1], r[2], r[3], r[4], r[6]) for r in v.tolist()],
                                             f['num_rows'])
            feeds[holder(role)] = v
        got = sess.partial_run(handles[hid], getattr(model, call['fetch']), feed_dict=feeds)
        want = z['c%d_result' % k]
        assert got.shape == want.shape, (k, call['fetch'], got.shape, want.shape)
        if call['fetch'] == 'predicted_tokens':
            # the decoder margins of the oracle decide which positions may differ (near ties)
            batch = {'input_seq_batch': z[call['feeds']['text_seq_batch']['key']],
                     'seq_length_batch': z[call['feeds']['seq_length_batch']['key']]}
            enc = O.encoder_forward(w64, batch['input_seq_batch'], batch['seq_length_batch'], np.float64)
            dec = O.decoder_forward(w64, enc, asm.P, asm.W, asm.b, d.T_decoder, np.float64)
            assert np.array_equal(dec['predicted_tokens'], want)
            greedy_tokens_under_margin_rule(got, dec, 'eval_clevr.py batch %d' % hid)
            tokens_equal = np.array_equal(got, want)
        else:
            # phase 2 ran the program the SCRIPT assembled from the oracle's tokens on word_vecs of the HIP
            # phase 1: comparable when the HIP tokens are the oracle's (they are, bar near ties)
            assert tokens_equal, 'greedy layouts differ from the recording at a near tie: re-seed the scratch imdb'
            err = float(np.abs(got - want).max())
            worst = max(worst, err)
            assert err < 1e-4, (k, err)
            answers += list(np.argmax(got, axis=1))
    # ---- the answers the script wrote -------------------------------------------------------------------
    assert len(answers) == EC.N_QUESTIONS
    wrote = z['answers_written']
    scores = np.concatenate([z['c%d_result' % k] for k, c in enumerate(meta['calls']) if c['fetch'] == 'scores'])
    top2 = np.sort(scores, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 2e-4          # rows whose arg max cannot move within the tolerance
    assert np.array_equal(np.asarray(answers)[decided], wrote[decided])
    assert decided.mean() > 0.9


# ---- exp_vqa/eval_vqa2.py and exp_shapes/eval_shapes.py (VERDICT r4 "missing" #3) ---------------------------
def _replay(z, model, ph, feature_fn, on_tokens, tol=1e-4):
    """issue the recorded partial_run calls to `model` over the HIP engine; returns (scores per phase-2 call,
    worst |diff| against the recorded values)"""
    from n2nmn_amd.nmn3_assembler import PackedLayouts
    from n2nmn_amd.runtime import Session, placeholder
    meta = json.loads(bytes(z['meta']))
    sess = Session()

    def holder(role):
        if role not in ph:
            _, dt, shape = role.split(':')
            ph[role] = placeholder(dt, json.loads(shape.replace('None', 'null')))
        return ph[role]

    handles, scores, worst, tokens_equal = {}, [], 0.0, True
    for k, call in enumerate(meta['calls']):
        hid = call['handle']
        if hid not in handles:
            st = meta['setups'][hid]
            handles[hid] = sess.partial_run_setup([getattr(model, f) for f in st['fetches']],
                                                  [holder(r) for r in st['feeds']])
        feeds = {}
        for role, f in call['feeds'].items():
            v = z[f['key']]
            if f['kind'] == 'image_ids':
                v = np.concatenate([feature_fn(int(i)) for i in v], axis=0)
            elif f['kind'] == 'packed':
                v = PackedLayouts.from_nodes([(r[0], r[1], r[2], r[3], r[4], r[6]) for r in v.tolist()],
                                             f['num_rows'])
            feeds[holder(role)] = v
        got = sess.partial_run(handles[hid], getattr(model, call['fetch']), feed_dict=feeds)
        want = z['c%d_result' % k]
        assert got.shape == want.shape, (k, call['fetch'], got.shape, want.shape)
        if call['fetch'] == 'predicted_tokens':
            tokens_equal = on_tokens(call, got, want)
        else:
            assert tokens_equal, 'greedy layouts differ from the recording at a near tie: re-seed the scratch data'
            err = float(np.abs(got - want).max())
            worst = max(worst, err)
            assert err < tol, (k, err)
            scores.append(got)
    return scores, worst


def test_replay_of_eval_vqa2_session_on_the_hip_engine():
    """tests/golden/eval_driver_trace_vqa2.npz: what the UNMODIFIED exp_vqa/eval_vqa2.py asked of
    n2nmn_amd.models_vqa (constructor keywords incl. use_qpn / qpn_dropout / reduce_visfeat_dim, two batches of
    50 + 3 questions at 14 x 14 x 2048, lstm_dim 1000, 17 742 words, 3 001 answers), issued to the same classes
    over the HIP engine; then the script's own `scores_val[:, 0] = -1e10` (:137) and arg max against the answers
    it wrote."""
    import eval_driver_more as EM
    from oracle import n2nmn_oracle as O
    from n2nmn_amd import models_vqa
    from n2nmn_amd.runtime import placeholder
    z = np.load(os.path.join(os.path.dirname(TRACE), 'eval_driver_trace_vqa2.npz'))
    meta = json.loads(bytes(z['meta']))
    ph = {k: placeholder(dt, shape) for k, (dt, shape) in meta['placeholders'].items()}
    asm = models_vqa.Assembler(list(models_vqa.VQA_MODULE_NAMES))
    kw = dict(meta['model_kwargs'])
    assert kw.pop('assembler') == 'Assembler' and kw['use_qpn'] is True and kw['reduce_visfeat_dim'] is False
    model = models_vqa.NMN3Model(ph['image_feat_grid'], ph['text_seq_batch'], ph['seq_length_batch'],
                                 assembler=asm, **kw)
    ph['loom_input_tensor'] = model.compiler.loom_input_tensor
    w = EM.vqa_weights()
    model.load_weights(w)
    assert type(model.engine).__module__ == 'n2nmn_amd.engine'
    w64 = {k: v.astype(np.float64) for k, v in w.items()}

    def on_tokens(call, got, want):
        seq, lens = z[call['feeds']['text_seq_batch']['key']], z[call['feeds']['seq_length_batch']['key']]
        enc = O.encoder_forward(w64, seq, lens, np.float64)
        dec = O.decoder_forward(w64, enc, asm.P, asm.W, asm.b, 13, np.float64)
        assert np.array_equal(dec['predicted_tokens'], want)
        greedy_tokens_under_margin_rule(got, dec, 'eval_vqa2.py')
        return np.array_equal(got, want)

    scores, worst = _replay(z, model, ph, EM.vqa_feature_of, on_tokens)
    print('worst |HIP - recorded| over %d phase-2 calls: %.2e' % (len(scores), worst))
    sc = np.concatenate(scores)
    rec = np.concatenate([z['c%d_result' % k] for k, c in enumerate(meta['calls']) if c['fetch'] == 'scores'])
    assert sc.shape == (EM.VQA_N, 3001)
    sc[:, 0] = -1e10                                   # exp_vqa/eval_vqa2.py:137: remove the <unk> answer
    rec = rec.copy()
    rec[:, 0] = -1e10
    top2 = np.sort(rec, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 2e-4         # rows whose arg max cannot move within the tolerance
    assert np.array_equal(np.argmax(sc, axis=1)[decided], z['answers_written'][decided])
    assert decided.mean() > 0.9 and not (np.argmax(sc, axis=1) == 0).any()


def test_replay_of_eval_shapes_session_on_the_hip_engine():
    """tests/golden/eval_driver_trace_shapes.npz: the UNMODIFIED exp_shapes/eval_shapes.py on the reference's own
    `train.tiny` split (64 images), issued to n2nmn_amd.models_shapes.NMN3ModelAtt over the HIP engine: the
    convnet (two GEMMs through n2nmn_fc_forward), the automaton-free decoder with the <eos> latch, the modules
    at SHAPES dimensions."""
    import eval_driver_more as EM
    from oracle import n2nmn_oracle as O
    from oracle import n2nmn_oracle_shapes as S
    from n2nmn_amd import models_shapes
    from n2nmn_amd.runtime import placeholder
    z = np.load(os.path.join(os.path.dirname(TRACE), 'eval_driver_trace_shapes.npz'))
    meta = json.loads(bytes(z['meta']))
    ph = {k: placeholder(dt, shape) for k, (dt, shape) in meta['placeholders'].items()}
    kw = dict(meta['model_kwargs'])
    model = models_shapes.NMN3ModelAtt(ph['image_feat_grid'], ph['text_seq_batch'], ph['seq_length_batch'], **kw)
    ph['loom_input_tensor'] = model.compiler.loom_input_tensor
    # (the fitted weights need the reference's data files: rebuilt from the recording's own feeds instead)
    w = _shapes_weights_from_trace(z, meta, S, O)
    model.load_weights(w)
    assert type(model.engine).__module__ == 'n2nmn_amd.engine'
    w64 = {k: v.astype(np.float64) for k, v in w.items()}

    def on_tokens(call, got, want):
        seq, lens = z[call['feeds']['text_seq_batch']['key']], z[call['feeds']['seq_length_batch']['key']]
        enc = O.encoder_forward(w64, seq, lens, np.float64)
        dec = S.decoder_forward(w64, enc, 11, 4, np.float64)
        assert np.array_equal(dec['predicted_tokens'], want)
        sc = dec['token_scores']
        top2 = np.sort(sc, axis=2)[:, :, -2:]
        near = (top2[:, :, 1] - top2[:, :, 0]) < 1e-3
        assert np.array_equal(got[~near], want[~near])
        # the convnet of the face against the oracle's, on the images the script fed
        img = z[call['feeds']['image_feat_grid']['key']]
        feat = model.convnet(img).cpu().numpy()
        assert np.abs(feat - S.shapes_convnet(w64, img.astype(np.float64))).max() < 1e-4
        return np.array_equal(got, want)

    scores, worst = _replay(z, model, ph, None, on_tokens)
    print('worst |HIP - recorded|: %.2e' % worst)
    sc = np.concatenate(scores)
    top2 = np.sort(np.concatenate([z['c%d_result' % k] for k, c in enumerate(meta['calls'])
                                   if c['fetch'] == 'scores']), axis=1)
    decided = (top2[:, 1] - top2[:, 0]) > 2e-4
    assert np.array_equal(np.argmax(sc, axis=1)[decided], z['answers_written'][decided])
    assert decided.mean() > 0.9


def _shapes_weights_from_trace(z, meta, S, O):
    """eval_driver_more.shapes_weights() without the reference checkout: the same seeded weights, the token
    classifier fitted on the questions of the recording (its phase-1 feeds) and the layouts the recorded run
    decoded -- which ARE the data set's ground-truth layouts (the script reported layout accuracy 1.0)"""
    call = [c for c in meta['calls'] if c['fetch'] == 'predicted_tokens'][0]
    k = meta['calls'].index(call)
    seq, lens = z[call['feeds']['text_seq_batch']['key']], z[call['feeds']['seq_length_batch']['key']]
    gt = z['c%d_result' % k]
    assert b'layout accuracy = 1.0' in bytes(z['summary'])
    w = synth.make_weights_from_shapes(S.variable_shapes(14, 5), seed=0)
    enc = O.encoder_forward(w, seq, lens, np.float64)
    dec = S.decoder_forward(w, enc, 11, 4, np.float64, True, gt)
    X = dec['token_features'].reshape(-1, dec['token_features'].shape[-1])
    X1 = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1)
    sol = np.linalg.lstsq(X1, 8.0 * np.eye(5)[gt.reshape(-1)], rcond=1e-6)[0]
    w[O._DEC + 'token_prediction/weights'] = sol[:-1].astype(np.float32)
    w[O._DEC + 'token_prediction/biases'] = sol[-1].astype(np.float32)
    return w
