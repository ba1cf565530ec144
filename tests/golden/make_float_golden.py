#!/usr/bin/env python3
"""Generate tests/golden/float_golden.npz by RUNNING THE REFERENCE'S OWN MODEL CODE.

The reference (ronghanghu/n2nmn) is TensorFlow-1.0.0 / TensorFlow-Fold graph-building python; neither
package can be installed here.  oracle/tf1_stub/ provides `tensorflow` and `tensorflow_fold` modules
that execute every op eagerly on torch CPU tensors in float64, so the UNMODIFIED files

    models_clevr/nmn3_model.py, nmn3_netgen_att.py, nmn3_modules.py, nmn3_assembler.py
    models_vqa/nmn3_model.py, nmn3_modules.py, question_prior_net.py (+ its nmn3_netgen_att.py)
    models_shapes/nmn3_model.py, nmn3_modules.py, nmn3_netgen_att.py, nmn3_assembler.py, shapes_convnet.py
    util/cnn.py, util/empty_safe_conv.py
    the loss blocks of exp_clevr/train_clevr_gt_layout.py and train_clevr_rl_gt_layout.py
      (their source lines are exec'd from the checkout, not restated)

compute the numbers this fixture holds: NMN3Model is constructed on the seeded inputs of
tests/golden/float_cases.py, `Assembler.assemble` turns its predicted tokens into expression trees,
and the Fold stub evaluates `compiler.output_tensors[0]` with per-(operator, depth) batching.
Gradients are torch autograd through that same reference code.

Only works in the build container (needs /root/reference).  No bytecode is written into the
reference tree.  Usage:  python tests/golden/make_float_golden.py [--check]
(--check regenerates in memory and compares with the committed file instead of writing it).
"""
import ast
import json
import os
import re
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
OUT = os.path.join(HERE, 'float_golden.npz')


def setup_reference_imports():
    """stub tensorflow first on sys.path, then the reference checkout, then this repo."""
    if not hasattr(np, 'bool'):
        np.bool = bool                                   # models_clevr/nmn3_assembler.py:221
    for p in (ROOT, REF, os.path.join(ROOT, 'oracle', 'tf1_stub')):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import tensorflow as tf
    assert 'tf1_stub' in tf.__file__
    return tf


tf = setup_reference_imports()
import torch                                              # noqa: E402
import tensorflow_fold as td                              # noqa: E402,F401
sys.path.insert(0, HERE)
import float_cases as FC                                  # noqa: E402


def T64(x):
    x = np.asarray(x)
    return torch.as_tensor(x.astype(np.float64) if x.dtype.kind == 'f' else x)


def n(x):
    return x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def fresh_graph(weights, none_dim, requires_grad=False):
    tf.reset_default_graph()
    tf.config.float_dtype = torch.float64
    tf.config.preloaded = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    tf.config.strict_preload = True
    tf.config.none_dim = none_dim
    tf.config.requires_grad = requires_grad
    tf.config.multinomial_uniforms = None


def exec_loss_block(script, env, first='compiler = nmn3_model_trn.compiler'):
    """exec the reference's loss / optimiser lines (from `compiler = nmn3_model_trn.compiler` to
    `solver_op = solver.apply_gradients(gradients)`) with the script's own training constants."""
    lines = open(os.path.join(REF, script)).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(first))
    end = next(i for i, l in enumerate(lines) if l.startswith('solver_op = solver.apply_gradients'))
    for l in lines[:start]:                              # module-level literal constants
        m = re.match(r'^(\w+)\s*=\s*([^#]+?)\s*(#.*)?$', l)
        if m:
            try:
                env.setdefault(m.group(1), ast.literal_eval(m.group(2)))
            except (ValueError, SyntaxError):
                pass
    code = '\n'.join(lines[start:end + 1])
    exec(compile(code, os.path.join(REF, script), 'exec'), env)
    return env


def build_clevr(case, sampling=False, use_gt=None, gt=None, grad=False, uniforms=None):
    from models_clevr.nmn3_assembler import Assembler
    from models_clevr.nmn3_model import NMN3Model
    d, batch = FC.clevr_inputs(case)
    fresh_graph(FC.clevr_weights(), d.N, requires_grad=grad)
    if uniforms is not None:
        rows = iter(uniforms)
        tf.config.multinomial_uniforms = lambda nrows: next(rows)
    asm = Assembler(os.path.join(REF, 'exp_clevr/data/vocabulary_layout.txt'))
    kw = {}
    if use_gt is not None:
        kw = dict(use_gt_layout=torch.tensor(bool(use_gt)), gt_layout_batch=torch.as_tensor(gt))
    model = NMN3Model(T64(batch['image_feat_batch']), torch.as_tensor(batch['input_seq_batch']),
                      torch.as_tensor(batch['seq_length_batch']), T_decoder=d.T_decoder,
                      num_vocab_txt=d.num_vocab_txt, embed_dim_txt=d.embed_dim_txt,
                      num_vocab_nmn=d.num_vocab_nmn, embed_dim_nmn=d.embed_dim_nmn,
                      lstm_dim=d.lstm_dim, num_layers=d.num_layers, assembler=asm,
                      encoder_dropout=False, decoder_dropout=False, decoder_sampling=sampling,
                      num_choices=d.num_choices, **kw)
    tokens = n(model.predicted_tokens)
    expr_list, validity = asm.assemble(tokens)
    scores = tf.Session().run(model.scores, feed_dict=model.compiler.build_feed_dict(expr_list))
    return d, batch, asm, model, expr_list, np.asarray(validity, bool), scores


def seq2seq_outputs(out, key, model, full=False):
    s = model.att_seq2seq
    out[key + '/predicted_tokens'] = n(model.predicted_tokens).astype(np.int32)
    out[key + '/token_probs'] = n(model.token_probs)
    out[key + '/neg_entropy'] = n(model.neg_entropy)
    out[key + '/word_vecs'] = n(model.word_vecs)
    out[key + '/atts'] = n(model.atts)
    out[key + '/log_seq_prob'] = n(model.log_seq_prob)
    for l, st in enumerate(s.encoder_states):
        out[key + '/encoder_state_c%d' % l] = n(st.c)
        out[key + '/encoder_state_h%d' % l] = n(st.h)
    if full:
        out[key + '/encoder_outputs'] = n(s.encoder_outputs)
        out[key + '/encoder_h_transformed'] = n(s.encoder_h_transformed)


def probes(out, key, named):
    meta = {}
    for name, t in named.items():
        p = FC.probe(name, n(t))
        out[key + '/' + name] = p.pop('values')
        meta[name] = p
    return meta


def case_greedy(out, meta):
    d, batch, asm, model, exprs, validity, scores = build_clevr('greedy')
    seq2seq_outputs(out, 'greedy', model, full=True)
    out['greedy/scores'] = n(scores)
    out['greedy/validity'] = validity
    meta['greedy'] = dict(fold_batches=model.compiler.batch_sizes,
                          entropy_reg=float(model.entropy_reg.detach()),
                          l2_reg=float(model.l2_reg.detach()),
                          variables=sorted(v.op.name for v in tf.trainable_variables()))


def case_gt(out, meta):
    d0 = FC.clevr_dims('gt')
    gt = FC.gt_layouts(d0)
    d, batch, asm, model, exprs, validity, scores = build_clevr('gt', use_gt=True, gt=gt, grad=True)
    assert validity.all() and np.array_equal(n(model.predicted_tokens), gt)
    seq2seq_outputs(out, 'gt', model)
    out['gt/scores'] = n(scores)
    proxy = types.SimpleNamespace(compiler=model.compiler, scores=scores,
                                  log_seq_prob=model.log_seq_prob, l2_reg=model.l2_reg,
                                  entropy_reg=model.entropy_reg)
    env = exec_loss_block('exp_clevr/train_clevr_gt_layout.py', dict(
        tf=tf, nmn3_model_trn=proxy, answer_label_batch=torch.as_tensor(batch['answer_label_batch'])))
    solver = env['solver']
    m = dict(total_loss=float(env['total_loss'].detach()),
             avg_sample_loss=float(env['avg_sample_loss'].detach()),
             seq_likelihood_loss=float(env['seq_likelihood_loss'].detach()),
             l2_reg=float(model.l2_reg.detach()),
             weight_decay=env['weight_decay'], max_grad_l2_norm=env['max_grad_l2_norm'],
             fold_batches=model.compiler.batch_sizes)
    m['grad'] = probes(out, 'gt/grad', {v.op.name: g for g, v in solver.last_raw_gradients})
    m['clipped'] = probes(out, 'gt/clipped', {v.op.name: g for g, v in env['gradients']})
    env['solver_op'].run()                               # one Adam step (TF 1.0.0 defaults)
    m['adam_w1'] = probes(out, 'gt/adam_w1', {v.op.name: v for v in tf.trainable_variables()})
    meta['gt'] = m


def _rl_block(out, meta_key, key, model, scores, batch, validity_in):
    proxy = types.SimpleNamespace(compiler=model.compiler, scores=scores,
                                  log_seq_prob=model.log_seq_prob, l2_reg=model.l2_reg,
                                  entropy_reg=model.entropy_reg)
    env = exec_loss_block('exp_clevr/train_clevr_rl_gt_layout.py', dict(
        tf=tf, nmn3_model_trn=proxy, answer_label_batch=torch.as_tensor(batch['answer_label_batch']),
        expr_validity_batch=torch.as_tensor(validity_in)))
    solver = env['solver']
    m = dict(total_loss=float(env['total_loss'].detach()),
             avg_sample_loss=float(env['avg_sample_loss'].detach()),
             policy_gradient_loss=float(env['policy_gradient_loss'].detach()),
             entropy_reg=float(model.entropy_reg.detach()), l2_reg=float(model.l2_reg.detach()),
             baseline_before=float(env['baseline']), invalid_expr_loss=env['invalid_expr_loss'],
             lambda_entropy=env['lambda_entropy'], baseline_decay=env['baseline_decay'],
             weight_decay=env['weight_decay'], finetune_lr=env['finetune_lr'])
    out[key + '/validity_in'] = validity_in
    m['grad'] = probes(out, key + '/grad', {v.op.name: g for g, v in solver.last_raw_gradients})
    env['baseline_update_op'].run()                      # graph-mode side effect: after the read
    m['baseline_after'] = float(env['baseline'])
    return m


def case_sampled(out, meta):
    d0 = FC.clevr_dims('sampled')
    u = FC.sample_uniforms(d0)
    for key in ('sampled', 'sampled_inv'):
        d, batch, asm, model, exprs, validity, scores = build_clevr(
            'sampled', sampling=True, grad=True, uniforms=u)
        validity_in = validity.copy()
        if key == 'sampled_inv':
            # the automaton never emits an invalid layout, so to run the reference's
            # invalid_expr_loss branch the validity INPUT of the loss block is overridden for rows
            # 1 and 4 (expr_validity_batch is a placeholder in the reference)
            validity_in[[1, 4]] = False
        else:
            seq2seq_outputs(out, key, model)
            out[key + '/scores'] = n(scores)
            out[key + '/validity'] = validity
        meta[key] = _rl_block(out, meta, key, model, scores, batch, validity_in)


def case_modules(out, meta):
    from models_clevr.nmn3_modules import Modules
    d, x = FC.module_inputs()
    fresh_graph({k: v for k, v in FC.clevr_weights().items() if '/module_variables/' in k}, d.N)
    with tf.variable_scope('neural_module_network'):
        with tf.variable_scope('layout_execution'):
            mods = Modules(T64(x['image_feat']), T64(x['word_vecs']), d.num_choices)
            ti, bi = torch.as_tensor(x['time_idx']), torch.as_tensor(x['batch_idx'])
            for name, nin in FC.MODULE_CALLS:
                args = [T64(x['input_0']), T64(x['input_1'])][:nin]
                out['modules/' + name] = n(getattr(mods, name)(*args, ti, bi))
    meta['modules'] = dict(variables=sorted(v.op.name for v in tf.trainable_variables()))


def case_vqa(out, meta):
    from models_vqa.nmn3_assembler import Assembler
    from models_vqa.nmn3_model import NMN3Model
    d, batch, gt = FC.vqa_setup()
    for mode in ('greedy', 'gt'):
        fresh_graph(FC.vqa_weights(d), d.N)
        asm = Assembler(os.path.join(REF, 'exp_vqa/data/vocabulary_layout.txt'))
        kw = dict(use_gt_layout=torch.tensor(True), gt_layout_batch=torch.as_tensor(gt)) \
            if mode == 'gt' else {}
        model = NMN3Model(T64(batch['image_feat_batch']), torch.as_tensor(batch['input_seq_batch']),
                          torch.as_tensor(batch['seq_length_batch']), T_decoder=d.T_decoder,
                          num_vocab_txt=d.num_vocab_txt, embed_dim_txt=d.embed_dim_txt,
                          num_vocab_nmn=d.num_vocab_nmn, embed_dim_nmn=d.embed_dim_nmn,
                          lstm_dim=d.lstm_dim, num_layers=d.num_layers, assembler=asm,
                          encoder_dropout=False, decoder_dropout=False, decoder_sampling=False,
                          num_choices=d.num_choices, use_qpn=True, qpn_dropout=False, **kw)
        key = 'vqa_' + mode
        exprs, validity = asm.assemble(n(model.predicted_tokens))
        scores = tf.Session().run(model.scores, feed_dict=model.compiler.build_feed_dict(exprs))
        seq2seq_outputs(out, key, model)
        out[key + '/scores'] = n(scores)
        out[key + '/validity'] = np.asarray(validity, bool)
        meta[key] = dict(fold_batches=model.compiler.batch_sizes,
                         variables=sorted(v.op.name for v in tf.trainable_variables()))


def case_vqa_train(out, meta):
    """models_vqa with encoder / decoder / question-prior dropout and the loss block of
    exp_vqa/train_vqa_gt_layout.py:113-131 (no clipping, weight_decay 0, Adam defaults)."""
    from models_vqa.nmn3_assembler import Assembler
    from models_vqa.nmn3_model import NMN3Model
    d, batch, gt = FC.vqa_setup()
    masks = FC.vqa_dropout_masks(d)
    labels = FC.vqa_labels(d)
    fresh_graph(FC.vqa_weights(d), d.N, requires_grad=True)
    # the masks are handed out in the order the graph asks for them: T_enc encoder steps, T_dec
    # decoder steps (DropoutWrapper on layer 0), then the two dropouts of question_prior_net
    order = [masks['enc0'][t] for t in range(d.T_encoder)] + \
            [masks['dec0'][t] for t in range(d.T_decoder)] + [masks['qpn_h'], masks['qpn_fc1']]
    it = iter(order)

    def next_mask(shape, keep_prob):
        m = next(it)
        assert tuple(shape) == m.shape and keep_prob == 0.5, (shape, m.shape, keep_prob)
        return m
    tf.config.dropout_masks = next_mask
    asm = Assembler(os.path.join(REF, 'exp_vqa/data/vocabulary_layout.txt'))
    model = NMN3Model(T64(batch['image_feat_batch']), torch.as_tensor(batch['input_seq_batch']),
                      torch.as_tensor(batch['seq_length_batch']), T_decoder=d.T_decoder,
                      num_vocab_txt=d.num_vocab_txt, embed_dim_txt=d.embed_dim_txt,
                      num_vocab_nmn=d.num_vocab_nmn, embed_dim_nmn=d.embed_dim_nmn,
                      lstm_dim=d.lstm_dim, num_layers=d.num_layers, assembler=asm,
                      encoder_dropout=True, decoder_dropout=True, decoder_sampling=False,
                      num_choices=d.num_choices, use_qpn=True, qpn_dropout=True,
                      use_gt_layout=torch.tensor(True), gt_layout_batch=torch.as_tensor(gt))
    assert next(it, None) is None, 'not every dropout mask was consumed'
    tf.config.dropout_masks = None
    exprs, validity = asm.assemble(n(model.predicted_tokens))
    assert np.all(validity) and np.array_equal(n(model.predicted_tokens), gt)
    scores = tf.Session().run(model.scores, feed_dict=model.compiler.build_feed_dict(exprs))
    key = 'vqa_train'
    seq2seq_outputs(out, key, model)
    out[key + '/scores'] = n(scores)
    proxy = types.SimpleNamespace(compiler=model.compiler, scores=scores,
                                  log_seq_prob=model.log_seq_prob, l2_reg=model.l2_reg,
                                  entropy_reg=model.entropy_reg)
    env = exec_loss_block('exp_vqa/train_vqa_gt_layout.py', dict(
        tf=tf, nmn3_model_trn=proxy, answer_label_batch=torch.as_tensor(labels)),
        first='softmax_loss_per_sample = ')
    solver = env['solver']
    m = dict(total_loss=float(env['total_loss'].detach()),
             avg_sample_loss=float(env['avg_sample_loss'].detach()),
             seq_likelihood_loss=float(env['seq_likelihood_loss'].detach()),
             weight_decay=env['weight_decay'], fold_batches=model.compiler.batch_sizes,
             variables=sorted(v.op.name for v in tf.trainable_variables()))
    m['grad'] = probes(out, key + '/grad', {v.op.name: g for g, v in solver.last_raw_gradients})
    env['solver_op'].run()                               # one Adam step (TF 1.0.0 defaults)
    m['adam_w1'] = probes(out, key + '/adam_w1', {v.op.name: v for v in tf.trainable_variables()})
    meta[key] = m


def case_shapes(out, meta):
    """models_shapes (BASELINE.json configs[0]): NMN3ModelAtt = shapes_convnet + the SHAPES layout
    generator (no validity automaton, <eos> latch) + Find / Transform / And / Answer, on the 12
    dataset questions of shapes_golden.json, free-running and with the ground-truth layouts."""
    from models_shapes.nmn3_assembler import Assembler
    from models_shapes.nmn3_model import NMN3ModelAtt
    d, batch, gt, w, nv_txt, nv_nmn = FC.shapes_setup()
    N = batch['image_batch'].shape[0]
    for mode in ('greedy', 'gt'):
        fresh_graph(w, N)
        asm = Assembler(os.path.join(REF, 'exp_shapes/data/vocabulary_layout.txt'))
        kw = dict(use_gt_layout=torch.tensor(True), gt_layout_batch=torch.as_tensor(gt)) \
            if mode == 'gt' else {}
        model = NMN3ModelAtt(T64(batch['image_batch']), torch.as_tensor(batch['text_seq_batch']),
                             torch.as_tensor(batch['seq_length_batch']), T_decoder=d['T_decoder'],
                             num_vocab_txt=nv_txt, embed_dim_txt=d['embed_dim_txt'],
                             num_vocab_nmn=nv_nmn, embed_dim_nmn=d['embed_dim_nmn'],
                             lstm_dim=d['lstm_dim'], num_layers=2, EOS_idx=asm.EOS_idx,
                             encoder_dropout=False, decoder_dropout=False, decoder_sampling=False,
                             num_choices=d['num_choices'], **kw)
        key = 'shapes_' + mode
        exprs, validity = asm.assemble(n(model.predicted_tokens))
        scores = tf.Session().run(model.scores, feed_dict=model.compiler.build_feed_dict(exprs))
        seq2seq_outputs(out, key, model)
        out[key + '/image_feat_grid'] = n(model.image_feat_grid)
        out[key + '/scores'] = n(scores)
        out[key + '/validity'] = np.asarray(validity, bool)
        meta[key] = dict(fold_batches=model.compiler.batch_sizes,
                         variables=sorted(v.op.name for v in tf.trainable_variables()))


def generate():
    out, meta = {}, {}
    for fn in (case_greedy, case_gt, case_sampled, case_modules, case_vqa, case_vqa_train, case_shapes):
        fn(out, meta)
        print('%-14s done (%d arrays so far)' % (fn.__name__, len(out)), flush=True)
    out['meta_json'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    return out


def main():
    out = generate()
    if '--check' in sys.argv:
        old = np.load(OUT)
        assert sorted(old.files) == sorted(out), 'fixture keys changed'
        worst = 0.0
        for k in out:
            if k == 'meta_json':
                continue
            a, b = np.asarray(old[k], np.float64), np.asarray(out[k], np.float64)
            worst = max(worst, float(np.max(np.abs(a - b))) if a.size else 0.0)
        print('committed fixture vs regenerated: max |diff| = %.3e' % worst)
        assert worst <= 1e-12
        return
    np.savez_compressed(OUT, **out)
    print('wrote %s (%.1f KB)' % (OUT, os.path.getsize(OUT) / 1024))


if __name__ == '__main__':
    main()
