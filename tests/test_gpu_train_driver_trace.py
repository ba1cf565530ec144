"""The reference's TRAINING drivers driving the HIP engine and the HIP Trainer, as far as that can be arranged
when the reference checkout and the GPU never meet.

tests/golden/train_driver_trace_{gt,rl,scratch}.npz record what exp_clevr/train_clevr_gt_layout.py,
exp_clevr/train_clevr_rl_gt_layout.py and exp_clevr/train_clevr_scratch.py, executed UNMODIFIED on the CPU box, asked of the drop-in and got back from
the fp64 oracle doubles (tests/golden/make_train_driver_trace.py; re-recorded and compared on every CPU run by
tests/test_reference_train_driver_source.py).  Here the scripts' graph-building statements are issued to the same
`n2nmn_amd.runtime.tf` names in the same order (train_clevr_gt_layout.py:84-130 / rl :82-132 -- the lines are cited
beside each statement), NMN3Model builds its own HIP engine, the fetched `train_step` builds the HIP Trainer from the
matched loss graph, and the recorded batches are replayed through the scripts' two partial_run calls per iteration.
Every value returned is compared with what the oracle returned to the script; after the last iteration every
variable is compared with the oracle's.

Tolerances: iteration 1 is a pure forward of the initial weights (1e-4, the logit bar).  From iteration 2 on the
weights have gone through Adam steps computed in fp32 here and in fp64 there; the first steps of Adam move every
element by ~lr whatever the size of its gradient (m / sqrt(v) = +-1), so elements whose gradient is of the order
of its fp32 round-off may step the other way: the trajectories stay within a few lr of each other, not within
1e-4.  The bars below (scores 5e-3, weights 21 * lr * 0.2) were set from the first measured run and are
stated in the assertions."""
import json
import os

import numpy as np
import pytest

import eval_driver_common as EC
import train_driver_common as TC
from n2nmn_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, json.loads(bytes(z['meta']))


def _features(ids, d):
    return np.concatenate([EC.feature_of(int(i), d) for i in ids], axis=0)


def _probes(z, weights):
    import sys
    sys.path.insert(0, GOLD)
    import make_train_driver_trace as MT
    worst = {}
    for k in z.files:
        if not k.startswith('w_'):
            continue
        name = k[2:]
        flat = np.asarray(weights[name].detach().cpu().numpy() if hasattr(weights[name], 'detach') else weights[name],
                          np.float64).reshape(-1)
        worst[name] = float(np.abs(flat[MT.probe_indices(name, flat.size)] - z[k]).max())
    return worst


def test_replay_of_train_clevr_gt_layout_on_the_hip_trainer(tmp_path):
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.nmn3_model import NMN3Model
    from n2nmn_amd.runtime import tf
    from n2nmn_amd import runtime, runtime_train
    z, meta = _load('train_driver_trace_gt.npz')
    d = TC.train_dims()
    runtime._MODELS.clear()
    runtime_train._GLOBALS.clear()
    kw = dict(meta['model_kwargs'])
    assert kw.pop('assembler') == 'Assembler' and kw.pop('use_gt_layout') == 'Const' and \
        kw.pop('gt_layout_batch') == 'Placeholder'
    # ---- the script's graph, statement by statement (train_clevr_gt_layout.py) --------------------------------
    sess = tf.Session(config=tf.ConfigProto(gpu_options=tf.GPUOptions(allow_growth=True)))            # :15-17
    assembler = Assembler(list(synth.CLEVR_MODULE_NAMES))                                             # :60
    input_seq_batch = tf.placeholder(tf.int32, [None, None])                                          # :76
    seq_length_batch = tf.placeholder(tf.int32, [None])
    image_feat_batch = tf.placeholder(tf.float32, [None, d.H, d.W, d.D])
    expr_validity_batch = tf.placeholder(tf.bool, [None])
    answer_label_batch = tf.placeholder(tf.int32, [None])
    use_gt_layout = tf.constant(True, dtype=tf.bool)
    gt_layout_batch = tf.placeholder(tf.int32, [None, None])                                          # :82
    model = NMN3Model(image_feat_batch, input_seq_batch, seq_length_batch, assembler=assembler,
                      use_gt_layout=use_gt_layout, gt_layout_batch=gt_layout_batch, **kw)              # :85-97
    compiler, scores, log_seq_prob = model.compiler, model.scores, model.log_seq_prob                 # :99-101
    softmax_loss_per_sample = tf.nn.sparse_softmax_cross_entropy_with_logits(logits=scores, labels=answer_label_batch)
    avg_sample_loss = tf.reduce_mean(softmax_loss_per_sample)                                         # :110
    seq_likelihood_loss = tf.reduce_mean(-log_seq_prob)                                               # :111
    total_training_loss = seq_likelihood_loss + avg_sample_loss
    total_loss = total_training_loss + meta['weight_decay'] * model.l2_reg                            # :114
    solver = tf.train.AdamOptimizer()                                                                 # :117
    gradients = solver.compute_gradients(total_loss)
    gradients = [(tf.clip_by_norm(g, meta['hyper']['max_grad_l2_norm']), v) for g, v in gradients]    # :122-123
    solver_op = solver.apply_gradients(gradients)
    with tf.control_dependencies([solver_op]):                                                        # :129-130
        train_step = tf.constant(0)
    log_writer = tf.summary.FileWriter(str(tmp_path / 'tb'), tf.get_default_graph())                  # :134
    loss_ph = tf.placeholder(tf.float32, [])
    log_step_trn = tf.summary.merge([tf.summary.scalar('avg_sample_loss', loss_ph)])
    snapshot_saver = tf.train.Saver(max_to_keep=None)                                                 # :159
    sess.run(tf.global_variables_initializer())                                                       # :160
    model.load_weights(synth.make_weights(d, seed=3))           # (the recording started from these: run_train_script)
    assert type(model.engine).__module__ == 'n2nmn_amd.engine'
    n_iter = len(meta['iterations'])
    worst = dict(scores=0.0, loss=0.0, entropy=0.0)
    for i in range(n_iter):
        h = sess.partial_run_setup([model.predicted_tokens, model.entropy_reg, scores, avg_sample_loss, train_step],
                                   [input_seq_batch, seq_length_batch, image_feat_batch, compiler.loom_input_tensor,
                                    expr_validity_batch, answer_label_batch, gt_layout_batch])        # :168-173
        tokens, entropy_reg_val = sess.partial_run(h, (model.predicted_tokens, model.entropy_reg), feed_dict={
            input_seq_batch: z['b%d_input_seq' % i], seq_length_batch: z['b%d_seq_length' % i],
            image_feat_batch: _features(z['b%d_image_ids' % i], d), gt_layout_batch: z['b%d_gt_layout' % i]})
        assert np.array_equal(tokens, z['r%d_tokens' % i]) and np.array_equal(tokens, z['b%d_gt_layout' % i])
        expr_list, expr_validity_array = assembler.assemble(tokens)                                   # :182
        assert np.all(expr_validity_array)
        expr_feed = compiler.build_feed_dict(expr_list)
        expr_feed[expr_validity_batch] = expr_validity_array
        expr_feed[answer_label_batch] = z['b%d_labels' % i]
        scores_val, avg_sample_loss_val, _ = sess.partial_run(h, (scores, avg_sample_loss, train_step),
                                                              feed_dict=expr_feed)                    # :193-194
        es = float(np.abs(scores_val - z['r%d_scores' % i]).max())
        el = abs(float(avg_sample_loss_val) - float(z['r%d_avg_sample_loss' % i]))
        ee = abs(float(entropy_reg_val) - float(z['r%d_entropy_reg' % i]))
        if i == 0:
            assert es <= 1e-4 and el <= 1e-4 and ee <= 1e-4, (es, el, ee)
        worst = dict(scores=max(worst['scores'], es), loss=max(worst['loss'], el), entropy=max(worst['entropy'], ee))
    step = train_step._step
    assert type(step.trainer).__module__ == 'n2nmn_amd.train' and step.trainer.iteration == n_iter
    assert step.plan.objective == 0 and step.trainer.weight_decay == meta['weight_decay']
    print('train_clevr_gt_layout replay, %d iterations: worst |scores| %.2e, |avg_sample_loss| %.2e, |entropy_reg| %.2e'
          % (n_iter, worst['scores'], worst['loss'], worst['entropy']))
    assert worst['scores'] <= 5e-3 and worst['loss'] <= 2e-3 and worst['entropy'] <= 2e-3, worst
    wp = _probes(z, model.get_weights())
    lr = meta['hyper']['lr']
    bad = {k: v for k, v in wp.items() if v > 0.2 * n_iter * lr}
    print('weights after %d steps: worst probe diff %.2e (%s)' % (n_iter, max(wp.values()), max(wp, key=wp.get)))
    assert not bad, bad
    # the summary and the snapshot the script writes at its intervals (:205-223)
    summary = sess.run(log_step_trn, {loss_ph: avg_sample_loss_val})
    log_writer.add_summary(summary, n_iter)
    assert runtime_train.read_events(log_writer.path)[0][0] == n_iter
    snapshot_file = snapshot_saver.save(sess, str(tmp_path / 'tfmodel' / ('%08d' % n_iter)), write_meta_graph=False)
    from n2nmn_amd import tf_checkpoint
    back = tf_checkpoint.read_checkpoint(snapshot_file)
    now = model.get_weights()
    assert set(back) == set(now)
    for k, v in now.items():
        assert np.array_equal(back[k], v.cpu().numpy()), k
    tf.train.Saver().restore(sess, snapshot_file)               # (eval_clevr.py:90-91 on the file just written)


def _replay_policy_gradient(trace, tmp_path, scratch):
    """train_clevr_rl_gt_layout.py (scratch=False) or train_clevr_scratch.py (scratch=True: T_decoder 6, Adam at its
    default learning rate, no snapshot restored -- the run starts from the initializer, whose draws the recording
    replaced with seeded weights)"""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.nmn3_model import NMN3Model
    from n2nmn_amd.runtime import tf
    from n2nmn_amd import runtime, runtime_train, tf_checkpoint
    z, meta = _load(trace)
    d = TC.train_dims(TC.T_DECODER_SCRATCH if scratch else None)
    runtime._MODELS.clear()
    runtime_train._GLOBALS.clear()
    kw = dict(meta['model_kwargs'])
    assert kw.pop('assembler') == 'Assembler' and 'use_gt_layout' not in kw
    assert kw['T_decoder'] == d.T_decoder
    rl = meta['rl']
    # ---- the script's graph (train_clevr_rl_gt_layout.py; train_clevr_scratch.py is the same block 3 lines up) ----
    sess = tf.Session(config=tf.ConfigProto(gpu_options=tf.GPUOptions(allow_growth=True)))
    assembler = Assembler(list(synth.CLEVR_MODULE_NAMES))
    input_seq_batch = tf.placeholder(tf.int32, [None, None])                                          # :82-86
    seq_length_batch = tf.placeholder(tf.int32, [None])
    image_feat_batch = tf.placeholder(tf.float32, [None, d.H, d.W, d.D])
    expr_validity_batch = tf.placeholder(tf.bool, [None])
    answer_label_batch = tf.placeholder(tf.int32, [None])
    model = NMN3Model(image_feat_batch, input_seq_batch, seq_length_batch, assembler=assembler, **kw)  # :89-99
    compiler, scores, log_seq_prob = model.compiler, model.scores, model.log_seq_prob
    softmax_loss_per_sample = tf.nn.sparse_softmax_cross_entropy_with_logits(logits=scores, labels=answer_label_batch)
    final_loss_per_sample = tf.where(expr_validity_batch, softmax_loss_per_sample,
                                     tf.ones_like(softmax_loss_per_sample) * rl['invalid_expr_loss'])   # :112-114
    avg_sample_loss = tf.reduce_mean(final_loss_per_sample)                                           # :119
    baseline = tf.Variable(rl['invalid_expr_loss'], trainable=False, dtype=tf.float32)                # :120
    baseline_update_op = tf.assign_add(baseline, (1 - rl['baseline_decay']) * (avg_sample_loss - baseline))
    policy_gradient_loss = tf.reduce_mean(tf.stop_gradient(final_loss_per_sample - baseline) * log_seq_prob)
    total_training_loss = policy_gradient_loss + avg_sample_loss
    total_loss = tf.add_n([total_training_loss, rl['lambda_entropy'] * model.entropy_reg,
                           meta['weight_decay'] * model.l2_reg])                                      # :127-129
    if scratch:
        solver = tf.train.AdamOptimizer()                                                             # scratch :127
    else:
        solver = tf.train.AdamOptimizer(learning_rate=meta['hyper']['lr'])                            # :132
    gradients = solver.compute_gradients(total_loss)
    gradients = [(tf.clip_by_norm(g, meta['hyper']['max_grad_l2_norm']), v) for g, v in gradients]
    solver_op = solver.apply_gradients(gradients)
    with tf.control_dependencies([solver_op, baseline_update_op]):                                    # :144-145
        train_step = tf.constant(0)
    sess.run(tf.global_variables_initializer())                                                       # :165
    if scratch:
        model.load_weights(synth.make_weights(d, seed=3))       # (the recording started from these: run_train_script)
    else:
        # `snapshot_loader.restore(sess, pretrained_model)` (:168-169): a TensorFlow-format checkpoint of the weights
        # the recording's scratch tree held
        tf_checkpoint.write_checkpoint(str(tmp_path / '00050000'), synth.make_weights(d, seed=3))
        snapshot_loader = tf.train.Saver([v for v in tf.global_variables() if v != baseline])
        snapshot_loader.restore(sess, str(tmp_path / '00050000'))
    n_iter = len(meta['iterations'])
    same_tokens, compared = True, 0
    for i in range(n_iter):
        h = sess.partial_run_setup([model.predicted_tokens, model.entropy_reg, scores, avg_sample_loss, train_step],
                                   [input_seq_batch, seq_length_batch, image_feat_batch, compiler.loom_input_tensor,
                                    expr_validity_batch, answer_label_batch])                         # :178-183
        tokens, entropy_reg_val = sess.partial_run(h, (model.predicted_tokens, model.entropy_reg), feed_dict={
            input_seq_batch: z['b%d_input_seq' % i], seq_length_batch: z['b%d_seq_length' % i],
            image_feat_batch: _features(z['b%d_image_ids' % i], d)})
        assert tokens.shape[0] == d.T_decoder
        expr_list, expr_validity_array = assembler.assemble(tokens)
        assert np.all(expr_validity_array)                       # (the script asserts it too, :194)
        expr_feed = compiler.build_feed_dict(expr_list)
        expr_feed[expr_validity_batch] = expr_validity_array
        expr_feed[answer_label_batch] = z['b%d_labels' % i]
        scores_val, avg_sample_loss_val, _ = sess.partial_run(h, (scores, avg_sample_loss, train_step),
                                                              feed_dict=expr_feed)
        # the layouts are SAMPLED from the same uniforms (host generator, seed 0): equal to the recording's as long as
        # no draw falls within the fp32 / fp64 difference of a cumulative probability; values are compared while they are
        same_tokens = same_tokens and np.array_equal(tokens, z['r%d_tokens' % i])
        if i == 0:
            assert same_tokens, 'iteration 1 samples from identical weights and uniforms'
        if same_tokens:
            compared += 1
            es = float(np.abs(scores_val - z['r%d_scores' % i]).max())
            el = abs(float(avg_sample_loss_val) - float(z['r%d_avg_sample_loss' % i]))
            assert es <= (1e-4 if i == 0 else 5e-3) and el <= (1e-4 if i == 0 else 2e-3), (i, es, el)
    step = train_step._step
    assert step.plan.objective == 1 and type(step.trainer).__module__ == 'n2nmn_amd.train'
    assert step.trainer.hyper['lr'] == meta['hyper']['lr'] and step.trainer.weight_decay == meta['weight_decay']
    for k in rl:
        assert abs(step.trainer.rl[k] - rl[k]) < 1e-6, k
    print('%s replay: %d of %d iterations sampled the recorded layouts and were compared' % (trace, compared, n_iter))
    assert compared >= 3
    b = float(sess.run(baseline))
    assert 0.5 < b < 4.0 and abs(b - step.trainer.get_baseline()) < 1e-7


def test_replay_of_train_clevr_rl_gt_layout_on_the_hip_trainer(tmp_path):
    _replay_policy_gradient('train_driver_trace_rl.npz', tmp_path, scratch=False)


def test_replay_of_train_clevr_scratch_on_the_hip_trainer(tmp_path):
    """tests/golden/train_driver_trace_scratch.npz: exp_clevr/train_clevr_scratch.py, UNMODIFIED, on the CPU box"""
    _replay_policy_gradient('train_driver_trace_scratch.npz', tmp_path, scratch=True)
