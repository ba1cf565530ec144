"""The reference's own TRAINING drivers, UNMODIFIED, against the drop-in (VERDICT r5 "missing" #1; SURVEY 8(b):
"what calls it: ... train scripts train_clevr_gt_layout.py:60-223").

`/root/reference/exp_clevr/train_clevr_gt_layout.py`, `.../train_clevr_rl_gt_layout.py` and
`.../train_clevr_scratch.py` are executed as scripts (runpy, `__main__`); their imports are answered by the drop-in --

    import tensorflow as tf  -> n2nmn_amd.runtime.tf: placeholder, constant, nn.sparse_softmax_cross_entropy_with_logits,
                                reduce_mean, where, ones_like, stop_gradient, add_n, Variable, assign_add,
                                train.AdamOptimizer().compute_gradients / apply_gradients, clip_by_norm,
                                control_dependencies, summary.{FileWriter, scalar, merge}, get_default_graph,
                                global_variables_initializer, global_variables, train.Saver (n2nmn_amd.runtime_train)
    models_clevr.* / util.clevr_train.data_reader -> as for eval_clevr.py (tests/eval_driver_common.py)

Every line of the scripts runs: graph construction, the loss block (:104-113 / rl :107-129) MATCHED onto the
Trainer's objective, 21 iterations of partial_run pairs, the TensorBoard summary of iteration 20.  The compute
behind the Python face is the CPU oracle here (tests/oracle_engine.py: OracleEngine + OracleTrainer, fp64
autograd); what the scripts asked of the face is recorded (tests/golden/train_driver_trace*.npz) and replayed
over the HIP engine + HIP Trainer on the GPU box (tests/test_gpu_train_driver_trace.py)."""
import glob
import os
import sys

import numpy as np
import pytest

import eval_driver_common as EC
import train_driver_common as TC

pytestmark = pytest.mark.skipif(not os.path.exists(TC.SCRIPT_GT), reason='reference checkout not present')


def _recorder(d):
    return EC.SessionRecorder(d, n_questions=TC.N_QUESTIONS)


def test_train_clevr_gt_layout_script_runs_unmodified_against_the_drop_in(tmp_path, monkeypatch, capsys):
    from oracle import n2nmn_oracle_grad as G
    from oracle_engine import OracleEngine, OracleTrainer
    from n2nmn_amd import runtime_train
    OracleTrainer.made.clear()
    rec = _recorder(TC.train_dims())
    g, d, batches, w = TC.run_train_script(TC.SCRIPT_GT, tmp_path, monkeypatch, OracleEngine, OracleTrainer, rec)
    assert len(batches) == TC.N_ITERS
    # ---- the graph the script built was matched onto the behavioural-cloning step with ITS constants ----
    assert len(OracleTrainer.made) == 1
    tr = OracleTrainer.made[0]
    assert tr.weight_decay == g['weight_decay'] == 5e-6
    assert tr.hyper == dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, max_grad_l2_norm=float(g['max_grad_l2_norm']))
    assert [o for o, _ in tr.history] == [0] * TC.N_ITERS and tr.iteration == TC.N_ITERS
    plan = g['train_step']._step.plan
    assert plan.objective == 0 and plan.labels_ph is g['answer_label_batch'] and plan.validity_ph is None
    # ---- what the script fetched: an independent run of the oracle over the same batches, from the same
    # initial weights (loss_and_grads + clip + Adam; the script's own numpy bookkeeping for the accuracy) -----
    names = list(g['assembler'].module_names)
    w64 = {k: np.asarray(v, np.float64) for k, v in w.items()}
    m = {k: np.zeros_like(v) for k, v in w64.items()}
    v = {k: np.zeros_like(x) for k, x in w64.items()}
    calls = [c for c in rec.calls if c['fetch'] == '(scores, avg_sample_loss, train_step)']
    assert len(calls) == TC.N_ITERS
    for it in range(3):
        b = batches[it]
        L, grads, ex = G.loss_and_grads(w64, names, b, TC.T_DECODER, d.num_choices, b['gt_layout_batch'], 5e-6)
        scores_val, loss_val, _ = calls[it]['result_list']
        assert np.abs(scores_val - ex['scores']).max() < 1e-9, it
        assert abs(float(loss_val) - L['avg_sample_loss']) < 1e-6, it
        w64, m, v = G.adam_step(w64, grads, m, v, it + 1)
    # the weights moved, and by what Adam moves them (first step: lr per element whose gradient is not tiny)
    moved = np.abs(g['nmn3_model_trn'].engine.weights[EC_name('FindModule/conv_image/weights')] -
                   np.asarray(w[EC_name('FindModule/conv_image/weights')], np.float64)).max()
    assert 1e-3 < moved < 0.1
    # ---- the training log and the TensorBoard summary of iteration 20 ------------------------------------
    out = capsys.readouterr().out
    assert 'iter = 20\n\tloss = ' in out and 'validity = 1.000000' in out
    ev = glob.glob(str(tmp_path / 'exp_clevr' / 'tb' / 'clevr_gt_layout' / 'events.out.tfevents.*'))
    assert len(ev) == 1
    events = runtime_train.read_events(ev[0])
    assert [s for s, _ in events] == [20]
    assert set(events[0][1]) == {'avg_sample_loss', 'entropy', 'avg_accuracy', 'validity'}
    assert abs(events[0][1]['avg_sample_loss'] - float(calls[19]['result_list'][1])) < 1e-6
    assert events[0][1]['validity'] == 1.0
    # ---- the committed recording (replayed over the HIP engine: tests/test_gpu_train_driver_trace.py) ----
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_train_driver_trace as MT
    z = np.load(MT.OUT_GT)
    assert MT.same(MT.pack(rec, batches, tr), {k: z[k] for k in z.files}) is None


def EC_name(tail):
    from n2nmn_amd.spec import _MOD
    return _MOD + tail


def test_train_clevr_rl_gt_layout_script_runs_unmodified_against_the_drop_in(tmp_path, monkeypatch, capsys):
    from oracle_engine import OracleEngine, OracleTrainer
    OracleTrainer.made.clear()
    rec = _recorder(TC.train_dims())
    g, d, batches, w = TC.run_train_script(TC.SCRIPT_RL, tmp_path, monkeypatch, OracleEngine, OracleTrainer, rec,
                                           with_snapshot=True, seed_weights=False)
    assert len(OracleTrainer.made) == 1
    tr = OracleTrainer.made[0]
    # fine-tuning learning rate and the REINFORCE constants come from the script's graph
    assert tr.hyper['lr'] == g['finetune_lr'] == 1e-4 and tr.hyper['max_grad_l2_norm'] == 10.0
    assert tr.rl == dict(invalid_expr_loss=0.5, lambda_entropy=0.005, baseline_decay=0.99)
    assert tr.weight_decay == 5e-6
    assert [o for o, _ in tr.history] == [1] * TC.N_ITERS
    plan = g['train_step']._step.plan
    assert plan.objective == 1 and plan.validity_ph is g['expr_validity_batch'] and plan.baseline is g['baseline']
    # `snapshot_loader.restore(sess, pretrained_model)` (rl :168-169): the run started from the checkpoint, not
    # from the initializer's draws -- the first scores are the snapshot weights' scores
    from oracle import n2nmn_oracle as O
    calls = [c for c in rec.calls if c['fetch'] == '(scores, avg_sample_loss, train_step)']
    first_tokens = [c for c in rec.calls if c['fetch'].startswith('(predicted_tokens')][0]['result_list'][0]
    ref = O.forward({k: np.asarray(x, np.float64) for k, x in w.items()}, list(g['assembler'].module_names),
                    batches[0], TC.T_DECODER, d.num_choices, np.float64, use_gt_layout=True, gt_layout=first_tokens)
    assert np.abs(calls[0]['result_list'][0] - ref['scores']).max() < 1e-9
    # sampled layouts differ between questions and are valid (the script asserts it too)
    assert len({tuple(c) for c in first_tokens.T}) > 1
    # the baseline is the EMA the script's assign_add describes
    b = 0.5
    for _, L in tr.history:
        b = b + (1 - 0.99) * (L['avg_sample_loss'] - b)
    assert abs(tr.get_baseline() - b) < 1e-12
    assert abs(float(g['sess'].run(g['baseline'])) - b) < 1e-6
    out = capsys.readouterr().out
    assert 'iter = 20\n\tloss = ' in out
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_train_driver_trace as MT
    z = np.load(MT.OUT_RL)
    assert MT.same(MT.pack(rec, batches, tr), {k: z[k] for k in z.files}) is None


def test_train_clevr_scratch_script_runs_unmodified_against_the_drop_in(tmp_path, monkeypatch, capsys):
    """/root/reference/exp_clevr/train_clevr_scratch.py, every line of it: policy gradient from freshly initialised
    weights, T_decoder = 6, no ground-truth layouts loaded (load_gt_layout=False, :71-72), invalid_expr_loss = ln 28,
    lambda_entropy = 0.01, weight_decay = 0, Adam at its default learning rate, no snapshot restored."""
    from oracle_engine import OracleEngine, OracleTrainer
    OracleTrainer.made.clear()
    d = TC.train_dims(TC.T_DECODER_SCRATCH)
    rec = _recorder(d)
    g, d, batches, w = TC.run_train_script(TC.SCRIPT_SCRATCH, tmp_path, monkeypatch, OracleEngine, OracleTrainer, rec,
                                           t_decoder=TC.T_DECODER_SCRATCH)
    assert g['T_decoder'] == 6 and all('gt_layout_batch' not in b for b in batches)
    assert len(OracleTrainer.made) == 1
    tr = OracleTrainer.made[0]
    assert tr.hyper == dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, max_grad_l2_norm=10.0)
    assert tr.weight_decay == 0.0 and tr.rl['lambda_entropy'] == 0.01 and tr.rl['baseline_decay'] == 0.99
    assert abs(tr.rl['invalid_expr_loss'] - np.log(28)) < 1e-6          # (the graph holds it as a float32 constant)
    assert [o for o, _ in tr.history] == [1] * TC.N_ITERS
    plan = g['train_step']._step.plan
    assert plan.objective == 1 and plan.baseline is g['baseline'] and plan.weight_decay == 0.0
    # sampled layouts: at most 6 tokens, valid, not all alike
    first_tokens = [c for c in rec.calls if c['fetch'].startswith('(predicted_tokens')][0]['result_list'][0]
    assert first_tokens.shape[0] == 6 and len({tuple(c) for c in first_tokens.T}) > 1
    b = float(np.log(28))
    for _, L in tr.history:
        b = b + (1 - 0.99) * (L['avg_sample_loss'] - b)
    assert abs(tr.get_baseline() - b) < 1e-6
    assert 'iter = 20\n\tloss = ' in capsys.readouterr().out
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_train_driver_trace as MT
    z = np.load(MT.OUT_SCRATCH)
    assert MT.same(MT.pack(rec, batches, tr), {k: z[k] for k in z.files}) is None


def test_a_loss_graph_the_step_does_not_implement_is_refused():
    """'refuse loudly any loss graph that is not the one :104-124 builds'"""
    from n2nmn_amd import runtime_train as R
    from n2nmn_amd.runtime import Fetch, tf

    class M:            # a model face: only the fetch names matter to the matcher
        pass
    m = M()
    scores, lsp, l2 = Fetch(m, 'scores', 2), Fetch(m, 'log_seq_prob', 1), Fetch(m, 'l2_reg', 2)
    labels = tf.placeholder(tf.int32, [None])
    ce = tf.nn.sparse_softmax_cross_entropy_with_logits(logits=scores, labels=labels)
    ok = tf.reduce_mean(-lsp) + tf.reduce_mean(ce) + 5e-6 * l2
    plan = R.match_loss(ok)
    assert plan.objective == 0 and plan.weight_decay == 5e-6 and plan.labels_ph is labels
    for bad in (tf.reduce_mean(ce),                                          # no sequence likelihood term
                2.0 * tf.reduce_mean(-lsp) + tf.reduce_mean(ce),             # another weighting
                tf.reduce_mean(-lsp) + tf.reduce_mean(ce * ce),              # another per-sample loss
                tf.reduce_mean(-lsp) + tf.reduce_mean(ce) + tf.reduce_mean(scores)):
        with pytest.raises(NotImplementedError, match='not one the training step implements'):
            R.match_loss(bad)
