#!/usr/bin/env python3
"""Recorded runs of the reference's TRAINING drivers against the drop-in:
exp_clevr/train_clevr_gt_layout.py, exp_clevr/train_clevr_rl_gt_layout.py and exp_clevr/train_clevr_scratch.py
(policy gradient from random weights: no ground-truth layouts are even loaded), executed unmodified in the
scratch tree of tests/train_driver_common.py over the CPU oracle doubles (OracleEngine + OracleTrainer, fp64).

Recorded per script: the batches its reader delivered (text, lengths, labels, layouts, image features by question
index), what the model constructor and the matched loss graph asked of the Trainer (objective, weight decay,
Adam hyper-parameters, clip norm, REINFORCE constants), and per iteration what its two partial_run calls
returned (tokens, entropy_reg; scores, avg_sample_loss), plus probes of every variable after the last step.
tests/test_gpu_train_driver_trace.py rebuilds the same graph through n2nmn_amd.runtime.tf on the GPU box and
replays the batches over the HIP engine and the HIP Trainer.

The models_vqa training drivers exp_vqa/train_vqa2_gt_layout.py and train_vqa2_rl_gt_layout.py (lstm_dim 1000, dropout
on both LSTM stacks and the question prior net, no clipping / clip 10, GloVe rows assigned into embedding_mat) are
recorded the same way (pack_vqa: 48 answer columns of `scores` per question, arg-max, max; digests of every
iteration's dropout masks, which the replay regenerates from the face's host generator).

    python tests/golden/make_train_driver_trace.py [--check]
"""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), HERE]
OUT_GT = os.path.join(HERE, 'train_driver_trace_gt.npz')
OUT_RL = os.path.join(HERE, 'train_driver_trace_rl.npz')
OUT_SCRATCH = os.path.join(HERE, 'train_driver_trace_scratch.npz')
OUT_VQA2_GT = os.path.join(HERE, 'train_driver_trace_vqa2_gt.npz')
OUT_VQA2_RL = os.path.join(HERE, 'train_driver_trace_vqa2_rl.npz')
VQA_SCORE_PROBES = 48          # answer columns of `scores` kept per question (of 3001)
PROBES = 8                     # elements per variable compared after the last step


def probe_indices(name, size):
    rng = np.random.default_rng(abs(hash_name(name)) % (2 ** 32))
    return np.sort(rng.choice(size, size=min(PROBES, size), replace=False))


def hash_name(name):
    h = 0
    for ch in name.encode():
        h = (h * 131 + ch) % (2 ** 31 - 1)
    return h


def pack(rec, batches, trainer):
    out = {}
    iters = []
    p1 = [c for c in rec.calls if c['fetch'].startswith('(predicted_tokens')]
    p2 = [c for c in rec.calls if c['fetch'] == '(scores, avg_sample_loss, train_step)']
    assert len(p1) == len(p2) == len(batches)
    for i, (b, c1, c2) in enumerate(zip(batches, p1, p2)):
        out['b%d_input_seq' % i] = np.asarray(b['input_seq_batch'], np.int32)
        out['b%d_seq_length' % i] = np.asarray(b['seq_length_batch'], np.int32)
        out['b%d_labels' % i] = np.asarray(b['answer_label_batch'], np.int32)
        if 'gt_layout_batch' in b:
            out['b%d_gt_layout' % i] = np.asarray(b['gt_layout_batch'], np.int32)
        kind, ids = c1['feeds']['image_feat_grid']
        assert kind == 'image_ids' and min(ids) >= 0
        out['b%d_image_ids' % i] = np.asarray(ids, np.int32)
        out['r%d_tokens' % i] = np.asarray(c1['result_list'][0], np.int32)
        out['r%d_entropy_reg' % i] = np.asarray(c1['result_list'][1], np.float64)
        out['r%d_scores' % i] = np.asarray(c2['result_list'][0], np.float64)
        out['r%d_avg_sample_loss' % i] = np.asarray(c2['result_list'][1], np.float64)
        iters.append(dict(feeds1=sorted(c1['feeds']), feeds2=sorted(c2['feeds'])))
    for name, v in trainer.engine.weights.items():
        flat = np.asarray(v, np.float64).reshape(-1)
        out['w_' + name] = flat[probe_indices(name, flat.size)]
    meta = dict(model_kwargs=rec.model_kwargs, iterations=iters, weight_decay=trainer.weight_decay, hyper=trainer.hyper,
                rl=trainer.rl, objectives=[o for o, _ in trainer.history],
                setups=[{k: v for k, v in s.items() if k != 'keep'} for s in list(rec.setups.values())[:1]])
    out['meta'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    return out


def mask_digest(masks):
    """one number per mask of a handle: sum of (index + 1) over the kept elements, mod 2^31 - 1 (the replay regenerates
    the masks from the face's host generator and must arrive at the same ones)"""
    out = []
    for k in ('enc0', 'dec0', 'qpn_h', 'qpn_fc1'):
        if k in masks:
            m = np.asarray(masks[k]).reshape(-1) > 0
            out.append(int((np.flatnonzero(m).astype(np.int64) + 1).sum() % (2 ** 31 - 1)))
        else:
            out.append(-1)
    return np.asarray(out, np.int64)


def vqa_probe_columns(num_choices):
    rng = np.random.default_rng(77)
    return np.sort(rng.choice(num_choices, size=VQA_SCORE_PROBES, replace=False))


def pack_vqa(rec, batches, trainer, glove):
    """like pack(), for the models_vqa training drivers: `scores` is [N, 3001] -- kept are VQA_SCORE_PROBES columns,
    the arg-max and the max per question; the dropout masks of every iteration as digests"""
    out = {}
    iters = []
    p1 = [c for c in rec.calls if c['fetch'].startswith('(predicted_tokens')]
    p2 = [c for c in rec.calls if c['fetch'] == '(scores, avg_sample_loss, train_step)']
    assert len(p1) == len(p2) == len(batches) == len(trainer.mask_history)
    cols = None
    for i, (b, c1, c2) in enumerate(zip(batches, p1, p2)):
        out['b%d_input_seq' % i] = np.asarray(b['input_seq_batch'], np.int32)
        out['b%d_seq_length' % i] = np.asarray(b['seq_length_batch'], np.int32)
        out['b%d_labels' % i] = np.asarray(b['answer_label_batch'], np.int32)
        if 'gt_layout_batch' in b:
            out['b%d_gt_layout' % i] = np.asarray(b['gt_layout_batch'], np.int32)
        kind, ids = c1['feeds']['image_feat_grid']
        assert kind == 'image_ids' and min(ids) >= 0
        out['b%d_image_ids' % i] = np.asarray(ids, np.int32)
        out['r%d_tokens' % i] = np.asarray(c1['result_list'][0], np.int32)
        out['r%d_entropy_reg' % i] = np.asarray(c1['result_list'][1], np.float64)
        sc = np.asarray(c2['result_list'][0], np.float64)
        if cols is None:
            cols = vqa_probe_columns(sc.shape[1])
        out['r%d_scores_probe' % i] = sc[:, cols]
        out['r%d_scores_argmax' % i] = np.argmax(sc, axis=1).astype(np.int32)
        out['r%d_scores_max' % i] = sc.max(axis=1)
        out['r%d_avg_sample_loss' % i] = np.asarray(c2['result_list'][1], np.float64)
        out['r%d_mask_digest' % i] = mask_digest(trainer.mask_history[i])
        iters.append(dict(feeds1=sorted(c1['feeds']), feeds2=sorted(c2['feeds'])))
    for name, v in trainer.engine.weights.items():
        flat = np.asarray(v, np.float64).reshape(-1)
        out['w_' + name] = flat[probe_indices(name, flat.size)]
    out['glove_probe'] = np.asarray(glove, np.float64).reshape(-1)[probe_indices('glove', glove.size)]
    meta = dict(model_kwargs=rec.model_kwargs, iterations=iters, weight_decay=trainer.weight_decay, hyper=trainer.hyper,
                rl=trainer.rl, objectives=[o for o, _ in trainer.history], dropout=trainer.dropout)
    out['meta'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    return out


def record_vqa(which):
    """which: 'vqa2_gt' -> exp_vqa/train_vqa2_gt_layout.py, 'vqa2_rl' -> exp_vqa/train_vqa2_rl_gt_layout.py"""
    import eval_driver_common as EC
    import eval_driver_more as EM
    from make_eval_driver_trace import _Patch
    from n2nmn_amd import models_vqa
    from oracle_engine import OracleTrainer, OracleVQAEngine, OracleVQATrainer
    mp = _Patch()
    OracleTrainer.made.clear()
    rec = EC.SessionRecorder(None, model_cls=models_vqa.NMN3Model, feature_fn=EM.vqa_feature_of,
                             n_questions=EM.VQA_TRAIN_N)
    with tempfile.TemporaryDirectory() as tmp:
        try:
            if which == 'vqa2_gt':
                g, batches, w, glove, answers = EM.run_vqa_train_script(
                    'train_vqa2_gt_layout.py', Path(tmp), mp, OracleVQAEngine, OracleVQATrainer, rec)
            else:
                g, batches, w, glove, answers = EM.run_vqa_train_script(
                    'train_vqa2_rl_gt_layout.py', Path(tmp), mp, OracleVQAEngine, OracleVQATrainer, rec,
                    snapshot='vqa2_gt_layout/00080000', seed_weights=False)
        finally:
            mp.undo()
    return pack_vqa(rec, batches, OracleTrainer.made[0], glove)


def same(a, b):
    """None if the two packed traces agree (floats to 1e-9: fp64 oracle, thread-count dependent sums), else why"""
    if set(a) != set(b):
        return 'keys differ: %s' % sorted(set(a) ^ set(b))
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.shape != y.shape:
            return '%s: shape %s vs %s' % (k, x.shape, y.shape)
        if x.dtype.kind == 'f':
            if not np.allclose(x, y, rtol=0, atol=1e-9):
                return '%s: max |diff| %.3e' % (k, float(np.abs(x - y).max()))
        elif not np.array_equal(x, y):
            return '%s differs' % k
    return None


def record(which):
    import eval_driver_common as EC
    import train_driver_common as TC
    from make_eval_driver_trace import _Patch
    from oracle_engine import OracleEngine, OracleTrainer
    mp = _Patch()
    OracleTrainer.made.clear()
    rec = EC.SessionRecorder(TC.train_dims(TC.T_DECODER_SCRATCH if which == 'scratch' else None),
                             n_questions=TC.N_QUESTIONS)
    with tempfile.TemporaryDirectory() as tmp:
        try:
            if which == 'gt':
                g, d, batches, w = TC.run_train_script(TC.SCRIPT_GT, Path(tmp), mp, OracleEngine, OracleTrainer, rec)
            elif which == 'scratch':
                g, d, batches, w = TC.run_train_script(TC.SCRIPT_SCRATCH, Path(tmp), mp, OracleEngine, OracleTrainer,
                                                       rec, t_decoder=TC.T_DECODER_SCRATCH)
            else:
                g, d, batches, w = TC.run_train_script(TC.SCRIPT_RL, Path(tmp), mp, OracleEngine, OracleTrainer, rec,
                                                       with_snapshot=True, seed_weights=False)
        finally:
            mp.undo()
    return pack(rec, batches, OracleTrainer.made[0])


if __name__ == '__main__':
    for which, path in (('gt', OUT_GT), ('rl', OUT_RL), ('scratch', OUT_SCRATCH), ('vqa2_gt', OUT_VQA2_GT),
                        ('vqa2_rl', OUT_VQA2_RL)):
        fresh = record_vqa(which) if which.startswith('vqa') else record(which)
        if '--check' in sys.argv:
            z = np.load(path)
            why = same(fresh, {k: z[k] for k in z.files})
            print(path, 'matches the reference script run' if why is None else 'DIFFERS: ' + why)
            if why is not None:
                sys.exit(1)
        else:
            np.savez_compressed(path, **fresh)
            print('wrote', path, os.path.getsize(path), 'bytes')
