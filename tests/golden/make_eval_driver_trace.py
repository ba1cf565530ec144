#!/usr/bin/env python3
"""Recorded run of the reference's exp_clevr/eval_clevr.py against the drop-in (missing #4).

The reference checkout exists only in the build container and the GPU only on the gpurun box, so the
reference's file cannot drive the HIP engine directly.  What CAN travel is what the file DID: this script
executes it, unmodified, in the scratch tree of tests/eval_driver_common.py (the drop-in answers its
imports, the CPU oracle computes behind the drop-in's Python face) and records

  * the keyword arguments it constructed NMN3Model with and the placeholders it made,
  * every `sess.partial_run(handle, fetch, feed_dict)` it issued -- fetch, feeds (arrays; image features
    by question index; the packed program of `compiler.build_feed_dict(expr_list)` as its node list) --
    and the value it got back (oracle, fp64),
  * the answers it wrote to its eval_outputs file.

tests/test_gpu_eval_driver_trace.py replays exactly these calls on the same drop-in objects over the HIP
engine and compares.  tests/test_reference_driver_source.py re-records on every CPU run and checks the
committed file is what the reference's script still does.

    python tests/golden/make_eval_driver_trace.py [--check]
"""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
OUT = os.path.join(HERE, 'eval_driver_trace.npz')


class _Patch:
    """the three monkeypatch calls the helpers use, outside pytest"""

    def __init__(self):
        self._undo = []

    def setattr(self, obj, name, value):
        old = getattr(obj, name)
        self._undo.append(lambda: setattr(obj, name, old))
        setattr(obj, name, value)

    def setitem(self, d, k, v):
        had, old = k in d, d.get(k)
        self._undo.append((lambda: d.__setitem__(k, old)) if had else (lambda: d.pop(k, None)))
        d[k] = v

    def chdir(self, p):
        old = os.getcwd()
        self._undo.append(lambda: os.chdir(old))
        os.chdir(str(p))

    def undo(self):
        for f in reversed(self._undo):
            f()


def pack_trace(rec, answers_written, answers, result_dtype=None):
    """recorder -> dict of arrays for np.savez (answers_written: words of `answers`, or indices already;
    result_dtype: store the returned values in this type -- float32 keeps the 3001-answer VQA logits small)"""
    out, calls, handles, setups = {}, [], {}, []
    for k, c in enumerate(rec.calls):
        feeds = {}
        for name, (kind, v) in c['feeds'].items():
            key = 'c%d_feed%d' % (k, len(feeds))
            if kind == 'packed':
                out[key] = np.asarray([list(r) for r in v.nodes().tolist()], np.int32).reshape(-1, 8)
                feeds[name] = dict(kind='packed', key=key, num_rows=int(v.num_rows))
            elif kind == 'image_ids':
                out[key] = np.asarray(v, np.int32)
                feeds[name] = dict(kind='image_ids', key=key)
            else:
                out[key] = v
                feeds[name] = dict(kind='array', key=key)
        out['c%d_result' % k] = c['result'] if (result_dtype is None or c['result'].dtype.kind != 'f') \
            else c['result'].astype(result_dtype)
        if c['handle'] not in handles:
            handles[c['handle']] = len(handles)
            setups.append({k: v for k, v in rec.setups[c['handle']].items() if k != 'keep'})
        calls.append(dict(fetch=c['fetch'], handle=handles[c['handle']], feeds=feeds))
    meta = dict(model_kwargs=rec.model_kwargs,
                placeholders={k: [str(v[1]), list(v[2])] for k, v in rec.placeholders.items()},
                setups=setups, calls=calls)
    out['meta'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    out['answers_written'] = np.asarray([a if answers is None else answers.index(a) for a in answers_written], np.int32)
    return out


def record():
    import eval_driver_common as EC
    from oracle_engine import OracleEngine
    from n2nmn_amd.spec import Dims
    mp = _Patch()
    rec = EC.SessionRecorder(Dims())
    with tempfile.TemporaryDirectory() as tmp:
        try:
            g, d, data, words, answers, w = EC.run_reference_script(Path(tmp), mp, OracleEngine, rec)
            written = [l.strip() for l in open(Path(tmp) / 'exp_clevr' / 'eval_outputs' / 'exp0' / '00050000.syn.txt')]
        finally:
            mp.undo()
    return pack_trace(rec, written, answers)


OUT_VQA = os.path.join(HERE, 'eval_driver_trace_vqa2.npz')
OUT_SHAPES = os.path.join(HERE, 'eval_driver_trace_shapes.npz')


def record_vqa():
    """exp_vqa/eval_vqa2.py (tests/eval_driver_more.py): the answers it wrote come from its results json"""
    import json as _json
    import eval_driver_common as EC
    import eval_driver_more as EM
    from oracle_engine import OracleVQAEngine
    from n2nmn_amd import models_vqa
    mp = _Patch()
    rec = EC.SessionRecorder(None, model_cls=models_vqa.NMN3Model, feature_fn=EM.vqa_feature_of,
                             n_questions=EM.VQA_N)
    with tempfile.TemporaryDirectory() as tmp:
        try:
            g, data, words, answers, w = EM.run_vqa_script(Path(tmp), mp, OracleVQAEngine, rec)
            res = _json.load(open(Path(tmp) / 'exp_vqa' / 'eval_outputs' / 'exp0' /
                                  'vqa_OpenEnded_mscoco_syn_exp0_00040000_results.json'))
        finally:
            mp.undo()
    assert [r['question_id'] for r in res] == [1000 + i for i in range(EM.VQA_N)]
    return pack_trace(rec, [r['answer'] for r in res], answers, result_dtype=np.float32)


def record_shapes():
    """exp_shapes/eval_shapes.py: it writes accuracies, not answers -- its per-question predictions are
    the arg max of the scores it fetched (`predictions`, :166)"""
    import eval_driver_common as EC
    import eval_driver_more as EM
    from oracle_engine import OracleShapesEngine
    from n2nmn_amd import models_shapes
    mp = _Patch()
    rec = EC.SessionRecorder(None, model_cls=models_shapes.NMN3ModelAtt, feature_fn=None)
    with tempfile.TemporaryDirectory() as tmp:
        try:
            g, w = EM.run_shapes_script(Path(tmp), mp, OracleShapesEngine, rec)
            summary = open(Path(tmp) / 'exp_shapes' / 'results' / 'exp0' / '00040000.train.tiny.txt').read()
        finally:
            mp.undo()
    preds = np.concatenate([np.argmax(c['result'], axis=1) for c in rec.calls if c['fetch'] == 'scores'])
    t = pack_trace(rec, list(preds), None)
    t['summary'] = np.frombuffer(summary.encode(), np.uint8)
    return t


def same(a, b):
    if set(a) != set(b):
        return 'keys differ: %s' % sorted(set(a) ^ set(b))
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.shape != y.shape:
            return '%s: shape %s vs %s' % (k, x.shape, y.shape)
        if x.dtype.kind == 'f':
            if np.abs(x - y).max() > 1e-12:
                return '%s: differs by %g' % (k, np.abs(x - y).max())
        elif not np.array_equal(x, y):
            return '%s: differs' % k
    return None


if __name__ == '__main__':
    bad = 0
    for out, rec_fn in ((OUT, record), (OUT_VQA, record_vqa), (OUT_SHAPES, record_shapes)):
        t = rec_fn()
        if '--check' in sys.argv:
            z = np.load(out)
            err = same(t, {k: z[k] for k in z.files})
            print(os.path.basename(out) + ':', err or 'reproduced')
            bad += bool(err)
            continue
        np.savez_compressed(out, **t)
        print('wrote', out, '%d calls, %.0f KB' % (len(json.loads(bytes(t['meta']))['calls']), os.path.getsize(out) / 1e3))
    sys.exit(1 if bad else 0)
