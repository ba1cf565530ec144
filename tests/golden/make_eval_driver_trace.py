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


def pack_trace(rec, answers_written, answers):
    """recorder -> dict of arrays for np.savez"""
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
        out['c%d_result' % k] = c['result']
        if c['handle'] not in handles:
            handles[c['handle']] = len(handles)
            setups.append({k: v for k, v in rec.setups[c['handle']].items() if k != 'keep'})
        calls.append(dict(fetch=c['fetch'], handle=handles[c['handle']], feeds=feeds))
    meta = dict(model_kwargs=rec.model_kwargs,
                placeholders={k: [str(v[1]), list(v[2])] for k, v in rec.placeholders.items()},
                setups=setups, calls=calls)
    out['meta'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    out['answers_written'] = np.asarray([answers.index(a) for a in answers_written], np.int32)
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
    t = record()
    if '--check' in sys.argv:
        z = np.load(OUT)
        err = same(t, {k: z[k] for k in z.files})
        print('eval_driver_trace.npz:', err or 'reproduced')
        sys.exit(1 if err else 0)
    np.savez_compressed(OUT, **t)
    print('wrote', OUT, '%d calls, %.0f KB' % (len(json.loads(bytes(t['meta']))['calls']), os.path.getsize(OUT) / 1e3))
