"""Minimal stand-ins for the three TensorFlow objects the reference's driver loop touches
(exp_clevr/eval_clevr.py:72-75,103-132): placeholders, fetchable model attributes and a session with
`partial_run_setup` / `partial_run`.  They carry no graph: a `partial_run` that fetches a phase-1
attribute launches phase 1 on the GPU, one that fetches `scores` launches phase 2.
"""
from __future__ import annotations

import numpy as np


class Placeholder:
    """tf.placeholder(dtype, shape): a named slot that `feed_dict` binds to an array/tensor."""
    _count = 0

    def __init__(self, dtype=None, shape=None, name=None):
        Placeholder._count += 1
        self.dtype, self.shape = dtype, shape
        self.name = name or 'Placeholder_%d' % Placeholder._count

    def __repr__(self):
        return '<Placeholder %s %s>' % (self.name, self.shape)


def placeholder(dtype=None, shape=None, name=None):
    return Placeholder(dtype, shape, name)


class Fetch:
    """A fetchable attribute of a model (`model.predicted_tokens`, `model.scores`, ...)."""

    def __init__(self, owner, name, phase):
        self.owner, self.name, self.phase = owner, name, phase

    def __repr__(self):
        return '<Fetch %s (phase %d)>' % (self.name, self.phase)


def resolve(x, feeds):
    """placeholder -> fed value; anything else is already a value."""
    if isinstance(x, Placeholder):
        if x not in feeds:
            raise ValueError('placeholder %s was not fed' % x.name)
        return feeds[x]
    return x


class _Handle:
    def __init__(self, fetches, feeds):
        self.allowed_fetches = list(fetches)
        self.allowed_feeds = list(feeds)
        self.feeds = {}
        self.phase1 = None      # results of phase 1 (device tensors)
        self.results = {}


class Session:
    """`sess.partial_run_setup(fetches, feeds)` then `sess.partial_run(h, fetch, feed_dict)`, as
    in exp_clevr/eval_clevr.py:105-132.  Returned values are numpy arrays, like TF's."""

    def __init__(self):
        self.last = {}           # name of the last fetch -> value returned (debugging aid)

    def partial_run_setup(self, fetches, feeds=None):
        return _Handle(fetches if isinstance(fetches, (list, tuple)) else [fetches], feeds or [])

    def partial_run(self, handle, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        for k in (feed_dict or {}):
            if handle.allowed_feeds and k not in handle.allowed_feeds:
                raise ValueError('feed %r was not declared in partial_run_setup' % (k,))
        handle.feeds.update(feed_dict or {})
        out = []
        for f in flist:
            if not isinstance(f, Fetch):
                raise TypeError('cannot fetch %r' % (f,))
            out.append(f.owner._fetch(f, handle))
            self.last[f.name] = out[-1]
        return out[0] if single else out

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, _NoOp):        # sess.run(tf.global_variables_initializer()): nothing to do,
            return None                       # variables live in the engine and are set by Saver.restore
        h = self.partial_run_setup(fetches, list((feed_dict or {}).keys()))
        return self.partial_run(h, fetches, feed_dict)


# ---- the handful of `tf.*` names the reference's driver scripts use (exp_clevr/eval_clevr.py:15-19,
# 72-75,90-91), so that a driver runs with `from n2nmn_amd.runtime import tf` in place of
# `import tensorflow as tf` and nothing else changed -----------------------------------------------
_MODELS = []          # models built since the last Saver.restore: what TF's default graph would hold


def register_model(model):
    _MODELS.append(model)


class _Saver:
    """tf.train.Saver: restore(sess, path) loads reference-named variables into every model built
    so far -- from `path`.npz / `path` (an .npz of name -> array) or a TensorFlow V2 checkpoint
    prefix (n2nmn_amd.tf_checkpoint)."""

    def __init__(self, *args, **kwargs):
        pass

    def restore(self, sess, path):
        import os
        if os.path.exists(path + '.npz') or path.endswith('.npz'):
            z = np.load(path if path.endswith('.npz') else path + '.npz')
            weights = {k: z[k] for k in z.files}
        else:
            from . import tf_checkpoint
            weights = tf_checkpoint.read_checkpoint(path, skip_missing=True)
        if not _MODELS:
            raise RuntimeError('Saver.restore: no model has been built')
        for m in _MODELS:
            m.load_weights(weights)

    def save(self, sess, path, **kwargs):
        raise NotImplementedError('the inference drop-in does not write TensorFlow checkpoints')


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _NoOp:
    """tf.global_variables_initializer() (exp_shapes/eval_shapes.py:135)"""


def _config(**kwargs):
    return _Namespace(**kwargs)


tf = _Namespace(
    Session=lambda config=None, **kw: Session(),
    ConfigProto=_config, GPUOptions=_config,
    placeholder=placeholder,
    int32='int32', int64='int64', float32='float32', float64='float64', bool='bool',
    train=_Namespace(Saver=_Saver),
    global_variables_initializer=lambda: _NoOp(),
)


def to_numpy(t):
    if hasattr(t, 'detach'):
        return t.detach().cpu().numpy()
    return np.asarray(t)
