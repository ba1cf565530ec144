"""Stand-ins for the TensorFlow objects the reference's driver loops touch: placeholders, fetchable
model attributes and a session with `partial_run_setup` / `partial_run` (exp_clevr/eval_clevr.py:72-75,
103-132), plus -- for the training drivers (exp_clevr/train_clevr_gt_layout.py:60-223,
train_clevr_rl_gt_layout.py) -- the loss / optimizer / summary / saver names of n2nmn_amd.runtime_train.
They carry no TensorFlow graph: a `partial_run` that fetches a phase-1 attribute launches phase 1 on the GPU,
one that fetches `scores` launches phase 2, one that fetches the script's `train_step` runs one Trainer
iteration (forward, backward, clip, Adam).
"""
from __future__ import annotations

import numpy as np

from . import runtime_train as _rt
from .runtime_train import Sym


class Placeholder(Sym):
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


class Fetch(Sym):
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
        self.env = {}           # id(graph node) -> value computed by this run (runtime_train.evaluate)
        # the `train_step` constant among the declared fetches, if any: a model face that trains under dropout
        # draws ONE set of masks per handle, in phase 1, for the step that runs in phase 2 (TensorFlow evaluates
        # a dropout op once per partial_run handle)
        self.train_const = next((f for f in self.allowed_fetches if isinstance(f, _rt.Const) and f.deps), None)

    def train_step(self):
        c = self.train_const
        if c is None:
            return None
        if getattr(c, '_step', None) is None:
            c._step = _rt.TrainStep(c)
        return c._step


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
        # a fetched `train_step` (a constant under tf.control_dependencies([solver_op, ...]),
        # train_clevr_gt_layout.py:126-130) runs the iteration FIRST: the scores and losses fetched beside
        # it are the training forward's, as in TensorFlow's single session run
        steps = {}
        for f in flist:
            if isinstance(f, _rt.Const) and f.deps:
                if getattr(f, '_step', None) is None:
                    f._step = _rt.TrainStep(f)
                steps[id(f)] = f._step.run(handle)
        out = []
        for f in flist:
            if id(f) in steps:
                out.append(steps[id(f)])
            elif isinstance(f, Fetch):
                out.append(f.owner._fetch(f, handle))
                self.last[f.name] = out[-1]
            elif isinstance(f, _rt.Const):
                out.append(np.asarray(f.value))
            elif isinstance(f, Sym):
                out.append(self._evaluate(f, handle))
            else:
                raise TypeError('cannot fetch %r' % (f,))
        return out[0] if single else out

    def _evaluate(self, node, handle):
        """a loss expression of the script, from what this run computed (the matched terms come from the
        device; anything else is numpy arithmetic over fetched values and fed placeholders)"""
        env = dict(handle.env)
        for k, v in handle.feeds.items():
            env[id(k)] = np.asarray(v) if not hasattr(v, 'handle') else v
        seen = set()

        def bind(x):
            if id(x) in seen:
                return
            seen.add(id(x))
            if isinstance(x, Fetch) and id(x) not in env:
                env[id(x)] = x.owner._fetch(x, handle)
            elif isinstance(x, _rt.Variable) and id(x) not in env:
                env[id(x)] = np.float32(x.read())
            elif isinstance(x, _rt.Op) and id(x) not in env:
                for a in x.args:
                    bind(a)
        bind(node)
        return _rt.evaluate(node, env)

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, _NoOp):
            # sess.run(tf.global_variables_initializer()): every model built so far that has no weights yet
            # gets the reference's declared initial values (runtime_train.initial_weights); a Saver.restore
            # afterwards overwrites them (train_clevr_rl_gt_layout.py:165-169)
            for m in _MODELS:
                init = getattr(m, 'initialize_variables', None)
                if init is not None:
                    init(_SEED[0])
            return None
        if isinstance(fetches, _rt._Merged):
            return _rt.SummaryValue({p.tag: float(np.asarray(resolve(p.tensor, feed_dict or {})))
                                     for p in fetches.parts})
        if isinstance(fetches, _rt.Variable):
            return np.float32(fetches.read())
        if isinstance(fetches, _rt.Op) and fetches.kind == 'assign':
            # sess.run(tf.assign(embedding_mat, glove_mat)) (exp_vqa/train_vqa2_gt_layout.py:166-169)
            var, value = fetches.args
            if not isinstance(var, _rt.ModelVariable):
                raise NotImplementedError('tf.assign: model variables only')
            value = np.asarray(resolve(value, feed_dict or {}), np.float32)
            if tuple(value.shape) != tuple(var.shape):
                raise ValueError('tf.assign(%s): shape %s, the variable is %s' % (var.name, value.shape, var.shape))
            w = {k: to_numpy(v) for k, v in var.model.get_weights().items()}
            w[var.name] = value
            var.model.load_weights(w)
            return value
        h = self.partial_run_setup(fetches, list((feed_dict or {}).keys()))
        return self.partial_run(h, fetches, feed_dict)


# ---- the handful of `tf.*` names the reference's driver scripts use (exp_clevr/eval_clevr.py:15-19,
# 72-75,90-91), so that a driver runs with `from n2nmn_amd.runtime import tf` in place of
# `import tensorflow as tf` and nothing else changed -----------------------------------------------
_MODELS = []          # models built since the last Saver.restore: what TF's default graph would hold


def register_model(model):
    _MODELS.append(model)


_SEED = [0]            # tf.set_random_seed(seed): seeds global_variables_initializer's draws


class _Saver:
    """tf.train.Saver: restore(sess, path) loads reference-named variables into every model built
    so far -- from `path`.npz / `path` (an .npz of name -> array) or a TensorFlow V2 checkpoint
    prefix (n2nmn_amd.tf_checkpoint); save(sess, path) writes a TensorFlow V2 checkpoint of the models'
    variables (and the script's own tf.Variables) under their reference names."""

    def __init__(self, var_list=None, *args, **kwargs):
        self.var_list = list(var_list) if var_list is not None else None

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

    def save(self, sess, path, global_step=None, write_meta_graph=True, **kwargs):
        """`<path>.index` + `<path>.data-00000-of-00001` (train_clevr_gt_layout.py:221-223).  Stored: every
        variable of every model built so far (float32, reference names) and the script's tf.Variables.
        NOT stored: the Adam slot variables TensorFlow's Saver would add (`<name>/Adam`, `<name>/Adam_1`,
        `beta1_power`, `beta2_power`) -- the optimiser state stays inside the library; the reference's eval
        and fine-tuning scripts restore model variables only (train_clevr_rl_gt_layout.py:168-169)."""
        from . import tf_checkpoint
        if global_step is not None:
            path = '%s-%d' % (path, int(global_step))
        tensors = {}
        for m in _MODELS:
            tensors.update({k: to_numpy(v) for k, v in m.get_weights().items()})
        for v in _rt._GLOBALS:
            tensors[v.name] = np.asarray(v.read(), np.float32)
        if self.var_list is not None:
            keep = {getattr(v, 'name', None) for v in self.var_list}
            tensors = {k: v for k, v in tensors.items() if k in keep}
        tf_checkpoint.write_checkpoint(path, tensors)
        return path


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _NoOp:
    """tf.global_variables_initializer() (exp_shapes/eval_shapes.py:135)"""


def _config(**kwargs):
    return _Namespace(**kwargs)


def _set_random_seed(seed):
    _SEED[0] = int(seed)


def _op(kind):
    def make(*args, **kw):
        kw.pop('name', None)
        return _rt.Op(kind, *args, **kw)
    return make


def _softmax_ce(_sentinel=None, labels=None, logits=None, name=None):
    if _sentinel is not None or labels is None or logits is None:
        raise ValueError('sparse_softmax_cross_entropy_with_logits takes named arguments (labels=, logits=)')
    return _rt.Op('softmax_ce', logits, labels)


_SCOPE = []


class _VariableScope:
    """tf.variable_scope(name, reuse=True): a prefix for tf.get_variable (train_vqa2_gt_layout.py:166-167)"""

    def __init__(self, name, reuse=None, **kw):
        self.name = name if isinstance(name, str) else getattr(name, 'name', str(name))

    def __enter__(self):
        _SCOPE.append(self.name.strip('/'))
        return self

    def __exit__(self, *exc):
        _SCOPE.pop()
        return False


def _get_variable(name, *args, **kwargs):
    """an EXISTING variable of a model built so far, by its scoped name (reuse=True: nothing is ever created here)"""
    full = '/'.join(_SCOPE + [name])
    for v in _rt.global_variables():
        if getattr(v, 'name', None) == full:
            return v
    raise ValueError('Variable %s does not exist (tf.get_variable under reuse=True)' % full)


tf = _Namespace(
    variable_scope=_VariableScope, get_variable=_get_variable,
    assign=lambda ref, value, **kw: _rt.Op('assign', ref, value),
    Session=lambda config=None, **kw: Session(),
    ConfigProto=_config, GPUOptions=_config,
    placeholder=placeholder,
    int32='int32', int64='int64', float32='float32', float64='float64', bool='bool',
    train=_Namespace(Saver=_Saver, AdamOptimizer=_rt.AdamOptimizer),
    global_variables_initializer=lambda: _NoOp(),
    # ---- the training drivers' loss block (n2nmn_amd.runtime_train) ----
    constant=lambda value, dtype=None, shape=None, name=None: _rt.Const(value, dtype),
    Variable=_rt.Variable,
    nn=_Namespace(sparse_softmax_cross_entropy_with_logits=_softmax_ce),
    reduce_mean=lambda x, axis=None, name=None: _rt.Op('mean', x),
    where=lambda c, x, y, name=None: _rt.Op('where', c, x, y),
    ones_like=lambda x, dtype=None, name=None: _rt.Op('ones_like', x),
    stop_gradient=lambda x, name=None: _rt.Op('stop_gradient', x),
    add_n=lambda xs, name=None: _rt.Op('add_n', *list(xs)),
    assign_add=lambda ref, value, **kw: _rt.Op('assign_add', ref, value),
    clip_by_norm=_rt.clip_by_norm,
    control_dependencies=lambda ops: _rt._ControlDependencies(ops),
    global_variables=_rt.global_variables,
    trainable_variables=lambda: [v for v in _rt.global_variables() if getattr(v, 'trainable', False)],
    get_default_graph=lambda: _Namespace(),
    set_random_seed=_set_random_seed,
    summary=_Namespace(FileWriter=_rt.FileWriter, scalar=lambda name, tensor, **kw: _rt._Scalar(name, tensor),
                       merge=lambda inputs, **kw: _rt._Merged(inputs)),
)


def to_numpy(t):
    if hasattr(t, 'detach'):
        return t.detach().cpu().numpy()
    return np.asarray(t)
