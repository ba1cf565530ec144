"""Eager stand-in for the slice of TensorFlow-Fold 0.0.1 that models_*/nmn3_model.py uses -- TEST
INFRASTRUCTURE (see ../tensorflow/__init__.py).

Fold's contract: every example's tree is evaluated independently, invalid
examples give the constant, output row i belongs to input i; operations are batched per (block,
depth).  `Compiler._fold_eval` does exactly that: it instantiates each example's tree, then runs
depth 1, 2, ... and calls every `td.Function`'s python callable ONCE per depth on the stacked inputs
of all its instances at that depth -- so the reference's module code sees batches of size Nb >= 1
with `time_idx` / `batch_idx` tensors of shape [Nb], like under Loom.
"""
from __future__ import annotations

import numpy as _np
import torch as _t
import tensorflow as _tf


class _Block:
    def __rshift__(self, other):
        return _Pipe(self, other)

    def instantiate(self, x, plan):
        raise NotImplementedError


class _Pipe(_Block):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def instantiate(self, x, plan):
        return self.b.instantiate(self.a.instantiate(x, plan), plan)


class _Const:
    """A value known without running anything (depth 0)."""
    depth = 0

    def __init__(self, value):
        self.value = value


class _Call:
    """One instance of a td.Function, waiting for its inputs."""

    def __init__(self, fn_block, args):
        self.fn_block, self.args = fn_block, args
        self.depth = 1 + max([a.depth for a in args] + [0])
        self.value = None


class Scalar(_Block):
    def __init__(self, dtype='float32'):
        self.dtype = dtype

    def instantiate(self, x, plan):
        return _Const(_t.tensor(x, dtype=_tf._dt(self.dtype)))


class Record(_Block):
    def __init__(self, fields):
        self.fields = list(fields.items()) if isinstance(fields, dict) else list(fields)

    def instantiate(self, x, plan):
        return tuple(block.instantiate(x[key], plan) for key, block in self.fields)


class Function(_Block):
    def __init__(self, fn):
        self.fn = fn

    def instantiate(self, x, plan):
        args = list(x) if isinstance(x, tuple) else [x]
        call = _Call(self, args)
        plan.append(call)
        return call


class ScopedLayer(Function):
    """td.ScopedLayer(layer_fn, name_or_scope): the callable runs under its own variable scope,
    variables created on the first call and reused afterwards (models_shapes/nmn3_model.py:60-89)."""

    def __init__(self, layer_fn, name_or_scope=None):
        self._layer_fn = layer_fn
        name = name_or_scope or getattr(layer_fn, '__name__', 'scoped_layer')
        with _tf.variable_scope(name) as vs:          # scope is fixed where the layer is declared
            self._scope = _tf.VariableScope(vs.name, False)
        self._built = False
        super().__init__(self._call)

    def _call(self, *args):
        with _tf.variable_scope(self._scope, reuse=self._built):
            self._built = True
            return self._layer_fn(*args)


class GetItem:
    def __init__(self, key):
        self.key = key

    def __call__(self, x):
        return x[self.key]


class OneOf(_Block):
    def __init__(self, key_fn, case_blocks, pre_block=None):
        self.key_fn, self.cases = key_fn, dict(case_blocks)

    def instantiate(self, x, plan):
        return self.cases[self.key_fn(x)].instantiate(x, plan)


class PyObjectType:
    pass


class TensorType:
    def __init__(self, shape, dtype='float32'):
        self.shape, self.dtype = shape, dtype


class ForwardDeclaration:
    def __init__(self, input_type=None, output_type=None):
        self._target = None

    def __call__(self):
        decl = self

        class _Ref(_Block):
            def instantiate(self, x, plan):
                return decl._target.instantiate(x, plan)
        return _Ref()

    def resolve_to(self, block):
        self._target = block


class Void(_Block):
    def instantiate(self, x, plan):
        return None


class FromTensor(_Block):
    def __init__(self, value):
        self.value = value

    def instantiate(self, x, plan):
        return _Const(_tf._T(_np.asarray(self.value)))


class _Lazy:
    """A graph-mode tensor that depends on the Fold input: evaluated by tf.Session.run(feed_dict).
    Supports the one arithmetic form the reference applies to it (models_vqa/nmn3_model.py:112)."""

    def __init__(self, fn):
        self._fn = fn

    def _fold_eval(self, feed_dict):
        return self._fn(feed_dict)

    def __add__(self, other):
        return _Lazy(lambda fd: self._fold_eval(fd) + (other._fold_eval(fd) if isinstance(other, _Lazy) else other))

    __radd__ = __add__


class _Output(_Lazy):
    def __init__(self, compiler):
        super().__init__(lambda fd: compiler._fold_eval(fd[compiler.loom_input_tensor]))


class Compiler:
    def __init__(self, root):
        self.root = root
        self.loom_input_tensor = object()
        self.output_tensors = [_Output(self)]
        self.batch_sizes = []          # (function name, depth, Nb) of the last evaluation

    @classmethod
    def create(cls, root_block_like):
        return cls(root_block_like)

    def build_feed_dict(self, examples):
        return {self.loom_input_tensor: list(examples)}

    def _fold_eval(self, examples):
        plan = []
        roots = [self.root.instantiate(e, plan) for e in examples]
        self.batch_sizes = []
        max_depth = max([c.depth for c in plan] + [0])
        for depth in range(1, max_depth + 1):
            groups = {}
            for c in plan:
                if c.depth == depth:
                    groups.setdefault(id(c.fn_block), []).append(c)
            for calls in groups.values():
                block = calls[0].fn_block
                nargs = len(calls[0].args)
                stacked = [_t.stack([c.args[k].value for c in calls]) for k in range(nargs)]
                out = block.fn(*stacked)
                assert out.shape[0] == len(calls)
                self.batch_sizes.append((getattr(block.fn, '__name__', '?'), depth, len(calls)))
                for i, c in enumerate(calls):
                    c.value = out[i]
        return _t.stack([r.value for r in roots])
