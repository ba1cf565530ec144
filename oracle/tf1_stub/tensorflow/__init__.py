"""Eager stand-in for the slice of TensorFlow 1.0.0 that ronghanghu/n2nmn calls -- TEST INFRASTRUCTURE.

Purpose: let the reference's OWN model code (models_*/nmn3_modules.py, nmn3_netgen_att.py,
nmn3_model.py, question_prior_net.py, shapes_convnet.py, util/cnn.py, util/empty_safe_conv.py) run
unmodified in this container, where TensorFlow cannot be installed, so that the float path of the
oracle and of the HIP kernels is pinned to numbers produced by reference code
(tests/golden/make_float_golden.py puts this directory in front of sys.path, then imports the
reference from /root/reference).

How: every `tf.*` op executes immediately on a torch CPU tensor (float64 by default), so graph
construction IS evaluation, and torch autograd gives the gradients of whatever the reference code
computed.  Only the ops the reference calls exist (list: DESIGN.md section 6).  What is RESTATED here,
because it is TensorFlow library code and not reference code: BasicLSTMCell / MultiRNNCell /
DropoutWrapper, dynamic_rnn, raw_rnn, variable scoping, and each primitive op's arithmetic
(SURVEY.md Appendix A.1-A.4).

Never imported by the product (n2nmn_amd/), by bench.py or by the GPU tests.
"""
from __future__ import annotations

import builtins as _b
import contextlib
import math
import types
from collections import namedtuple

import numpy as _np
import torch as _t

# ----------------------------------------------------------------------------------------------
# configuration hooks (set by the fixture generator, never by reference code)
# ----------------------------------------------------------------------------------------------


class _Config:
    float_dtype = _t.float64       # what tf.float32 maps to (float64 = ground truth run)
    none_dim = 1                   # size given to `None` dimensions of tf.placeholder
    requires_grad = False          # created variables track gradients
    preloaded = {}                 # variable name -> numpy array (values for get_variable)
    strict_preload = True          # a variable missing from `preloaded` is an error
    multinomial_uniforms = None    # callable(n_rows) -> uniforms in [0,1) for tf.multinomial
    dropout_masks = None           # callable(shape, keep_prob) -> {0,1} mask for dropout


config = _Config()

float32 = 'float32'
float64 = 'float64'
int32 = 'int32'
int64 = 'int64'
bool = 'bool'  # noqa: A001  (tf.bool)
newaxis = None


def _dt(dtype):
    if dtype is None:
        return None
    if isinstance(dtype, _t.dtype):
        return dtype
    name = getattr(dtype, 'name', dtype)
    return {'float32': config.float_dtype, 'float64': _t.float64, 'int32': _t.int32,
            'int64': _t.int64, 'bool': _t.bool}[name]


class _Shape(list):
    def as_list(self):
        return list(self)


def _get_shape(self):
    return _Shape(int(s) for s in self.shape)


# static-shape API of tf.Tensor on torch tensors (this process only ever runs fixture generation)
_t.Tensor.get_shape = _get_shape
_t.Tensor.set_shape = lambda self, shape: None


def _T(x, like=None):
    """python scalar / numpy / list -> tensor (dtype of `like` for python floats)."""
    if isinstance(x, _t.Tensor):
        return x
    if isinstance(x, (float, int)) and not isinstance(x, (_np.bool_,)) and like is not None:
        return _t.tensor(x, dtype=like.dtype)
    if isinstance(x, float):
        return _t.tensor(x, dtype=config.float_dtype)
    if isinstance(x, _np.ndarray):
        if x.dtype in (_np.float32, _np.float64):
            return _t.as_tensor(x.astype(_np.float64)).to(config.float_dtype)
        return _t.as_tensor(x)
    if isinstance(x, (list, tuple)) and any(isinstance(v, _t.Tensor) for v in x):
        return _t.stack([_T(v) for v in x])
    return _t.as_tensor(x)


def _shape_arg(s):
    if isinstance(s, _t.Tensor):
        return [int(v) for v in s.tolist()]
    if isinstance(s, (int, _np.integer)):
        return [int(s)]
    return [int(v) for v in s]


def convert_to_tensor(value, dtype=None, name=None):
    t = _T(value)
    return t.to(_dt(dtype)) if dtype is not None else t


def constant(value, dtype=None, shape=None, name=None):
    t = convert_to_tensor(value, dtype)
    return t.reshape(_shape_arg(shape)) if shape is not None else t


def placeholder(dtype, shape=None, name=None):
    dims = [config.none_dim if s is None else int(s) for s in (shape or [])]
    return _t.zeros(dims, dtype=_dt(dtype))


def shape(x):
    return _Shape(int(s) for s in _T(x).shape)


def size(x):
    return _T(x).numel()


def cast(x, dtype):
    return _T(x).to(_dt(dtype))


def reshape(x, shp):
    return _T(x).reshape(_shape_arg(shp))


def tile(x, multiples):
    return _T(x).repeat(*_shape_arg(multiples))


def concat(values, axis):
    return _t.cat([_T(v) for v in values], dim=axis)


def stack(values, axis=0):
    return _t.stack([_T(v) for v in values], dim=axis)


def zeros(shp, dtype=float32):
    return _t.zeros(_shape_arg(shp), dtype=_dt(dtype))


def ones(shp, dtype=float32):
    return _t.ones(_shape_arg(shp), dtype=_dt(dtype))


def zeros_like(x):
    return _t.zeros_like(_T(x))


def ones_like(x):
    return _t.ones_like(_T(x))


def range(*args, dtype=int32):  # noqa: A001
    return _t.arange(*[int(a) for a in args], dtype=_dt(dtype))


def linspace(start, stop, num):
    return _t.linspace(float(start), float(stop), int(num), dtype=config.float_dtype)


def stop_gradient(x):
    return _T(x).detach()


def _axes(axis):
    if axis is None:
        return None
    return tuple(axis) if isinstance(axis, (list, tuple)) else (int(axis),)


def reduce_sum(x, axis=None, keep_dims=False):
    x = _T(x)
    return x.sum() if axis is None else x.sum(dim=_axes(axis), keepdim=keep_dims)


def reduce_mean(x, axis=None, keep_dims=False):
    x = _T(x)
    return x.mean() if axis is None else x.mean(dim=_axes(axis), keepdim=keep_dims)


def reduce_max(x, axis=None, keep_dims=False):
    # gradient: split equally between tied extrema (TF _MinOrMaxGrad) == torch.amax
    x = _T(x)
    return x.max() if axis is None else _t.amax(x, dim=_axes(axis), keepdim=keep_dims)


def reduce_min(x, axis=None, keep_dims=False):
    x = _T(x)
    return x.min() if axis is None else _t.amin(x, dim=_axes(axis), keepdim=keep_dims)


def reduce_all(x, axis=None, keep_dims=False):
    x = _T(x)
    if axis is None:
        return x.all()
    for a in sorted(_axes(axis), reverse=True):
        x = x.all(dim=a, keepdim=keep_dims)
    return x


def reduce_any(x, axis=None, keep_dims=False):
    x = _T(x)
    if axis is None:
        return x.any()
    for a in sorted(_axes(axis), reverse=True):
        x = x.any(dim=a, keepdim=keep_dims)
    return x


class _MinMaxFirst(_t.autograd.Function):
    """tf.minimum / tf.maximum: on ties the whole gradient goes to the FIRST argument
    (TF _MinimumGrad uses x <= y, _MaximumGrad x >= y)."""

    @staticmethod
    def forward(ctx, x, y, is_min):
        xm = (x <= y) if is_min else (x >= y)
        ctx.save_for_backward(xm)
        ctx.shapes = (x.shape, y.shape)
        return _t.where(xm, x, y)

    @staticmethod
    def backward(ctx, g):
        (xm,) = ctx.saved_tensors
        gx = _t.where(xm, g, _t.zeros_like(g))
        gy = g - gx
        return gx.sum_to_size(ctx.shapes[0]), gy.sum_to_size(ctx.shapes[1]), None


def minimum(x, y):
    x = _T(x, like=y if isinstance(y, _t.Tensor) else None)
    y = _T(y, like=x)
    x, y = _t.broadcast_tensors(x, y)
    return _MinMaxFirst.apply(x, y, True)


def maximum(x, y):
    x = _T(x, like=y if isinstance(y, _t.Tensor) else None)
    y = _T(y, like=x)
    x, y = _t.broadcast_tensors(x, y)
    return _MinMaxFirst.apply(x, y, False)


def tanh(x):
    return _t.tanh(_T(x))


def sigmoid(x):
    return _t.sigmoid(_T(x))


def log(x):
    return _t.log(_T(x))


def exp(x):
    return _t.exp(_T(x))


def matmul(a, b):
    return _T(a) @ _T(b)


def tensordot(a, b, axes):
    a, b = _T(a), _T(b)
    assert axes == 1
    # integer-safe: sum_k a[..., k] * b[k, ...]
    k = a.shape[-1]
    out = None
    for i in _b.range(k):
        term = a[..., int(i)].reshape(a.shape[:-1] + (1,) * (b.dim() - 1)) * b[int(i)]
        out = term if out is None else out + term
    return out


def greater_equal(x, y):
    return _T(x) >= _T(y, like=_T(x))


def greater(x, y):
    return _T(x) > _T(y, like=_T(x))


def less(x, y):
    return _T(x) < _T(y, like=_T(x))


def equal(x, y):
    return _T(x) == _T(y, like=_T(x))


def logical_or(x, y):
    return _t.logical_or(_T(x), _T(y))


def logical_and(x, y):
    return _t.logical_and(_T(x), _T(y))


def where(cond, x=None, y=None):
    cond = _T(cond)
    x, y = _T(x), _T(y)
    if cond.dim() == 1 and x.dim() > 1:        # TF: rank-1 condition selects rows
        cond = cond.reshape([-1] + [1] * (x.dim() - 1))
    return _t.where(cond, x, y)


def argmax(x, axis=None, dimension=None):
    a = axis if axis is not None else dimension
    return _t.argmax(_T(x), dim=int(a))            # first maximal index, as tf.argmax


def gather(params, indices):
    return _T(params)[_T(indices).long()]


def gather_nd(params, indices):
    idx = _T(indices).long()
    return _T(params)[tuple(idx[..., i] for i in _b.range(idx.shape[-1]))]


def add_n(values):
    out = values[0]
    for v in values[1:]:
        out = out + v
    return out


def multinomial(logits, num_samples):
    """Inverse-CDF draw from softmax(logits) on uniforms supplied by the fixture generator (TF's own
    RNG stream is not reproducible; the C-ABI takes the uniforms as an input for the same reason)."""
    assert num_samples == 1 and config.multinomial_uniforms is not None
    logits = _T(logits)
    p = _t.softmax(logits, dim=1)
    cdf = _t.cumsum(p, dim=1)
    u = _T(config.multinomial_uniforms(logits.shape[0])).to(cdf.dtype).reshape(-1, 1)
    s = (cdf <= u * cdf[:, -1:]).sum(dim=1).clamp(max=logits.shape[1] - 1)
    return s.reshape(-1, 1)


def check_numerics(x, msg):
    return x


def assert_positive(x):
    return x


def cond(pred, fn1, fn2):
    return fn1() if _b.bool(pred) else fn2()


@contextlib.contextmanager
def device(name):
    yield


def RegisterGradient(name):
    return lambda fn: fn


class _Graph:
    @contextlib.contextmanager
    def gradient_override_map(self, m):
        yield


def get_default_graph():
    return _Graph()


class GraphKeys:
    REGULARIZATION_LOSSES = 'regularization_losses'
    TRAINABLE_VARIABLES = 'trainable_variables'


_collections = {}


def add_to_collection(name, value):
    _collections.setdefault(name, []).append(value)


def get_collection(name):
    return list(_collections.get(name, []))


# ----------------------------------------------------------------------------------------------
# variables and scopes (TF 1.0.0: reuse=True is inherited by sub-scopes; None/False inherit)
# ----------------------------------------------------------------------------------------------
class VariableScope:
    def __init__(self, name, reuse):
        self.name = name
        self.reuse = reuse

    def reuse_variables(self):
        self.reuse = True


_scope_stack = [VariableScope('', False)]
_variables = {}          # full name -> tensor (creation order preserved)


def reset_default_graph():
    del _scope_stack[1:]
    _scope_stack[0].reuse = False
    _variables.clear()
    _collections.clear()


def get_variable_scope():
    return _scope_stack[-1]


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None):
    cur = _scope_stack[-1]
    if isinstance(name_or_scope, VariableScope):     # re-entering a captured scope: absolute name
        full = name_or_scope.name
        inherited = name_or_scope.reuse or cur.reuse
    else:
        name = name_or_scope if name_or_scope is not None else default_name
        full = (cur.name + '/' + name) if cur.name else name
        inherited = cur.reuse
    vs = VariableScope(full, True if reuse else inherited)
    _scope_stack.append(vs)
    try:
        yield vs
    finally:
        _scope_stack.pop()


def _xavier(shp, dtype, conv):
    if len(shp) == 4:
        rf = shp[0] * shp[1]
        fan_in, fan_out = rf * shp[2], rf * shp[3]
    elif len(shp) == 2:
        fan_in, fan_out = shp
    else:
        fan_in = fan_out = shp[0]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (_t.rand(shp, dtype=dtype) * 2 - 1) * lim


def constant_initializer(value=0.):
    return lambda shp, dtype: _t.full(shp, float(value), dtype=dtype)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    vs = _scope_stack[-1]
    full = (vs.name + '/' + name) if vs.name else name
    if vs.reuse:
        if full not in _variables:
            raise ValueError('Variable %s does not exist (reuse=True)' % full)
        return _variables[full]
    if full in _variables:
        raise ValueError('Variable %s already exists (reuse not set)' % full)
    shp = _shape_arg(shape)
    fdt = config.float_dtype
    if full in config.preloaded:
        val = _t.as_tensor(_np.asarray(config.preloaded[full], _np.float64)).to(fdt).clone()
        if list(val.shape) != shp:
            raise ValueError('preloaded %s has shape %s, reference code asks for %s'
                             % (full, list(val.shape), shp))
    elif config.strict_preload:
        raise KeyError('reference code created variable %r which the fixture generator did not '
                       'provide' % full)
    elif initializer is not None:
        val = initializer(shp, fdt)
    else:
        val = _xavier(shp, fdt, False)
    val.requires_grad_(config.requires_grad)
    val.op = types.SimpleNamespace(name=full)
    _variables[full] = val
    return val


def trainable_variables():
    return list(_variables.values())


def global_variables():
    return list(_variables.values())


# ----------------------------------------------------------------------------------------------
# TensorArray (functional: write returns the array)
# ----------------------------------------------------------------------------------------------
class TensorArray:
    def __init__(self, dtype, size, infer_shape=True, **kw):
        self._items = [None] * int(size)

    def write(self, index, value):
        new = TensorArray.__new__(TensorArray)
        new._items = list(self._items)
        new._items[int(index)] = _T(value)
        return new

    def read(self, index):
        return self._items[int(index)]

    def stack(self):
        return _t.stack(self._items)


# ----------------------------------------------------------------------------------------------
# tf.nn
# ----------------------------------------------------------------------------------------------
def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _conv2d(input, filter, strides, padding, use_cudnn_on_gpu=None, data_format=None, name=None):
    x, w = _T(input), _T(filter)                     # NHWC, [kh, kw, in, out]; cross-correlation
    sh, sw = int(strides[1]), int(strides[2])
    xc = x.permute(0, 3, 1, 2)
    if padding == 'SAME':
        pt, pb = _same_pad(x.shape[1], w.shape[0], sh)
        pl, pr = _same_pad(x.shape[2], w.shape[1], sw)
        xc = _t.nn.functional.pad(xc, (pl, pr, pt, pb))
    y = _t.nn.functional.conv2d(xc, w.permute(3, 2, 0, 1), stride=(sh, sw))
    return y.permute(0, 2, 3, 1)


def _softmax(logits, dim=-1, name=None):
    return _t.softmax(_T(logits), dim=dim)          # max-subtracted, like TF's kernel


def _l2_normalize(x, dim, epsilon=1e-12, name=None):
    x = _T(x)
    ss = (x * x).sum(dim=dim, keepdim=True)
    return x * _t.rsqrt(_t.clamp(ss, min=epsilon))   # x * rsqrt(max(sum x^2, eps))


def _xw_plus_b(x, weights, biases, name=None):
    return _T(x) @ _T(weights) + _T(biases)


def _embedding_lookup(params, ids):
    return _T(params)[_T(ids).long()]


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    x = _T(x)
    if config.dropout_masks is None:
        raise RuntimeError('tf.nn.dropout needs config.dropout_masks (masks are inputs)')
    mask = _T(config.dropout_masks(tuple(x.shape), float(keep_prob))).to(x.dtype)
    return x * mask / float(keep_prob)               # TF: x / keep_prob * floor(keep_prob + u)


def _sparse_softmax_ce(logits=None, labels=None, name=None):
    logits = _T(logits)
    return _t.logsumexp(logits, dim=1) - logits.gather(1, _T(labels).long().reshape(-1, 1))[:, 0]


LSTMStateTuple = namedtuple('LSTMStateTuple', ('c', 'h'))


class _RNNCell:
    pass


class BasicLSTMCell(_RNNCell):
    """tf.contrib.rnn.BasicLSTMCell of TF 1.0.0: z = [x, h] W + b with
    variables `basic_lstm_cell/{weights,biases}`, gate order i, j, f, o, forget_bias added at run
    time."""

    def __init__(self, num_units, forget_bias=1.0, state_is_tuple=True):
        self._n = int(num_units)
        self._fb = float(forget_bias)

    def __call__(self, inputs, state, scope=None):
        with variable_scope(scope or 'basic_lstm_cell'):
            c, h = state
            xh = _t.cat([_T(inputs), h], dim=1)
            W = get_variable('weights', [xh.shape[1], 4 * self._n])
            b = get_variable('biases', [4 * self._n], initializer=constant_initializer(0.))
            z = xh @ W + b
            i, j, f, o = _t.split(z, self._n, dim=1)
            new_c = c * _t.sigmoid(f + self._fb) + _t.sigmoid(i) * _t.tanh(j)
            new_h = _t.tanh(new_c) * _t.sigmoid(o)
            return new_h, LSTMStateTuple(new_c, new_h)

    def zero_state(self, batch_size, dtype):
        z = _t.zeros([int(batch_size), self._n], dtype=_dt(dtype))
        return LSTMStateTuple(z, z.clone())


class DropoutWrapper(_RNNCell):
    def __init__(self, cell, input_keep_prob=1.0, output_keep_prob=1.0, seed=None):
        self._cell, self._okp = cell, float(output_keep_prob)

    def __call__(self, inputs, state, scope=None):
        out, new_state = self._cell(inputs, state, scope)
        if self._okp < 1.0:
            out = _dropout(out, self._okp)
        return out, new_state

    def zero_state(self, batch_size, dtype):
        return self._cell.zero_state(batch_size, dtype)


class MultiRNNCell(_RNNCell):
    """TF 1.0.0: layer i runs under scope `multi_rnn_cell/cell_<i>` (separate variables per layer
    even when the python list repeats one cell object)."""

    def __init__(self, cells, state_is_tuple=True):
        self._cells = list(cells)

    def __call__(self, inputs, state, scope=None):
        with variable_scope(scope or 'multi_rnn_cell'):
            cur, new_states = inputs, []
            for i, cell in enumerate(self._cells):
                with variable_scope('cell_%d' % i):
                    cur, ns = cell(cur, state[i])
                    new_states.append(ns)
        return cur, tuple(new_states)

    def zero_state(self, batch_size, dtype):
        return tuple(c.zero_state(batch_size, dtype) for c in self._cells)


def _map_state(fn, *states):
    s0 = states[0]
    if isinstance(s0, LSTMStateTuple):
        return LSTMStateTuple(*[fn(*[s[k] for s in states]) for k in (0, 1)])
    if isinstance(s0, tuple):
        return tuple(_map_state(fn, *[s[k] for s in states]) for k in _b.range(len(s0)))
    return fn(*states)


def _dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, dtype=None,
                 time_major=False, scope=None):
    """tf.nn.dynamic_rnn (A.2): past sequence_length the output row is zero and the state is
    carried through unchanged."""
    assert time_major
    x = _T(inputs)
    T, N = x.shape[0], x.shape[1]
    seq = _T(sequence_length) if sequence_length is not None else None
    with variable_scope(scope or 'rnn') as vs:
        state = initial_state if initial_state is not None else cell.zero_state(N, dtype)
        outs = []
        for t in _b.range(T):
            out, new_state = cell(x[int(t)], state)
            if seq is not None:
                act = (int(t) < seq).reshape(N, 1)
                out = _t.where(act, out, _t.zeros_like(out))
                state = _map_state(lambda n, o: _t.where(act, n, o), new_state, state)
            else:
                state = new_state
            outs.append(out)
            vs.reuse_variables()
            _scope_stack[-1].reuse = True
    return _t.stack(outs), state


def _raw_rnn(cell, loop_fn, parallel_iterations=None, swap_memory=False, scope=None):
    """tf.nn.raw_rnn (A.3)."""
    with variable_scope(scope or 'rnn'):
        time = 0
        finished, next_input, state, emit_structure, loop_state = loop_fn(time, None, None, None)
        emit_ta = []
        while not _b.bool(_T(finished).all()):
            out, cell_state = cell(next_input, state)
            _scope_stack[-1].reuse = True
            time += 1
            finished, next_input, state, emit, loop_state = loop_fn(time, out, cell_state, loop_state)
            emit_ta.append(emit)
    return emit_ta, state, loop_state


nn = types.SimpleNamespace(
    conv2d=_conv2d, softmax=_softmax, l2_normalize=_l2_normalize, xw_plus_b=_xw_plus_b,
    embedding_lookup=_embedding_lookup, dropout=_dropout, relu=lambda x, name=None: _t.relu(_T(x)),
    bias_add=lambda v, b, name=None: _T(v) + _T(b),
    l2_loss=lambda v, name=None: 0.5 * (_T(v) * _T(v)).sum(),
    sparse_softmax_cross_entropy_with_logits=_sparse_softmax_ce,
    dynamic_rnn=_dynamic_rnn, raw_rnn=_raw_rnn,
    sigmoid=sigmoid, tanh=tanh,
    max_pool=lambda value, ksize, strides, padding, name=None: _max_pool(value, ksize, strides, padding),
)


def _max_pool(value, ksize, strides, padding):
    x = _T(value).permute(0, 3, 1, 2)
    kh, kw, sh, sw = int(ksize[1]), int(ksize[2]), int(strides[1]), int(strides[2])
    if padding == 'SAME':
        pt, pb = _same_pad(x.shape[2], kh, sh)
        pl, pr = _same_pad(x.shape[3], kw, sw)
        x = _t.nn.functional.pad(x, (pl, pr, pt, pb), value=float('-inf'))
    return _t.nn.functional.max_pool2d(x, (kh, kw), (sh, sw)).permute(0, 2, 3, 1)


contrib = types.SimpleNamespace(
    rnn=types.SimpleNamespace(BasicLSTMCell=BasicLSTMCell, DropoutWrapper=DropoutWrapper,
                              MultiRNNCell=MultiRNNCell, LSTMStateTuple=LSTMStateTuple),
    layers=types.SimpleNamespace(
        xavier_initializer=lambda: (lambda shp, dtype: _xavier(shp, dtype, False)),
        xavier_initializer_conv2d=lambda: (lambda shp, dtype: _xavier(shp, dtype, True))),
)


# ----------------------------------------------------------------------------------------------
# Session: values already exist; `run` only converts (and evaluates Fold outputs, see
# tensorflow_fold stub)
# ----------------------------------------------------------------------------------------------
class Session:
    def __init__(self, *a, **kw):
        pass

    def run(self, fetches, feed_dict=None):
        def one(f):
            if hasattr(f, '_fold_eval'):
                return f._fold_eval(feed_dict or {})
            return f
        if isinstance(fetches, (list, tuple)):
            return type(fetches)(one(f) for f in fetches)
        return one(fetches)


# ----------------------------------------------------------------------------------------------
# training pieces used by exp_clevr/train_clevr_*_gt_layout.py loss blocks (TF library, restated:
# SURVEY Appendix A.4 clip_by_norm; TF 1.0.0 AdamOptimizer defaults)
# ----------------------------------------------------------------------------------------------
def clip_by_norm(t, clip_norm, axes=None, name=None):
    t = _T(t)
    norm = _t.sqrt((t * t).sum())
    return t * float(clip_norm) / _t.clamp(norm, min=float(clip_norm))


def Variable(initial_value, trainable=True, dtype=None, name=None):
    v = convert_to_tensor(initial_value, dtype=dtype).clone()
    if trainable:
        raise NotImplementedError('stub: trainable variables come from get_variable')
    return v


class _DeferredOp:
    """Graph-mode side effect: happens when the step runs, not where the python line is."""

    def __init__(self, fn):
        self._fn = fn

    def run(self):
        with _t.no_grad():
            self._fn()


def assign_add(ref, value):
    value = _T(value).detach().clone()
    return _DeferredOp(lambda: ref.add_(value))


@contextlib.contextmanager
def control_dependencies(ops):
    yield


class _AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def compute_gradients(self, loss):
        vs = trainable_variables()
        gs = _t.autograd.grad(loss, vs, allow_unused=True)
        self.last_raw_gradients = [(g if g is not None else _t.zeros_like(v), v)
                                   for g, v in zip(gs, vs)]
        # under Fold every module is part of the graph, so an operator absent from the minibatch has
        # a zero gradient there, not None
        return [(g if g is not None else _t.zeros_like(v), v) for g, v in zip(gs, vs)]

    def apply_gradients(self, grads_and_vars):
        def step():
            self.t += 1
            lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
            for g, v in grads_and_vars:
                k = v.op.name
                m = self.m.setdefault(k, _t.zeros_like(v))
                s = self.v.setdefault(k, _t.zeros_like(v))
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                s.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                v.sub_(lr_t * m / (_t.sqrt(s) + self.eps))
        return _DeferredOp(step)


train = types.SimpleNamespace(AdamOptimizer=_AdamOptimizer)
