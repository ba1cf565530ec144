"""The symbolic surface the reference's TRAINING drivers touch, mapped onto one `Trainer` step
(exp_clevr/train_clevr_gt_layout.py:60-223, exp_clevr/train_clevr_rl_gt_layout.py:84-235).

The scripts build their loss with a dozen TensorFlow calls --

    tf.nn.sparse_softmax_cross_entropy_with_logits, tf.reduce_mean, tf.where, tf.ones_like, tf.stop_gradient,
    tf.add_n, tf.Variable / tf.assign_add (the REINFORCE baseline), + - * on tensors,
    tf.train.AdamOptimizer().compute_gradients / apply_gradients, tf.clip_by_norm, tf.control_dependencies,
    tf.constant, tf.summary.*, tf.global_variables_initializer, tf.global_variables, tf.train.Saver.save

-- and fetch `(scores, avg_sample_loss, train_step)` in the second `partial_run` of every iteration.
Nothing here differentiates or interprets a general graph.  The expression the script hands to
`compute_gradients` is MATCHED against the two loss graphs the reference has:

    behavioural cloning (:104-113)   mean(-log_seq_prob) + mean(CE(scores, labels)) + wd * l2_reg
    policy gradient   (rl :107-129)  mean(stop_gradient(final - baseline) * log_seq_prob) + mean(final)
                                     + lambda_entropy * entropy_reg + wd * l2_reg,
                                     final = where(validity, CE, invalid_expr_loss)

whose forward, backward, per-tensor clip-by-norm and Adam update are the C-ABI's n2nmn_train_forward /
n2nmn_train_backward / n2nmn_adam_step (n2nmn_amd.train.Trainer).  Any other loss graph is refused with
NotImplementedError naming the term that did not match -- there is no fallback that would train something
else than what the script wrote.
"""
from __future__ import annotations

import os
import struct
import time
from typing import Dict, List, Optional

import numpy as np


# ---------------------------------------------------------------------------------------------------
# symbolic values
# ---------------------------------------------------------------------------------------------------
class Sym:
    """A value of the script's graph.  Arithmetic builds Op nodes; nothing is computed here."""

    def __neg__(self):
        return Op('neg', self)

    def __add__(self, o):
        return Op('add', self, o)

    def __radd__(self, o):
        return Op('add', o, self)

    def __sub__(self, o):
        return Op('sub', self, o)

    def __rsub__(self, o):
        return Op('sub', o, self)

    def __mul__(self, o):
        return Op('mul', self, o)

    def __rmul__(self, o):
        return Op('mul', o, self)

    def __truediv__(self, o):
        return Op('div', self, o)

    # identity semantics (placeholders are dict keys in feed_dict; `v != baseline` in the RL script)
    __hash__ = object.__hash__

    def __eq__(self, o):
        return self is o

    def __ne__(self, o):
        return self is not o


class Op(Sym):
    def __init__(self, kind, *args, **attrs):
        self.kind, self.args, self.attrs = kind, args, attrs

    def __repr__(self):
        return '%s(%s)' % (self.kind, ', '.join(repr(a) for a in self.args))


_DEPS: List[list] = []           # stack of tf.control_dependencies contexts


class Const(Sym):
    """tf.constant(v).  One created under tf.control_dependencies([...]) runs those ops when fetched
    (train_clevr_gt_layout.py:126-130: `train_step`)."""

    def __init__(self, value, dtype=None):
        self.value, self.dtype = value, dtype
        self.deps = [d for ctx in _DEPS for d in ctx]

    def __repr__(self):
        return 'const(%r)' % (self.value,)

    def __bool__(self):
        return bool(self.value)


class Variable(Sym):
    """tf.Variable(initial, trainable=False): the REINFORCE baseline (rl :120)."""

    def __init__(self, initial_value, trainable=True, dtype=None, name=None):
        self.initial_value, self.trainable, self.dtype = initial_value, trainable, dtype
        self.name = name or ('Variable' if not _GLOBALS else 'Variable_%d' % len(_GLOBALS))
        self._reader = None          # set by the step that owns the value (the baseline lives on the device)
        _GLOBALS.append(self)

    def read(self):
        return self._reader() if self._reader is not None else self.initial_value

    def __repr__(self):
        return '<Variable %s>' % self.name


class ModelVariable(Sym):
    """A trainable variable of a built model, by its reference name (what tf.global_variables() lists)."""

    def __init__(self, model, name, shape):
        self.model, self.name, self.shape, self.trainable = model, name, shape, True

    def __repr__(self):
        return '<tf.Variable %s %s>' % (self.name, self.shape)


_GLOBALS: List[Sym] = []


class _ControlDependencies:
    def __init__(self, ops):
        self.ops = list(ops)

    def __enter__(self):
        _DEPS.append(self.ops)
        return self

    def __exit__(self, *exc):
        _DEPS.pop()
        return False


# ---------------------------------------------------------------------------------------------------
# optimizer surface
# ---------------------------------------------------------------------------------------------------
class Gradient(Sym):
    def __init__(self, loss, var, clip_norm=None):
        self.loss, self.var, self.clip_norm = loss, var, clip_norm


def clip_by_norm(t, clip_norm, axes=None, name=None):
    if not isinstance(t, Gradient):
        raise NotImplementedError('tf.clip_by_norm is supported on the gradients of compute_gradients only')
    return Gradient(t.loss, t.var, float(clip_norm))


class TrainOp(Sym):
    def __init__(self, optimizer, loss, clip_norm, model):
        self.optimizer, self.loss, self.clip_norm, self.model = optimizer, loss, clip_norm, model


class AdamOptimizer:
    """tf.train.AdamOptimizer(learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08)"""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-08, use_locking=False, name='Adam'):
        self.hyper = dict(lr=float(learning_rate), beta1=float(beta1), beta2=float(beta2), eps=float(epsilon))

    def compute_gradients(self, loss, var_list=None, **kw):
        if var_list is not None:
            raise NotImplementedError('compute_gradients(var_list=...): the step updates every variable')
        model = _model_of(loss)
        return [(Gradient(loss, v), v) for v in model_variables(model)]

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gv = list(grads_and_vars)
        if not gv:
            raise ValueError('apply_gradients: no gradients')
        g0 = gv[0][0]
        model = _model_of(g0.loss)
        want = [v.name for v in model_variables(model)]
        if [v.name for _, v in gv] != want:
            raise NotImplementedError('apply_gradients: the step updates every trainable variable of the model, '
                                      'in compute_gradients order')
        for g, v in gv:
            if not isinstance(g, Gradient) or g.loss is not g0.loss or g.var is not v:
                raise NotImplementedError('apply_gradients: gradients must come from one compute_gradients call')
            if g.clip_norm != g0.clip_norm:
                raise NotImplementedError('apply_gradients: one clip norm for every tensor '
                                          '(train_clevr_gt_layout.py:121-122)')
        return TrainOp(self, g0.loss, g0.clip_norm, model)

    def minimize(self, loss, **kw):
        return self.apply_gradients(self.compute_gradients(loss))


def model_variables(model) -> List[ModelVariable]:
    if getattr(model, '_tf_variables', None) is None:
        model._tf_variables = [ModelVariable(model, n, s) for n, s in model.variable_shapes().items()]
    return model._tf_variables


def global_variables():
    """tf.global_variables(): every model variable built so far + the script's own tf.Variables
    (rl :168: `[v for v in tf.global_variables() if v != baseline]`)."""
    from . import runtime
    out: List[Sym] = []
    for m in runtime._MODELS:
        out += model_variables(m)
    return out + [v for v in _GLOBALS]


# ---------------------------------------------------------------------------------------------------
# matching the loss graph
# ---------------------------------------------------------------------------------------------------
def _is_fetch(x, name):
    from .runtime import Fetch
    return isinstance(x, Fetch) and x.name == name


def _num(x):
    """python / numpy scalar or a Const -> float, else None"""
    if isinstance(x, Const):
        x = x.value
    if isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool):
        return float(x)
    return None


def _terms(e, coef=1.0):
    """e as a list of (coefficient, node): flattens add / add_n / sub / neg and scalar multiples"""
    if isinstance(e, Op):
        if e.kind == 'add':
            return _terms(e.args[0], coef) + _terms(e.args[1], coef)
        if e.kind == 'add_n':
            return [t for a in e.args for t in _terms(a, coef)]
        if e.kind == 'sub':
            return _terms(e.args[0], coef) + _terms(e.args[1], -coef)
        if e.kind == 'neg':
            return _terms(e.args[0], -coef)
        if e.kind == 'mul':
            a, b = e.args
            if _num(a) is not None:
                return _terms(b, coef * _num(a))
            if _num(b) is not None:
                return _terms(a, coef * _num(b))
    n = _num(e)
    if n is not None:
        return [(coef * n, None)]
    return [(coef, e)]


def _model_of(e):
    """the model whose fetches the expression is built from"""
    from .runtime import Fetch
    found = []

    def walk(x):
        if isinstance(x, Fetch):
            if x.owner not in found:
                found.append(x.owner)
        elif isinstance(x, Op):
            for a in x.args:
                walk(a)
        elif isinstance(x, (Gradient,)):
            walk(x.loss)
    walk(e)
    owners = [getattr(o, 'model', o) for o in found]
    models = []
    for o in owners:
        if o not in models:
            models.append(o)
    if len(models) != 1:
        raise NotImplementedError('the loss must be built from the outputs of exactly one model (found %d)' % len(models))
    return models[0]


class LossPlan:
    """What the matched loss graph asks of the step."""

    def __init__(self):
        self.objective = None           # 0 behavioural cloning, 1 policy gradient
        self.weight_decay = 0.0
        self.lambda_entropy = 0.0
        self.invalid_expr_loss = None
        self.labels_ph = None
        self.validity_ph = None
        self.baseline = None
        self.avg_sample_loss = None     # the node the scripts fetch as `avg_sample_loss`
        self.policy_gradient = None

    def __repr__(self):
        return 'LossPlan(%s)' % ', '.join('%s=%r' % kv for kv in self.__dict__.items()
                                          if kv[0] not in ('avg_sample_loss', 'policy_gradient'))


def _match_ce(e, plan):
    """tf.nn.sparse_softmax_cross_entropy_with_logits(logits=model.scores, labels=<placeholder>)"""
    from .runtime import Placeholder
    if isinstance(e, Op) and e.kind == 'softmax_ce' and _is_fetch(e.args[0], 'scores') and \
            isinstance(e.args[1], Placeholder):
        if plan.labels_ph is not None and plan.labels_ph is not e.args[1]:
            return False
        plan.labels_ph = e.args[1]
        return True
    return False


def _match_final(e, plan):
    """the per-sample loss: CE (gt :104-108), or where(validity, CE, ones_like(CE) * invalid_expr_loss) (rl :107-113).
    Returns 'ce' / 'where' / None."""
    from .runtime import Placeholder
    if _match_ce(e, plan):
        return 'ce'
    if isinstance(e, Op) and e.kind == 'where' and isinstance(e.args[0], Placeholder) and _match_ce(e.args[1], plan):
        t = _terms(e.args[2])
        if len(t) == 1 and isinstance(t[0][1], Op) and t[0][1].kind == 'ones_like' and _match_ce(t[0][1].args[0], plan):
            if plan.validity_ph is not None and plan.validity_ph is not e.args[0]:
                return None
            if plan.invalid_expr_loss is not None and plan.invalid_expr_loss != t[0][0]:
                return None
            plan.validity_ph, plan.invalid_expr_loss = e.args[0], t[0][0]
            return 'where'
    return None


def match_loss(total_loss) -> LossPlan:
    plan = LossPlan()
    seen = dict(seq=0.0, avg=0.0, pg=0.0)
    kinds = set()

    def refuse(node, why='is not a term of the reference loss graphs'):
        raise NotImplementedError(
            'the loss handed to compute_gradients is not one the training step implements: term %r %s '
            '(exp_clevr/train_clevr_gt_layout.py:104-113, train_clevr_rl_gt_layout.py:107-129)' % (node, why))

    for coef, node in _terms(total_loss):
        if node is None:
            continue                                           # an additive constant: no gradient
        if _is_fetch(node, 'l2_reg'):
            plan.weight_decay += coef
        elif _is_fetch(node, 'entropy_reg'):
            plan.lambda_entropy += coef
        elif isinstance(node, Op) and node.kind == 'mean':
            inner = _terms(node.args[0])
            if len(inner) == 1 and _is_fetch(inner[0][1], 'log_seq_prob'):
                seen['seq'] += -coef * inner[0][0]             # mean(-log_seq_prob) counts +1
                continue
            if len(inner) == 1 and inner[0][0] == 1.0:
                x = inner[0][1]
                k = _match_final(x, plan)
                if k:
                    kinds.add(k)
                    seen['avg'] += coef
                    plan.avg_sample_loss = node
                    continue
                # mean(stop_gradient(final - baseline) * log_seq_prob)
                if isinstance(x, Op) and x.kind == 'mul':
                    a, b = x.args
                    if _is_fetch(a, 'log_seq_prob'):
                        a, b = b, a
                    if _is_fetch(b, 'log_seq_prob') and isinstance(a, Op) and a.kind == 'stop_gradient':
                        d = a.args[0]
                        if isinstance(d, Op) and d.kind == 'sub' and isinstance(d.args[1], Variable):
                            k = _match_final(d.args[0], plan)
                            if k:
                                kinds.add(k)
                                if plan.baseline is not None and plan.baseline is not d.args[1]:
                                    refuse(node, 'uses a second baseline variable')
                                plan.baseline = d.args[1]
                                plan.policy_gradient = node
                                seen['pg'] += coef
                                continue
            refuse(node)
        else:
            refuse(node)
    if seen['avg'] != 1.0:
        refuse(total_loss, 'must contain mean(per-sample loss) exactly once')
    if seen['pg'] == 0.0 and seen['seq'] == 1.0 and kinds == {'ce'} and plan.lambda_entropy == 0.0:
        plan.objective = 0
    elif seen['pg'] == 1.0 and seen['seq'] == 0.0 and kinds <= {'where', 'ce'} and plan.baseline is not None:
        plan.objective = 1
        if plan.invalid_expr_loss is None:
            plan.invalid_expr_loss = 0.0
    else:
        refuse(total_loss, 'combines its terms with coefficients the step does not implement (%r)' % (seen,))
    return plan


def match_baseline_update(op, plan: LossPlan) -> float:
    """tf.assign_add(baseline, (1 - decay) * (avg_sample_loss - baseline)) -> decay (rl :121-122)"""
    if not (isinstance(op, Op) and op.kind == 'assign_add' and op.args[0] is plan.baseline):
        raise NotImplementedError('unsupported control dependency %r' % (op,))
    t = _terms(op.args[1])
    ok = len(t) == 2 and {id(t[0][1]), id(t[1][1])} == {id(plan.avg_sample_loss), id(plan.baseline)}
    if ok:
        ca = [c for c, n in t if n is plan.avg_sample_loss][0]
        cb = [c for c, n in t if n is plan.baseline][0]
        ok = ca > 0 and abs(ca + cb) < 1e-12
    if not ok:
        raise NotImplementedError('the baseline update must be (1 - decay) * (avg_sample_loss - baseline), got %r'
                                  % (op.args[1],))
    return 1.0 - ca


# ---------------------------------------------------------------------------------------------------
# numpy evaluation of what a script may fetch besides the matched nodes (sums of losses, ...)
# ---------------------------------------------------------------------------------------------------
def evaluate(e, env):
    """env: id(node) -> value for fetched / matched nodes and fed placeholders"""
    if id(e) in env:
        return env[id(e)]
    n = _num(e)
    if n is not None:
        return np.float32(n)
    if isinstance(e, Op):
        a = [evaluate(x, env) for x in e.args]
        k = e.kind
        if k == 'neg':
            return -a[0]
        if k == 'add':
            return a[0] + a[1]
        if k == 'add_n':
            return sum(a[1:], a[0])
        if k == 'sub':
            return a[0] - a[1]
        if k == 'mul':
            return a[0] * a[1]
        if k == 'div':
            return a[0] / a[1]
        if k == 'mean':
            return np.mean(a[0], dtype=np.float32)
        if k == 'stop_gradient':
            return a[0]
        if k == 'ones_like':
            return np.ones_like(a[0])
        if k == 'where':
            return np.where(a[0], a[1], a[2])
        if k == 'softmax_ce':
            z = np.asarray(a[0], np.float32)
            z = z - z.max(axis=1, keepdims=True)
            lse = np.log(np.exp(z).sum(axis=1))
            return (lse - z[np.arange(z.shape[0]), np.asarray(a[1], np.int64)]).astype(np.float32)
    raise ValueError('cannot evaluate %r from what this run computed' % (e,))


# ---------------------------------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------------------------------
def _new_trainer(model, plan: LossPlan, op: TrainOp):
    from .train import Trainer
    if hasattr(model, 'new_trainer'):             # a face with its own Trainer (models_vqa: dropout)
        return model.new_trainer(plan, op)
    h = op.optimizer.hyper
    return Trainer(model.engine, weight_decay=plan.weight_decay, lr=h['lr'], beta1=h['beta1'], beta2=h['beta2'],
                   eps=h['eps'], max_grad_l2_norm=op.clip_norm if op.clip_norm is not None else 0.0)


class TrainStep:
    """Executes a fetched `train_step`: one iteration of the script's loop body on the model's Trainer."""

    def __init__(self, const: Const):
        ops = [d for d in const.deps if isinstance(d, TrainOp)]
        if len(ops) != 1:
            raise NotImplementedError('a train step is one apply_gradients op under tf.control_dependencies')
        self.op = ops[0]
        self.model = self.op.model
        self.plan = match_loss(self.op.loss)
        self.baseline_decay = None
        for d in const.deps:
            if d is self.op:
                continue
            self.baseline_decay = match_baseline_update(d, self.plan)
        if self.plan.objective == 1 and self.baseline_decay is None:
            raise NotImplementedError('the policy-gradient step updates its baseline in the same run (rl :131-132)')
        self.trainer = None

    def ensure_trainer(self):
        if self.trainer is None:
            plan = self.plan
            self.trainer = _new_trainer(self.model, plan, self.op)
            if plan.objective == 1:
                self.trainer.rl.update(invalid_expr_loss=plan.invalid_expr_loss, lambda_entropy=plan.lambda_entropy,
                                       baseline_decay=self.baseline_decay)
                self.trainer.set_baseline(float(plan.baseline.read()))
                plan.baseline._reader = self.trainer.get_baseline
        return self.trainer

    def run(self, handle):
        """forward + backward + clip + Adam on the feeds of this partial_run handle; fills handle.results"""
        from .runtime import resolve, to_numpy
        m, plan = self.model, self.plan
        self.ensure_trainer()
        if handle.phase1 is None:
            handle.phase1 = m.run_phase1_training(handle) if hasattr(m, 'run_phase1_training') else \
                m.run_phase1(handle.feeds)
        tokens = to_numpy(handle.phase1['predicted_tokens'])
        s2s = getattr(m, 'att_seq2seq', m)._inputs
        batch = dict(input_seq_batch=resolve(s2s['input_seq'], handle.feeds),
                     seq_length_batch=resolve(s2s['seq_len'], handle.feeds),
                     image_feat_batch=resolve(m.image_feat_grid, handle.feeds),
                     answer_label_batch=np.asarray(resolve(plan.labels_ph, handle.feeds), np.int32))
        # (the layouts the script assembled and fed back are the tokens of phase 1 -- the ground-truth
        # layout under use_gt_layout, the sampled one otherwise; the Trainer assembles them itself)
        resolve(m.compiler.loom_input_tensor, handle.feeds)           # (must have been fed, like TF's loom input)
        tr = self.trainer
        scale = tr.forward_backward(batch, tokens, objective=plan.objective)
        if plan.validity_ph is not None:
            fed = np.asarray(resolve(plan.validity_ph, handle.feeds), bool)
            if not np.array_equal(fed, np.asarray(tr.last_validity, bool)):
                raise ValueError('expr_validity_batch differs from the validity of the layouts the decoder produced')
        tr.apply(scale)
        losses = to_numpy(tr.losses)
        handle.results['scores'] = tr.scores
        env = handle.env
        env[id(plan.avg_sample_loss)] = np.float32(losses[0])
        if plan.policy_gradient is not None:
            env[id(plan.policy_gradient)] = np.float32(losses[1])
        return np.int32(0)


# ---------------------------------------------------------------------------------------------------
# tf.summary: scalar summaries written as TensorBoard event files
# ---------------------------------------------------------------------------------------------------
class _Scalar(Sym):
    def __init__(self, tag, tensor):
        self.tag, self.tensor = tag, tensor


class _Merged(Sym):
    def __init__(self, parts):
        self.parts = list(parts)


class SummaryValue:
    """what sess.run(merged_summary, feed_dict) returns: tag -> value"""

    def __init__(self, values: Dict[str, float]):
        self.values = values


def _vi(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _event(wall_time: float, step: int, values: Optional[Dict[str, float]] = None, file_version: str = None) -> bytes:
    """tensorflow/core/util/event.proto: wall_time = 1 (double), step = 2 (int64), file_version = 3, summary = 5;
    summary.proto: Summary.value = 1 {tag = 1, simple_value = 2 (float)}"""
    ev = b'\x09' + struct.pack('<d', wall_time) + b'\x10' + _vi(step)
    if file_version is not None:
        fv = file_version.encode()
        ev += b'\x1a' + _vi(len(fv)) + fv
    if values is not None:
        summ = b''
        for tag, v in values.items():
            t = tag.encode()
            val = b'\x0a' + _vi(len(t)) + t + b'\x15' + struct.pack('<f', float(v))
            summ += b'\x0a' + _vi(len(val)) + val
        ev += b'\x2a' + _vi(len(summ)) + summ
    return ev


class FileWriter:
    """tf.summary.FileWriter(logdir, graph): TFRecord-framed Event protos in
    `<logdir>/events.out.tfevents.<time>.<host>` (length, masked crc32c of the length, payload, masked crc32c)."""

    def __init__(self, logdir, graph=None, **kw):
        import socket
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._write(_event(time.time(), 0, file_version='brain.Event:2'))

    def _write(self, payload: bytes):
        from .tf_checkpoint import crc32c, mask_crc
        head = struct.pack('<Q', len(payload))
        self._f.write(head + struct.pack('<I', mask_crc(crc32c(head))) + payload +
                      struct.pack('<I', mask_crc(crc32c(payload))))
        self._f.flush()

    def add_summary(self, summary, global_step=None):
        if not isinstance(summary, SummaryValue):
            raise TypeError('add_summary expects what sess.run(<merged summary>) returned')
        self._write(_event(time.time(), int(global_step or 0), summary.values))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """[(step, {tag: value})] of an event file written by FileWriter (tests; checks both checksums)"""
    from .tf_checkpoint import crc32c, mask_crc, _proto_fields
    out, raw, pos = [], open(path, 'rb').read(), 0
    while pos < len(raw):
        n, = struct.unpack_from('<Q', raw, pos)
        if struct.unpack_from('<I', raw, pos + 8)[0] != mask_crc(crc32c(raw[pos:pos + 8])):
            raise ValueError('event file: bad length checksum')
        payload = raw[pos + 12:pos + 12 + n]
        if struct.unpack_from('<I', raw, pos + 12 + n)[0] != mask_crc(crc32c(payload)):
            raise ValueError('event file: bad payload checksum')
        pos += 16 + n
        step, vals = 0, None
        for num, wt, v in _proto_fields(payload):
            if num == 2:
                step = v
            elif num == 5:
                vals = {}
                for n1, _, sv in _proto_fields(v):
                    tag, val = None, None
                    for n2, w2, x in _proto_fields(sv):
                        if n2 == 1:
                            tag = x.decode()
                        elif n2 == 2:
                            val = struct.unpack('<f', struct.pack('<I', x))[0] if isinstance(x, int) else \
                                struct.unpack('<f', x)[0]
                    vals[tag] = val
        if vals is not None:
            out.append((step, vals))
    return out


# ---------------------------------------------------------------------------------------------------
# tf.global_variables_initializer(): the reference's declared initializers
# ---------------------------------------------------------------------------------------------------
def initial_weights(shapes: Dict[str, tuple], seed: int = 0) -> Dict[str, np.ndarray]:
    """fc / conv / 1x1 weights: Xavier uniform (util/cnn.py:14,53,101, util/empty_safe_conv.py:22,
    nmn3_netgen_att.py:150-151,157-158); biases: zeros (constant_initializer(0.)); variables created by
    tf.get_variable WITHOUT an initializer (embedding_mat, go_embedding, att_prediction/v, the BasicLSTMCell
    weights; nmn3_netgen_att.py:84,142,146,149) take TensorFlow 1.0.0's scope default,
    uniform_unit_scaling_initializer(factor=1.0): U(+-sqrt(3 / prod(shape[:-1]))) [TF-semantics, restated;
    TF's random stream itself is not reproducible here]; BasicLSTMCell biases: zeros."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes.items():
        leaf = name.rsplit('/', 1)[-1]
        default_init = leaf in ('embedding_mat', 'go_embedding', 'v') or 'basic_lstm_cell' in name
        if leaf == 'biases':
            w = np.zeros(shape, np.float32)
        elif default_init:
            fan = int(np.prod(shape[:-1])) if len(shape) > 1 else int(shape[0])
            lim = np.sqrt(3.0 / max(fan, 1))
            w = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = rf * shape[-2], rf * shape[-1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        out[name] = w
    return out
